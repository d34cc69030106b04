"""
Cosmology side-car -- the part of ``randomfield/cosmotools.py`` that touches the
N^3 grid: :func:`apply_lognormal_transform` (cosmotools.py:206-221).

The reference derives redshifts, growth function, mean matter densities and the
transverse distance from an astropy cosmology (cosmotools.py:14-203).  Those are
host-side O(nz) tables that the hot path only *consumes*; building them is out of
scope here (cosmology is not on the accelerated path), so
:class:`randomfield_amd.generate.Generator` takes them as (nz,) arrays.
"""
from __future__ import annotations

import numpy as np

__all__ = ["apply_lognormal_transform", "lognormal_tables", "simps_avg"]


def lognormal_tables(growth, sigma, nz):
    """The two float64 (nz,) tables of the lognormal map: a = sqrt(log t),
    b = sqrt(t) with t = 1 + (sigma*growth)**2 (cosmotools.py:216)."""
    g = np.broadcast_to(np.asarray(growth, np.float64), (nz,))
    t = 1 + (float(sigma) * g) ** 2
    return np.sqrt(np.log(t)), np.sqrt(t)


def apply_lognormal_transform(delta, growth, sigma=None):
    """
    Map a zero-mean normal field of standard deviation ``sigma`` (default: ``np.std(delta)``) onto a
    log-normal field with mean one and standard deviation ``growth * sigma``, in place
    (cosmotools.py:206-221); ``growth`` is a scalar or broadcasts against ``delta`` (per-z tables).
    Host (numpy) version: with spread = 1 + (sigma * growth)**2,
    delta -> exp(delta / sigma * sqrt(log spread)) / sqrt(spread), evaluated in that order.
    """
    sigma = np.std(delta) if sigma is None else sigma
    spread = np.square(sigma * growth) + 1
    np.divide(delta, sigma, out=delta)
    np.multiply(delta, np.sqrt(np.log(spread)), out=delta)
    np.exp(delta, out=delta)
    np.divide(delta, np.sqrt(spread), out=delta)
    return delta


def simps_avg(y, h):
    """Composite Simpson rule along the last axis for uniformly spaced samples (step ``h``) with the
    even-count handling of ``scipy.integrate.simps(even='avg')`` -- the routine the reference calls in
    ``calculate_lensing_potential`` (generate.py:405-406), which current scipy no longer ships: for an even
    number of samples, the average of (Simpson on the first N-1 + trapezoid on the last interval) and
    (trapezoid on the first interval + Simpson on the last N-1)."""
    y = np.asarray(y, np.float64)
    N = y.shape[-1]

    def basic(a):            # odd number of samples
        if a.shape[-1] < 3:
            return np.zeros(a.shape[:-1])
        return (h / 3.0) * (a[..., 0] + a[..., -1] + 4.0 * a[..., 1:-1:2].sum(-1) + 2.0 * a[..., 2:-1:2].sum(-1))

    if N % 2:
        return basic(y)
    if N < 2:
        return np.zeros(y.shape[:-1])
    first = basic(y[..., :-1]) + 0.5 * h * (y[..., -2] + y[..., -1])
    last = 0.5 * h * (y[..., 0] + y[..., 1]) + basic(y[..., 1:])
    return 0.5 * (first + last)
