"""
Cosmology side-car -- the part of ``randomfield/cosmotools.py`` that touches the
N^3 grid (:func:`apply_lognormal_transform`, cosmotools.py:206-221) plus thin
wrappers for the O(nz) background tables.

The reference computes redshifts, growth function and mean matter densities
with astropy (cosmotools.py:14-203).  Those are host-side O(nz) tables that the
hot path only *consumes*; astropy is optional here: when it is missing,
:class:`randomfield_amd.generate.Generator` accepts the tables as arrays.
"""
from __future__ import annotations

import numpy as np

__all__ = ["apply_lognormal_transform", "lognormal_tables", "create_cosmology", "have_astropy"]


def have_astropy():
    try:
        import astropy.cosmology  # noqa: F401
        return True
    except Exception:
        return False


def create_cosmology(*args, **kwargs):
    """Create a background cosmology (cosmotools.py:14-45); needs astropy."""
    try:
        import astropy.cosmology
    except ImportError:
        raise ImportError("astropy is required for create_cosmology(); pass growth_function= / "
                          "mean_matter_density= arrays to Generator instead.")
    if len(args) > 0 and len(kwargs) > 0:
        raise TypeError("Cannot specify both a name and parameters.")
    if len(args) > 1:
        raise TypeError("Invalid arguments: expected a name or parameters.")
    if len(args) == 1:
        if not isinstance(args[0], str):
            raise TypeError("Invalid arguments: expected a name or parameters.")
        if args[0] not in ("WMAP5", "WMAP7", "WMAP9", "Planck13"):
            raise ValueError("Unknown cosmology: {0}.".format(args[0]))
        return getattr(astropy.cosmology, args[0])
    if kwargs:
        return astropy.cosmology.FlatLambdaCDM(**kwargs)
    return astropy.cosmology.Planck13


def lognormal_tables(growth, sigma, nz):
    """The two float64 (nz,) tables of the lognormal map: a = sqrt(log t),
    b = sqrt(t) with t = 1 + (sigma*growth)**2 (cosmotools.py:216)."""
    g = np.broadcast_to(np.asarray(growth, np.float64), (nz,))
    t = 1 + (float(sigma) * g) ** 2
    return np.sqrt(np.log(t)), np.sqrt(t)


def apply_lognormal_transform(delta, growth, sigma=None):
    """
    Transform delta values drawn from a normal distribution with mean zero and
    standard deviation sigma to have a log-normal distribution with mean one
    and standard deviation growth * sigma (cosmotools.py:206-221).  Transforms
    are applied in place, overwriting the input delta field.  If sigma is not
    specified, np.std(delta) will be used.  Host (numpy) version.
    """
    if sigma is None:
        sigma = np.std(delta)
    t = 1 + (sigma * growth) ** 2
    delta /= sigma
    delta *= np.sqrt(np.log(t))
    delta = np.exp(delta, out=delta)
    delta /= np.sqrt(t)
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
