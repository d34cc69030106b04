"""
Utilities for working with tabulated power spectra -- host-side mirror of
``randomfield/powertools.py`` (file:line citations are to the reference).

These functions operate on host (numpy) arrays exactly like the reference's and
exist so that the free-function API keeps working.  The GPU path does not call
them per cell: :class:`randomfield_amd.generate.Generator` hands the O(n) tables
computed by :func:`ksq_axes` and :func:`sigma_table` to the HIP kernels, which
evaluate rows K and T (|k| and sigma(k)) on the fly inside the first FFT pass.
"""
from __future__ import annotations

import os.path

import numpy as np

from . import transform

__all__ = ["get_k_bounds", "create_ksq_grids", "ksq_axes", "fill_with_log10k", "validate_power", "filter_power",
           "sigma_table", "tabulate_sigmas", "load_default_power", "make_power"]


def get_k_bounds(data, spacing, packed=True):
    """Bounds of wavenumber values for the specified grid (powertools.py:16-24)."""
    nx, ny, nz = transform.expanded_shape(data, packed=packed)
    k0 = (2 * np.pi) / spacing
    k_min = k0 / max(nx, ny, nz)
    k_max = k0 * np.sqrt(3) / 2
    return k_min, k_max


def ksq_axes(nx, ny, nz, spacing, packed=True):
    """Per-axis float64 tables of k_a(i)**2 (powertools.py:27-34).

    These three O(n) tables are what the HIP kernels consume (``rf_set_kgrid``)."""
    lambda0 = spacing / (2 * np.pi)
    kx = np.fft.fftfreq(nx, lambda0)
    ky = np.fft.fftfreq(ny, lambda0)
    kz = np.fft.fftfreq(nz, lambda0)
    if packed:
        kz = kz[:nz // 2 + 1]
    return kx ** 2, ky ** 2, kz ** 2


def create_ksq_grids(data, spacing, packed):
    """Sparse broadcastable grids of kx**2, ky**2, kz**2 (powertools.py:27-37)."""
    nx, ny, nz = transform.expanded_shape(data, packed=packed)
    kx2, ky2, kz2 = ksq_axes(nx, ny, nz, spacing, packed=packed)
    return np.meshgrid(kx2, ky2, kz2, sparse=True, indexing="ij")


def fill_with_log10k(data, spacing, packed=True):
    """
    Fill an array with values of log10(k) (powertools.py:40-61).

    Note that the value at [0, 0, 0] will be log10(0) = -inf.  The rounding
    chain of the reference is kept: the float64 sum kx2+ky2 is rounded to the
    array dtype, kz2 is added in float64 and rounded again, then log10 and the
    halving are done in the array dtype.
    """
    kx2_grid, ky2_grid, kz2_grid = create_ksq_grids(data, spacing, packed)
    data.imag = 0
    np.add(kx2_grid, ky2_grid, out=data.real, casting="same_kind")
    np.add(data.real, kz2_grid, out=data.real, casting="same_kind")
    with np.errstate(divide="ignore"):
        np.log10(data.real, out=data.real)
    data.real *= 0.5
    return data


def validate_power(power):
    """Validates a power spectrum (powertools.py:64-82)."""
    if not isinstance(power, np.ndarray):
        raise ValueError("Invalid type for power: {0}.".format(type(power)))
    names = power.dtype.names or ()
    if "k" not in names or "Pk" not in names:
        raise ValueError('Missing required fields "k", "Pk" in power.')
    if not np.all(np.isfinite(power["k"])):
        raise ValueError("Power spectrum has some invalid values of k.")
    if not np.all(np.isfinite(power["Pk"])):
        raise ValueError("Power spectrum has some invalid values of P(k).")
    if not np.array_equal(power["k"], np.unique(power["k"])):
        raise ValueError("Power spectrum k values are not strictly increasing.")
    if power["k"][0] <= 0:
        raise ValueError("Power spectrum min(k) is <= 0.")
    if np.any(power["Pk"] < 0):
        raise ValueError("Power values P(k) are not all non-negative.")
    return power


def filter_power(power, sigma, out=None):
    """
    Apply a Gaussian filtering to a power spectrum (powertools.py:85-122):
    P(k) -> P(k) * exp(-(k*sigma)**2), i.e. delta(r) is convolved with a 3D
    Gaussian of width sigma.
    """
    if sigma < 0:
        raise ValueError("Invalid smoothing sigma: {0}.".format(sigma))
    if out is None:
        out = np.copy(power)
    elif out is not power:
        validate_power(power)
        if out.shape != power.shape:
            raise ValueError("Output power has wrong shape: {0}.".format(out.shape))
        out[:] = power
    if sigma > 0:
        out["Pk"] *= np.exp(-(power["k"] * sigma) ** 2)
    return out


def sigma_table(power, shape, spacing):
    """
    The two float64 tables of the sigma(k) interpolator (powertools.py:139-154):
    x_i = log10 k_i and s_i = N3 * sqrt(P_i / (2 Vbox)), after the reference's
    range check that the table covers the grid's [k_min, k_max].
    These are what the HIP kernels consume (``rf_set_power``).
    """
    validate_power(power)
    nx, ny, nz = shape
    N3 = nx * ny * nz
    Vbox = N3 * spacing ** 3
    power_k_min, power_k_max = np.min(power["k"]), np.max(power["k"])
    if power_k_min <= 0:
        raise ValueError("Power uses min(k) <= 0: {0}.".format(power_k_min))
    k0 = (2 * np.pi) / spacing
    data_k_min, data_k_max = k0 / max(nx, ny, nz), k0 * np.sqrt(3) / 2
    if power_k_min > data_k_min or power_k_max < data_k_max:
        raise ValueError("Power k range [{0}:{1}] does not cover data k range [{2}:{3}]."
                         .format(power_k_min, power_k_max, data_k_min, data_k_max))
    log10_k = np.log10(power["k"])
    sigma = N3 * np.sqrt(power["Pk"] / (2 * Vbox))
    return log10_k, sigma


def tabulate_sigmas(data, power, spacing, packed=True):
    """
    Replace an array of log10(k) values with the corresponding sigmas
    (powertools.py:125-164):  sigma**2 = (nx*ny*nz) * P(k) / (2 * Vbox).

    sigma is interpolated linearly in log10(k), zero outside the table
    (so the -inf at [0,0,0] becomes 0).
    """
    shape = transform.expanded_shape(data, packed=packed)
    log10_k, sigma = sigma_table(power, shape, spacing)
    data.real = np.interp(data.real, log10_k, sigma, left=0.0, right=0.0)
    return data


def make_power(k, Pk):
    """Build the structured array (fields 'k', 'Pk') the API expects."""
    k = np.asarray(k, float)
    power = np.empty(len(k), dtype=[("k", float), ("Pk", float)])
    power["k"] = k
    power["Pk"] = Pk
    return power


def load_default_power(scaled_by_h=True):
    """
    Loads the default power spectrum P(k, z=0) (powertools.py:167-197): 500 rows,
    1e-4 <= k <= 22 h/Mpc, Planck13 via CLASS; units h/Mpc and (Mpc/h)**3.

    The table is stored as ``data/default_power.npz`` (binary copy of the
    reference's data file).  ``scaled_by_h=False`` needs the cosmology's h.
    """
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "default_power.npz")
    try:
        table = np.load(path)
    except IOError:
        raise RuntimeError("Unable to load default_power.npz")
    power = make_power(table["k"], table["Pk"])
    if scaled_by_h is False:
        from . import cosmotools
        h = cosmotools.create_cosmology().h
        power["k"] *= h
        power["Pk"] /= h ** 3
    return power
