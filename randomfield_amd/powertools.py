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


def grid_k_range(shape, spacing):
    """(k_min, k_max) of an (nx, ny, nz) grid: the fundamental mode of the longest axis and the corner of the
    Nyquist cube (powertools.py:19-23)."""
    fundamental = 2 * np.pi / spacing
    return fundamental / max(shape), fundamental * np.sqrt(3) / 2


def get_k_bounds(data, spacing, packed=True):
    """Bounds of wavenumber values for the specified grid (powertools.py:16-24)."""
    return grid_k_range(transform.expanded_shape(data, packed=packed), spacing)


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

    Note that the value at [0, 0, 0] will be log10(0) = -inf.  Built from the
    three axis tables: the (nx, ny) plane of kx**2 + ky**2 is formed in float64
    and rounded to the array's real type once, kz**2 is then added to it in
    float64 and the sum rounded again -- the same two roundings per cell as the
    reference's in-place ufunc chain, which is what makes the result
    bit-identical to it -- then log10 and the halving in the array's real type.
    """
    kx2, ky2, kz2 = ksq_axes(*transform.expanded_shape(data, packed=packed), spacing, packed=packed)
    re = data.real
    plane = np.add.outer(kx2, ky2).astype(re.dtype)
    np.add(plane[:, :, None], kz2[None, None, :], out=re, casting="same_kind")
    with np.errstate(divide="ignore"):
        np.log10(re, out=re)
    np.multiply(re, re.dtype.type(0.5), out=re)
    data.imag = 0
    return data


# validate_power: (test on the table -> True when it FAILS, message), checked in this order (powertools.py:64-82)
_POWER_RULES = (
    (lambda p: not np.all(np.isfinite(p["k"])), "Power spectrum has some invalid values of k."),
    (lambda p: not np.all(np.isfinite(p["Pk"])), "Power spectrum has some invalid values of P(k)."),
    (lambda p: not np.array_equal(p["k"], np.unique(p["k"])), "Power spectrum k values are not strictly increasing."),
    (lambda p: p["k"][0] <= 0, "Power spectrum min(k) is <= 0."),
    (lambda p: np.any(p["Pk"] < 0), "Power values P(k) are not all non-negative."),
)


def validate_power(power):
    """Validates a power spectrum (powertools.py:64-82); returns it."""
    if not isinstance(power, np.ndarray):
        raise ValueError("Invalid type for power: {0}.".format(type(power)))
    if not {"k", "Pk"} <= set(power.dtype.names or ()):
        raise ValueError('Missing required fields "k", "Pk" in power.')
    for fails, message in _POWER_RULES:
        if fails(power):
            raise ValueError(message)
    return power


def filter_power(power, sigma, out=None):
    """
    Apply a Gaussian filtering to a power spectrum (powertools.py:85-122):
    P(k) -> P(k) * exp(-(k*sigma)**2), i.e. delta(r) is convolved with a 3D
    Gaussian of width sigma.  ``out=None`` returns a filtered copy, ``out=power``
    filters in place, any other ``out`` receives the filtered table.
    """
    if sigma < 0:
        raise ValueError("Invalid smoothing sigma: {0}.".format(sigma))
    if out is None:
        out = power.copy()
    elif out is not power:
        if validate_power(power).shape != out.shape:
            raise ValueError("Output power has wrong shape: {0}.".format(out.shape))
        out[...] = power
    if sigma > 0:
        np.multiply(out["Pk"], np.exp(-np.square(power["k"] * sigma)), out=out["Pk"])
    return out


def sigma_table(power, shape, spacing):
    """
    The two float64 tables of the sigma(k) interpolator (powertools.py:139-154):
    x_i = log10 k_i and s_i = N3 * sqrt(P_i / (2 Vbox)), after the reference's
    range check that the table covers the grid's [k_min, k_max].
    These are what the HIP kernels consume (``rf_set_power``).
    """
    k, Pk = validate_power(power)["k"], power["Pk"]
    cells = shape[0] * shape[1] * shape[2]
    volume = cells * spacing ** 3
    table_lo, table_hi = np.min(k), np.max(k)
    if table_lo <= 0:
        raise ValueError("Power uses min(k) <= 0: {0}.".format(table_lo))
    grid_lo, grid_hi = grid_k_range(shape, spacing)
    if table_lo > grid_lo or table_hi < grid_hi:
        raise ValueError("Power k range [{0}:{1}] does not cover data k range [{2}:{3}]."
                         .format(table_lo, table_hi, grid_lo, grid_hi))
    return np.log10(k), cells * np.sqrt(Pk / (2 * volume))


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


def load_default_power(scaled_by_h=True, h=None):
    """
    Loads the default power spectrum P(k, z=0) (powertools.py:167-197): 500 rows,
    1e-4 <= k <= 22 h/Mpc, Planck13 via CLASS; units h/Mpc and (Mpc/h)**3.

    The table is stored as ``data/default_power.npz`` (binary copy of the
    reference's data file).  ``scaled_by_h=False`` converts to 1/Mpc and Mpc**3 and
    needs the Hubble parameter ``h`` (the reference takes it from its astropy
    cosmology, which is outside this package's scope: Planck13 has h = 0.6777).
    """
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "default_power.npz")
    try:
        table = np.load(path)
    except IOError:
        raise RuntimeError("Unable to load default_power.npz")
    power = make_power(table["k"], table["Pk"])
    if scaled_by_h is False:
        if h is None:
            raise ValueError("load_default_power(scaled_by_h=False) needs h= (e.g. 0.6777 for Planck13).")
        power["k"] *= h
        power["Pk"] /= h ** 3
    return power
