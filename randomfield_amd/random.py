"""
Generate random numbers -- host-side mirror of ``randomfield/random.py``.

:func:`randomize` works on host arrays like the reference's.  On the GPU path
the multiplication by deviates happens inside the first FFT pass; this module
supplies the deviates of the reference stream (:func:`reference_normals`) when
same-seed parity with the reference is requested.
"""
from __future__ import annotations

import numpy as np

from . import transform

__all__ = ["randomize", "reference_normals"]


def reference_normals(seed, size):
    """``RandomState(seed).normal(size=size)``: MT19937 + polar method, the
    stream ``randomize`` draws (random.py:24-28).  ``seed=None`` seeds from the OS."""
    return np.random.RandomState(seed).normal(size=size)


def randomize(data, seed=None):
    """
    Randomize data by multiplying existing sigma values by normal deviates
    (random.py:12-29): imag := real, then every real component of the flat
    interleaved view is scaled by its own N(0,1) deviate (float64 product rounded
    to the array dtype).  The global numpy RNG state is left alone.
    """
    data.imag = data.real
    real_type = transform.scalar_type(data.dtype)
    real_size = 2 * data.size
    sigmas = data.view(real_type).reshape(real_size)
    sigmas *= reference_normals(seed, real_size)
    return data
