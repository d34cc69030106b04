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
    (random.py:12-29).  On entry the real parts hold sigma; on exit every complex
    cell is sigma * (g_re + i g_im) with consecutive deviates of the stream going
    to (re, im) of consecutive cells in C order; each product is formed in float64
    and rounded once to the array's type.  The global numpy RNG state is left alone.
    """
    pairs = data.view(transform.scalar_type(data.dtype)).reshape(-1, 2)     # one (re, im) row per cell, same memory
    pairs[:, 1] = pairs[:, 0]
    np.multiply(pairs, reference_normals(seed, pairs.size).reshape(pairs.shape), out=pairs, casting="same_kind")
    return data
