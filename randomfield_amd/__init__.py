"""
randomfield_amd -- MI355X-native Gaussian random field generation with the
``randomfield`` (dkirkby/randomfield) Generator / Plan API for its Fourier-space
sampling path.

    from randomfield_amd import Generator
    g = Generator(1024, 1024, 1024, 2.5, rng='native')
    delta = g.generate_delta_field(seed=123, save_potential=False)

The compute path is hand-written HIP for gfx950 behind a C-ABI
(``include/randomfield_hip.h``), loaded with ctypes; see DESIGN.md.
"""
from .generate import Generator  # noqa: F401
from . import transform, powertools, cosmotools  # noqa: F401
from . import random  # noqa: F401

__version__ = "0.1.0"
