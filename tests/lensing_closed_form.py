"""Closed-form known answers for the lensing line-of-sight integral (generate.py:397-411), independent of any
restatement of scipy.integrate.simps: TEST TOOLING shared by the CPU (oracle) and GPU (kernel) tests.

The reference computes, for every end slice e >= i_min,
    psi[e] = Simpson_{j = i_min..e} of  f_e(D_j),   f_e(D) = -2 [cotK(D) - cotK(D_e)] phi(D),   D_j = j h,
with simps(even='avg') for an even number of samples.  cotK = cosK / DA is built from an INPUT table DA, so it
can be made exactly linear in D: DA_j = cosK_j / (a + b D_j) gives cotK(D) = a + b D in flat and curved models
alike, and with phi a polynomial of degree <= 2 the integrand is a polynomial of degree <= 3 in D:

* odd sample count: Simpson's rule is exact for cubics, so psi[e] is the analytic integral;
* even count: 'avg' averages (Simpson on the first N-1 samples + trapezoid on the last interval) with (trapezoid on
  the first interval + Simpson on the rest); a trapezoid on [u, u+h] exceeds the integral of a cubic by exactly
  h^3/12 f''(u + h/2), hence psi[e] = integral + h^3/24 [f''(first midpoint) + f''(last midpoint)];
* one sample (e = i_min): 0.
"""
import numpy as np
from numpy.polynomial import Polynomial


def tables(nz, h, K, a=0.3, b=-0.0037):
    """DC, DA and the cotK table (cotK[0] = 1 as in generate.py:394-395) with cotK(D) = a + b D for D > 0."""
    DC = np.arange(nz) * h
    if K < 0:
        cosK = np.cosh(np.sqrt(-K) * DC)
    elif K > 0:
        cosK = np.cos(np.sqrt(K) * DC)
    else:
        cosK = np.ones(nz)
    DA = np.ones(nz)
    DA[1:] = cosK[1:] / (a + b * DC[1:])
    return DC, DA


def expected(phi_coeffs, nz, h, i_min, a=0.3, b=-0.0037):
    """psi[e] for phi(D) = sum_k phi_coeffs[k] D**k, e = 0..nz-1 (i_min >= 1 so that the cotK[0] = 1 special value
    is never sampled)."""
    assert i_min >= 1
    phi = Polynomial(phi_coeffs)
    out = np.zeros(nz)
    for e in range(i_min, nz):
        De, D0 = e * h, i_min * h
        f = Polynomial([-De, 1.0]) * phi * (-2.0 * b)           # -2 b (D - De) phi(D)
        F = f.integ()
        val = F(De) - F(D0)
        n = e - i_min + 1
        if n == 1:
            val = 0.0
        elif n % 2 == 0:
            f2 = f.deriv(2)
            val += h ** 3 / 24.0 * (f2(D0 + 0.5 * h) + f2(De - 0.5 * h))
        out[e] = val
    return out
