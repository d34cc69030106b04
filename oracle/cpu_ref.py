"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (numpy) of the hot path of dkirkby/randomfield: the Fourier
space sampling of a Gaussian random field

    |k| -> sigma(k) -> sigma * N(0,1) -> Hermitian symmetrise -> 3-D c2r FFT
    -> delta(x) (+ rms, + optional lognormal map)

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and only as the checker.  The product
package ``randomfield_amd`` never imports it.

Every function cites the reference file:line it follows (paths relative to the
reference checkout, ``randomfield/<file>``).  The restatement is pinned in two
ways (see ``oracle/make_golden.py`` and ``tests/test_oracle_golden.py``):

* stage by stage against outputs of the reference's own functions run in the
  build container (committed as ``tests/golden/*.npz``), and
* against the known-answer spot values recorded in SURVEY.md section 8c.

The arithmetic that the reference delegates to third-party libraries is
restated here from their published algorithms:

* ``numpy.random.RandomState(seed).normal`` (MT19937 + legacy polar method)
  -> :func:`legacy_normals` is a from-scratch restatement used to pin the
  stream definition; :func:`randomize` itself calls numpy's RandomState, which
  is the same generator the reference calls (``random.py:24``).
* ``scipy.interpolate.interp1d(kind='linear', bounds_error=False,
  fill_value=0)`` -> :func:`interp_sigma` (piecewise linear, zero outside).
* ``numpy.fft.irfftn`` -> used directly (pocketfft); numpy >= 2 transforms
  complex64 in single precision, older numpy in double then rounds once.
  :func:`c2r` offers both.
"""
from __future__ import annotations

import math

import numpy as np

__all__ = [
    "expanded_shape", "packed_shape", "k_bounds", "ksq_axes", "fill_log10k",
    "sigma_table", "interp_sigma", "tabulate_sigmas", "legacy_normals",
    "randomize", "symmetrize_packed", "is_hermitian_packed", "c2r", "r2c",
    "generate_kspace", "generate_delta_field", "lognormal", "scale_z",
    "potential_kspace", "philox4x32", "philox4x32_10", "NATIVE_PHILOX_ROUNDS", "philox_normals", "native_noise_index",
    "native_noise", "default_like_power", "simps_avg", "cot_k", "lensing_potential",
]


# --------------------------------------------------------------------------
# shape helpers
# --------------------------------------------------------------------------

def expanded_shape(shape, packed=True):
    """Real-space shape of a (possibly packed) k-space array.

    Follows transform.py:46-60: x and y must be even; a packed array must have
    an odd last dimension ``nz//2+1`` (which forces nz % 4 == 0).
    """
    nx, ny, nzk = shape
    if nx % 2 or ny % 2:
        raise ValueError("First two dimensions of array must be even.")
    if packed:
        if nzk % 2 == 0:
            raise ValueError("Last dimension of packed array must be odd.")
        return nx, ny, 2 * (nzk - 1)
    if nzk % 2:
        raise ValueError("Last dimension of unpacked array must be even.")
    return nx, ny, nzk


def packed_shape(nx, ny, nz):
    """(nx, ny, nz//2+1) -- transform.py:192."""
    return nx, ny, nz // 2 + 1


def k_bounds(nx, ny, nz, spacing):
    """k_min, k_max of the grid -- powertools.py:16-24."""
    k0 = (2 * np.pi) / spacing
    return k0 / max(nx, ny, nz), k0 * np.sqrt(3) / 2


# --------------------------------------------------------------------------
# row K: |k| per cell
# --------------------------------------------------------------------------

def ksq_axes(nx, ny, nz, spacing):
    """Per-axis float64 tables of k_a(i)**2 -- powertools.py:27-37.

    ``np.fft.fftfreq(n, d)`` is ``integer_index * (1.0/(n*d))`` in float64;
    the packed z axis keeps the first nz//2+1 entries, whose last one is the
    *negative* Nyquist frequency (squared, so the sign is irrelevant).
    """
    lam = spacing / (2 * np.pi)
    out = []
    for n in (nx, ny, nz):
        idx = np.arange(n)
        idx[idx >= (n + 1) // 2] -= n  # fftfreq ordering: 0..n/2-1, -n/2..-1
        out.append((idx * (1.0 / (n * lam))) ** 2)
    out[2] = out[2][: nz // 2 + 1]
    return out


def fill_log10k(nx, ny, nz, spacing, dtype=np.complex64):
    """Packed array whose real part is log10|k| (imag 0) -- powertools.py:40-61.

    Dtype chain for complex64 (verified bit exact):
    t = f32(f64 kx2 + f64 ky2); t = f32(f64(t) + f64 kz2); t = log10_f32(t);
    t *= 0.5f.  The DC cell is log10(0) = -inf.
    """
    rt = np.zeros(0, dtype).real.dtype
    kx2, ky2, kz2 = ksq_axes(nx, ny, nz, spacing)
    t = (kx2[:, None, None] + ky2[None, :, None]).astype(rt)        # :50
    t = (t.astype(np.float64) + kz2[None, None, :]).astype(rt)       # :52
    with np.errstate(divide="ignore"):
        t = np.log10(t)                                              # :56 (array dtype)
    t *= rt.type(0.5)                                                # :60
    data = np.zeros(packed_shape(nx, ny, nz), dtype)
    data.real = t
    return data


# --------------------------------------------------------------------------
# row T: sigma(k) lookup
# --------------------------------------------------------------------------

def sigma_table(power_k, power_Pk, nx, ny, nz, spacing):
    """Host-side float64 tables (log10 k_i, sigma_i) -- powertools.py:139-154."""
    N3 = nx * ny * nz
    Vbox = N3 * spacing ** 3
    k = np.asarray(power_k, np.float64)
    Pk = np.asarray(power_Pk, np.float64)
    kmin, kmax = k_bounds(nx, ny, nz, spacing)
    if k.min() <= 0:
        raise ValueError("Power uses min(k) <= 0.")
    if k.min() > kmin or k.max() < kmax:                              # :147-150
        raise ValueError("Power k range does not cover data k range.")
    return np.log10(k), N3 * np.sqrt(Pk / (2 * Vbox))


def interp_sigma(x, xt, st):
    """Piecewise-linear sigma(x) in float64, 0 outside [xt[0], xt[-1]].

    Restates scipy's ``interp1d(kind='linear', bounds_error=False,
    fill_value=0)`` as called at powertools.py:155-157: for xt[j] <= x <
    xt[j+1] the value is ``slope*(x - xt[j]) + st[j]`` with
    ``slope = (st[j+1]-st[j])/(xt[j+1]-xt[j])``; x == xt[-1] gives st[-1];
    anything else (including -inf and NaN) gives 0.
    """
    x = np.asarray(x, np.float64)
    n = len(xt)
    j = np.searchsorted(xt, x, side="right") - 1
    inside = (x >= xt[0]) & (x <= xt[-1])
    j = np.clip(j, 0, n - 2)
    slope = (st[j + 1] - st[j]) / (xt[j + 1] - xt[j])
    y = slope * (np.where(inside, x, xt[0]) - xt[j]) + st[j]
    # np.interp returns fp[j] exactly on a knot and fp[-1] on the last one
    y = np.where(x == xt[-1], st[-1], y)
    return np.where(inside, y, 0.0)


def tabulate_sigmas(data, power_k, power_Pk, spacing):
    """Replace log10|k| in data.real by sigma -- powertools.py:125-164."""
    nx, ny, nz = expanded_shape(data.shape)
    xt, st = sigma_table(power_k, power_Pk, nx, ny, nz, spacing)
    data.real = interp_sigma(data.real, xt, st)      # f64 result rounded to array dtype (:163)
    return data


# --------------------------------------------------------------------------
# row R: noise
# --------------------------------------------------------------------------

def _mt19937_init(seed):
    mt = np.empty(624, np.uint32)
    s = np.uint64(seed & 0xFFFFFFFF)
    for i in range(624):
        mt[i] = np.uint32(s)
        s = (np.uint64(1812433253) * (s ^ (s >> np.uint64(30))) + np.uint64(i + 1)) & np.uint64(0xFFFFFFFF)
    return mt


def _mt19937_words(seed, nwords):
    """First ``nwords`` tempered outputs of MT19937 seeded with the Knuth LCG
    (numpy ``RandomState(int_seed)`` legacy seeding).  Pure-python loop: only
    for small pins."""
    mt = [int(v) for v in _mt19937_init(seed)]
    out = []
    pos = 624
    while len(out) < nwords:
        if pos == 624:
            for kk in range(624):
                y = (mt[kk] & 0x80000000) | (mt[(kk + 1) % 624] & 0x7FFFFFFF)
                v = mt[(kk + 397) % 624] ^ (y >> 1)
                if y & 1:
                    v ^= 0x9908B0DF
                mt[kk] = v
            pos = 0
        y = mt[pos]
        pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        out.append(y & 0xFFFFFFFF)
    return out


def legacy_normals(seed, n):
    """From-scratch restatement of ``RandomState(seed).normal(size=n)``
    (called at random.py:24,28): MT19937, 53-bit doubles from two words,
    Marsaglia polar method returning ``f*x2`` first and caching ``f*x1``.

    Each polar attempt consumes exactly 4 words.  Slow
    (python loop) -- used only to pin the stream definition on a few thousand
    values against numpy and the golden fixture.
    """
    # acceptance is pi/4, so 4 words/attempt * 1.4 attempts/pair is ample
    words = _mt19937_words(seed, int(4 * (n // 2 + 8) * 1.6) + 64)
    w = iter(words)

    def dbl():
        a = next(w) >> 5
        b = next(w) >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0

    out = []
    while len(out) < n:
        while True:
            x1 = 2.0 * dbl() - 1.0
            x2 = 2.0 * dbl() - 1.0
            r2 = x1 * x1 + x2 * x2
            if r2 < 1.0 and r2 != 0.0:
                break
        f = math.sqrt(-2.0 * math.log(r2) / r2)      # libm, as numpy's C legacy_gauss
        out.append(f * x2)
        out.append(f * x1)
    return np.array(out[:n], np.float64)


def reference_noise(seed, ncells):
    """The reference's float64 deviates for a packed array of ``ncells`` complex
    cells: 2*ncells values, interleaved (re, im) in C order -- random.py:17-28."""
    return np.random.RandomState(seed).normal(size=2 * ncells)


def randomize(data, seed=None, noise=None):
    """data.imag = data.real; every real component *= N(0,1) -- random.py:12-29.

    The product is taken in float64 and rounded once to the array dtype
    (in-place ``f32 *= f64`` in numpy).
    """
    data.imag = data.real
    rt = data.real.dtype
    flat = data.reshape(-1).view(rt)
    if noise is None:
        noise = reference_noise(seed, data.size)
    flat[:] = (flat.astype(np.float64) * noise).astype(rt)
    return data


# --------------------------------------------------------------------------
# row S: Hermitian symmetrisation of the kz = 0 and kz = nz/2 planes
# --------------------------------------------------------------------------

def _sym_roles(nx, ny):
    """Classify the cells of one kz in {0, nz/2} plane -- transform.py:141-158.

    Returns (self_conj, dest) boolean (nx, ny) masks.  A destination cell takes
    conj(value at ((-ix) % nx, (-iy) % ny)); everything else is a source.
    """
    ix = np.arange(nx)[:, None]
    iy = np.arange(ny)[None, :]
    x_edge = (ix == 0) | (ix == nx // 2)
    y_edge = (iy == 0) | (iy == ny // 2)
    self_conj = x_edge & y_edge
    dest = (iy > ny // 2) | (y_edge & (ix > nx // 2))   # :141-142,146-147 | :150-151
    dest = np.broadcast_to(dest, (nx, ny)) & ~self_conj
    return self_conj, dest


def symmetrize_packed(data):
    """In-place Hermitian symmetrisation of a packed array -- transform.py:114-121,141-158."""
    nx, ny, nz = expanded_shape(data.shape)
    self_conj, dest = _sym_roles(nx, ny)
    jx = (-np.arange(nx)) % nx
    jy = (-np.arange(ny)) % ny
    for iz in (0, nz // 2):
        plane = data[:, :, iz]
        mirrored = np.conj(plane[jx][:, jy])
        plane[dest] = mirrored[dest]
        plane.imag[self_conj] = 0
    data.real[0, 0, 0] = 0                                          # :158
    return data


def is_hermitian_packed(data, rtol=1e-8, atol=1e-8):
    """Checker restating transform.py:93-111 for packed=True."""
    nx, ny, nz = expanded_shape(data.shape)
    jx = (-np.arange(nx // 2 + 1)) % nx
    jy = (-np.arange(ny // 2 + 1)) % ny
    for iz in (0, nz // 2):
        a = data[: nx // 2 + 1, : ny // 2 + 1, iz]
        b = np.conj(data[jx][:, jy, iz])
        if not np.allclose(a, b, rtol, atol):
            return False
    return True


# --------------------------------------------------------------------------
# row X: 3-D c2r / r2c (numpy normalisation) -- transform.py:303-315
# --------------------------------------------------------------------------

def c2r(data, double_fft=False):
    """delta = irfftn(data) with the numpy normalisation (1/N3 on the inverse).

    double_fft=False: what the reference does under numpy >= 2 (single
    precision pocketfft for complex64).  double_fft=True: transform in float64
    and round once to the array's real dtype (numpy 1.x behaviour, and also a
    tighter truth for error budgeting)."""
    nx, ny, nz = expanded_shape(data.shape)
    rt = data.real.dtype
    src = data.astype(np.complex128) if double_fft else data
    return np.fft.irfftn(src, s=(nx, ny, nz), axes=(0, 1, 2)).astype(rt)


def r2c(field):
    """rfftn with numpy normalisation (forward unnormalised) -- transform.py:270."""
    ct = np.result_type(field.dtype, np.complex64)
    return np.fft.rfftn(field, axes=(0, 1, 2)).astype(ct)


# --------------------------------------------------------------------------
# row G: the whole path
# --------------------------------------------------------------------------

def generate_kspace(nx, ny, nz, spacing, power_k, power_Pk, seed=None, noise=None,
                    dtype=np.complex64):
    """Steps 1-5 of generate.py:191-199 -> symmetrised k-space array."""
    data = fill_log10k(nx, ny, nz, spacing, dtype)
    tabulate_sigmas(data, power_k, power_Pk, spacing)
    randomize(data, seed=seed, noise=noise)
    symmetrize_packed(data)
    return data


def generate_delta_field(nx, ny, nz, spacing, power_k, power_Pk, seed=None, noise=None,
                         dtype=np.complex64, double_fft=False):
    """generate.py:191-199,218-219 with save_potential=False.

    Returns (delta, rms) with rms = np.std(delta) (population std, array dtype)."""
    data = generate_kspace(nx, ny, nz, spacing, power_k, power_Pk, seed, noise, dtype)
    delta = c2r(data, double_fft=double_fft)
    return delta, np.std(delta.reshape(-1))


def potential_kspace(data, spacing):
    """delta(k)/k**2 with 0 at DC -- generate.py:200-217 (save_potential=True)."""
    nx, ny, nz = expanded_shape(data.shape)
    rt = data.real.dtype
    kx2, ky2, kz2 = ksq_axes(nx, ny, nz, spacing)
    t = (kx2[:, None, None] + ky2[None, :, None]).astype(rt)        # :207
    t = (t.astype(np.float64) + kz2[None, None, :]).astype(rt)       # :208
    with np.errstate(divide="ignore"):
        t = np.reciprocal(t)                                         # :211
    pot = np.zeros_like(data)
    pot.real = t
    pot[0, 0, 0] = 0                                                 # :213
    pot *= data                                                      # :215
    return pot


# --------------------------------------------------------------------------
# lensing potential: generate.py:352-416
# --------------------------------------------------------------------------

def _basic_simps(y, start, stop, x):
    """Composite Simpson sum over the interval pairs [start, stop] of the last axis for samples at
    (possibly non-uniform) x -- the `_basic_simps` helper of scipy.integrate (scipy <= 1.10; the reference
    calls `scipy.integrate.simps`, removed from current scipy, whose published algorithm is restated here)."""
    h = np.diff(x)
    s0, s1, s2 = slice(start, stop, 2), slice(start + 1, stop + 1, 2), slice(start + 2, stop + 2, 2)
    h0, h1 = h[s0], h[s1]
    hsum, hprod, h0divh1 = h0 + h1, h0 * h1, h0 / h1
    tmp = hsum / 6.0 * (y[..., s0] * (2 - 1.0 / h0divh1) + y[..., s1] * hsum * hsum / hprod + y[..., s2] * (2 - h0divh1))
    return np.sum(tmp, axis=-1)


def simps_avg(y, x):
    """scipy.integrate.simps(y, x, axis=-1, even='avg') as shipped when the reference was written: for an odd
    number of samples the composite Simpson rule; for an even number the average of (Simpson on the first N-1
    samples + trapezoid on the last interval) and (trapezoid on the first interval + Simpson on the last N-1)."""
    y = np.asarray(y)
    x = np.asarray(x, np.float64)
    N = y.shape[-1]
    if N % 2 == 0:
        val = 0.5 * (x[-1] - x[-2]) * (y[..., -1] + y[..., -2])
        result = _basic_simps(y, 0, N - 3, x)
        val = val + 0.5 * (x[1] - x[0]) * (y[..., 1] + y[..., 0])
        result = result + _basic_simps(y, 1, N - 2, x)
        return result / 2.0 + val / 2.0
    return _basic_simps(y, 0, N - 2, x)


def cot_k(DC, DA, K=0.0):
    """cotK(D) along the line of sight -- generate.py:383-395 (K in (Mpc/h)**-2; K = 0: flat)."""
    DC = np.asarray(DC, np.float64)
    DA = np.asarray(DA, np.float64)
    if K < 0:
        cosK = np.cosh(np.sqrt(-K) * DC)
    elif K > 0:
        cosK = np.cos(np.sqrt(K) * DC)
    else:
        cosK = np.ones_like(DA, dtype=float)
    cotK = np.ones_like(cosK)
    cotK[1:] = cosK[1:] / DA[1:]
    return cotK


def lensing_potential(dPhi, DC, DA, K=0.0, i_min=None):
    """psi(r) = integral of -2 [cotK(D) - cotK(D_src)] dPhi along z with the reference's slice loop and
    Simpson rule -- generate.py:397-411.  dPhi: real (nx, ny, nz) array; returns a new array of its dtype."""
    nDC = len(DC)
    if i_min is None:
        i_min = nDC // 32
    if i_min < 0 or i_min >= nDC:
        raise ValueError("Invalid i_min {}. Expected 0 - {}.".format(i_min, nDC - 1))
    cotK = cot_k(DC, DA, K)
    psi = np.empty_like(dPhi)
    for i in range(nDC, i_min, -1):
        psi[:, :, i_min:i] = dPhi[:, :, i_min:i]
        psi[:, :, i_min:i] *= -2 * (cotK[i_min:i] - cotK[i - 1])
        psi[:, :, i - 1] = simps_avg(psi[:, :, i_min:i], np.asarray(DC, np.float64)[i_min:i])
    if i_min > 0:
        psi[:, :, :i_min] = 0.0
    return psi


# --------------------------------------------------------------------------
# row L: lognormal map and per-z scaling
# --------------------------------------------------------------------------

def lognormal(delta, growth, sigma=None):
    """In-place lognormal transform -- cosmotools.py:206-221.

    Four sequential in-place sweeps, each rounded to the array dtype."""
    if sigma is None:
        sigma = np.std(delta)
    t = 1 + (sigma * growth) ** 2
    delta /= sigma
    delta *= np.sqrt(np.log(t))
    np.exp(delta, out=delta)
    delta /= np.sqrt(t)
    return delta


def scale_z(delta, factor):
    """delta *= factor broadcast along z -- generate.py:273."""
    delta *= factor
    return delta


# --------------------------------------------------------------------------
# native counter-based RNG of the HIP path (this repo's own definition,
# restated here so the GPU's native mode can be checked value by value)
# --------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85


# rounds of the native stream: Philox4x32-7, the smallest round count Salmon et al. (SC'11, table 2) found
# Crush-resistant (it passes BigCrush); Random123's default of 10 adds a safety margin the field generator
# does not need and the VALU-bound generation pass pays for (rf_core.h RF_PHILOX_ROUNDS)
NATIVE_PHILOX_ROUNDS = 7


def philox4x32_10(counter_lo, counter_hi, key0, key1):
    """Philox4x32-10 (the Random123 default), kept for the known-answer vectors."""
    return philox4x32(counter_lo, counter_hi, key0, key1, rounds=10)


def philox4x32(counter_lo, counter_hi, key0, key1, rounds=NATIVE_PHILOX_ROUNDS):
    """Philox4x32-R (Salmon et al. 2011) on arrays of 64-bit counters.

    counter words = (lo32(counter_lo), hi32(counter_lo), lo32(counter_hi),
    hi32(counter_hi)); returns four uint32 arrays."""
    c = np.asarray(counter_lo, np.uint64)
    ch = np.broadcast_to(np.asarray(counter_hi, np.uint64), c.shape)
    m32 = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    c0, c1 = c & m32, c >> s32
    c2, c3 = ch & m32, ch >> s32
    k0, k1 = int(key0) & 0xFFFFFFFF, int(key1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        n0 = (p1 >> s32) ^ c1 ^ np.uint64(k0)
        n1 = p1 & m32
        n2 = (p0 >> s32) ^ c3 ^ np.uint64(k1)
        n3 = p0 & m32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def philox_normals(seed, cell_index, bits=32):
    """(re, im) float64 standard normals of the HIP path's native RNG for the
    given 64-bit noise-cell indices ("native noise").

    One Philox call serves the cell *pair* ``cell_index >> 1``; the even cell
    uses words (0, 1), the odd cell words (2, 3).  With ``bits`` = 32 (float64
    plans) u = (w + 0.5) / 2**32 in float64.  With ``bits`` = 24 (float32 plans,
    24-bit significands) the kernels convert each word to float32 (round to
    nearest even) and form u1 = fma(float32(w), 2**-32, 2**-33) with ONE rounding
    (so the radius keeps its resolution in the tail, down to u1 = 2**-33); the
    angle uses the top 23 bits of its word, u2 = (w >> 9) / 2**23.  Then
    r = sqrt(-2 ln u1), (re, im) = r * (cos, sin)(2 pi u2) (BoxMuller<float> in
    rf_core.h).
    """
    ci = np.asarray(cell_index, np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    w = philox4x32(ci >> np.uint64(1), np.uint64(0), seed & 0xFFFFFFFF, seed >> 32)
    odd = (ci & np.uint64(1)).astype(bool)
    wa = np.where(odd, w[2], w[0])
    wb = np.where(odd, w[3], w[1])
    if bits == 32:
        u1 = (wa.astype(np.float64) + 0.5) / float(2 ** 32)
        u2 = (wb.astype(np.float64) + 0.5) / float(2 ** 32)
    elif bits == 24:
        fa = wa.astype(np.float32).astype(np.float64)            # v_cvt_f32_u32
        # the float64 expression is exact (<= 34 significant bits), so one rounding to float32 = the fma
        u1 = (fa * 2.0 ** -32 + 2.0 ** -33).astype(np.float32).astype(np.float64)
        u2 = (wb >> np.uint32(9)).astype(np.float64) * 2.0 ** -23   # exact
    else:
        raise ValueError("bits must be 24 or 32")
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)


def native_noise_index(nx, ny, nz):
    """Noise-cell index of every API cell (ix, iy, iz), shape (nx, ny, nz//2+1):
    cells with iz < nz/2 are numbered in the device-internal order
    [ix][iy][nz/2]; the Nyquist plane iz = nz/2 follows (rf_core.h)."""
    nzc = nz // 2
    col = (np.arange(nx, dtype=np.uint64)[:, None] * np.uint64(ny) + np.arange(ny, dtype=np.uint64)[None, :])
    idx = np.empty((nx, ny, nzc + 1), np.uint64)
    idx[:, :, :nzc] = col[:, :, None] * np.uint64(nzc) + np.arange(nzc, dtype=np.uint64)[None, None, :]
    idx[:, :, nzc] = np.uint64(nx * ny * nzc) + col
    return idx


def native_noise(seed, nx, ny, nz, dtype=np.complex64):
    """The HIP path's native deviates laid out like the reference's noise vector
    (2*M float64 values, (re, im) interleaved in C order of the packed array)."""
    bits = 24 if np.dtype(dtype) == np.complex64 else 32
    re, im = philox_normals(seed, native_noise_index(nx, ny, nz), bits=bits)
    if bits == 24:  # the float32 kernels form the deviates in float32
        re, im = re.astype(np.float32).astype(np.float64), im.astype(np.float32).astype(np.float64)
    return np.stack([re, im], axis=-1).reshape(-1)


def default_like_power(nrows=500, amplitude=2.0e4):
    """Synthetic smooth P(k) used for throughput runs: 500 rows,
    k log-spaced 1e-4..22 h/Mpc, P(k) = A k / (1 + (k/0.02)^2)^1.8."""
    k = np.logspace(-4, np.log10(22.0), nrows)
    return k, amplitude * k / (1 + (k / 0.02) ** 2) ** 1.8
