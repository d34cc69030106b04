"""ctypes binding of the CPU kernel emulator (randomfield_amd/csrc/emu) -- test tooling."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "randomfield_amd", "csrc")
SO = os.path.join(CSRC, "emu", "librf_emu.so")

_c_dp = ctypes.POINTER(ctypes.c_double)


def _dp(a):
    return a.ctypes.data_as(_c_dp) if a is not None else None


def build(force=False):
    srcs = [os.path.join(CSRC, f) for f in ("emu/rf_emu.cpp", "rf_core.h", "rf_fft.h", "rf_configs.h", "rf_host.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, srcs[0]])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _gen_args(nx, ny, nz, spacing, log10k, sigma, seed, noise):
    from oracle import cpu_ref
    kx2, ky2, kz2 = (np.ascontiguousarray(a) for a in cpu_ref.ksq_axes(nx, ny, nz, spacing))
    log10k = np.ascontiguousarray(log10k, np.float64)
    sigma = np.ascontiguousarray(sigma, np.float64)
    mode = 0 if noise is None else 1
    if noise is not None:
        noise = np.ascontiguousarray(noise, np.float64)
    keep = (kx2, ky2, kz2, log10k, sigma, noise)
    args = (_dp(kx2), _dp(ky2), _dp(kz2), _dp(log10k), _dp(sigma), ctypes.c_int(len(log10k)),
            ctypes.c_int(mode), ctypes.c_uint64(seed or 0), _dp(noise))
    return args, keep


def generate_kspace(nx, ny, nz, spacing, log10k, sigma, seed=0, noise=None, dtype=np.complex64):
    args, keep = _gen_args(nx, ny, nz, spacing, log10k, sigma, seed, noise)
    out = np.empty((nx, ny, nz // 2 + 1), dtype)
    rc = lib().emu_generate_kspace(int(dtype == np.complex128), nx, ny, nz, *args, out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out


def realise(nx, ny, nz, spacing, log10k, sigma, seed=0, noise=None, dtype=np.complex64):
    args, keep = _gen_args(nx, ny, nz, spacing, log10k, sigma, seed, noise)
    rt = np.float32 if dtype == np.complex64 else np.float64
    out = np.empty((nx, ny, nz), rt)
    s1, s2 = ctypes.c_double(), ctypes.c_double()
    rc = lib().emu_realise(int(dtype == np.complex128), nx, ny, nz, *args, out.ctypes.data_as(ctypes.c_void_p),
                           ctypes.byref(s1), ctypes.byref(s2))
    assert rc == 0, rc
    return out, s1.value, s2.value


def c2r(kspace):
    nx, ny, nzh = kspace.shape
    nz = 2 * (nzh - 1)
    rt = np.float32 if kspace.dtype == np.complex64 else np.float64
    out = np.empty((nx, ny, nz), rt)
    s1, s2 = ctypes.c_double(), ctypes.c_double()
    ks = np.ascontiguousarray(kspace)
    rc = lib().emu_c2r(int(kspace.dtype == np.complex128), nx, ny, nz, ks.ctypes.data_as(ctypes.c_void_p),
                       out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(s1), ctypes.byref(s2))
    assert rc == 0, rc
    return out, s1.value, s2.value


def col_fft(data, N, direction, ncols, inner, outer_stride, row_stride):
    f = lib().emu_col_fft
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_longlong] * 4
    rc = f(int(data.dtype == np.complex128), N, direction, data.ctypes.data_as(ctypes.c_void_p), ncols, inner,
           outer_stride, row_stride)
    return rc


def realise_fast(nx, ny, nz, spacing, log10k, sigma, seed, dtype=np.float32):
    """Fused realisation with the fast native generation path (float32 arithmetic; a float64
    plan widens the generated values and transforms in double precision)."""
    args, keep = _gen_args(nx, ny, nz, spacing, log10k, sigma, seed, None)
    k0 = 2 * np.pi / spacing
    xlo = np.log10(k0 / max(nx, ny, nz)) - 0.01
    xhi = np.log10(k0 * np.sqrt(3) / 2) + 0.01
    out = np.empty((nx, ny, nz), dtype)
    s1, s2 = ctypes.c_double(), ctypes.c_double()
    rc = lib().emu_realise_fast(int(np.dtype(dtype) == np.float64), nx, ny, nz, *args[:6], ctypes.c_uint64(seed),
                                ctypes.c_double(xlo), ctypes.c_double(xhi), ctypes.c_double(k0 / nx),
                                out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(s1), ctypes.byref(s2))
    assert rc == 0, rc
    return out, s1.value, s2.value


def c2c(data, inverse):
    """Unpacked complex-to-complex transform through the emulated kernels (returns a new array)."""
    out = np.ascontiguousarray(data).copy()
    nx, ny, nz = out.shape
    rc = lib().emu_c2c(int(out.dtype == np.complex128), nx, ny, nz, 1 if inverse else -1,
                       out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return out


def r2c(field):
    nx, ny, nz = field.shape
    ct = np.complex64 if field.dtype == np.float32 else np.complex128
    out = np.empty((nx, ny, nz // 2 + 1), ct)
    f = np.ascontiguousarray(field)
    rc = lib().emu_r2c(int(ct == np.complex128), nx, ny, nz, f.ctypes.data_as(ctypes.c_void_p),
                       out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return out


def fast_sigma(log10k, sigma, xlo, xhi, k2):
    """fast float32 sigma lookup and the exact float64 interpolation for an array of |k|^2 values; also the number
    of per-bin records the fast table needed"""
    log10k = np.ascontiguousarray(log10k, np.float64)
    sigma = np.ascontiguousarray(sigma, np.float64)
    k2 = np.ascontiguousarray(k2, np.float32)
    fast = np.empty(k2.size, np.float32)
    exact = np.empty(k2.size, np.float64)
    f = lib().emu_fast_sigma
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    nb = f(log10k.ctypes.data, sigma.ctypes.data, len(log10k), xlo, xhi, k2.ctypes.data, k2.size, fast.ctypes.data,
           exact.ctypes.data)
    assert nb > 0, nb
    return fast, exact, nb


# ---- the generic (non-power-of-two) kernels of csrc/rf_generic.h ----------------------------------
def generic_c2r(kspace):
    nx, ny, nzh = kspace.shape
    nz = 2 * (nzh - 1)
    rt = np.float32 if kspace.dtype == np.complex64 else np.float64
    out = np.empty((nx, ny, nz), rt)
    s1, s2 = ctypes.c_double(), ctypes.c_double()
    ks = np.ascontiguousarray(kspace)
    rc = lib().emu_generic_c2r(int(kspace.dtype == np.complex128), nx, ny, nz, ks.ctypes.data_as(ctypes.c_void_p),
                               out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(s1), ctypes.byref(s2))
    assert rc == 0, rc
    return out, s1.value, s2.value


def generic_r2c(field):
    nx, ny, nz = field.shape
    ct = np.complex64 if field.dtype == np.float32 else np.complex128
    out = np.empty((nx, ny, nz // 2 + 1), ct)
    f = np.ascontiguousarray(field)
    rc = lib().emu_generic_r2c(int(ct == np.complex128), nx, ny, nz, f.ctypes.data_as(ctypes.c_void_p),
                               out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return out


def generic_c2c(data, inverse):
    out = np.ascontiguousarray(data).copy()
    nx, ny, nz = out.shape
    rc = lib().emu_generic_c2c(int(out.dtype == np.complex128), nx, ny, nz, 1 if inverse else -1,
                               out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return out
