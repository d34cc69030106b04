"""The kernel phase functions under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available
on this pool): every supported axis length of the strided passes, the c2r and r2c row passes and both
generation flavours, and the generic mixed-radix blocks, run in a separate process against an instrumented build of the emulator, whose LDS /
global "memory" are exactly-sized heap arrays -- any out-of-bounds index aborts the run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "randomfield_amd", "csrc")
SO = os.path.join(CSRC, "emu", "librf_emu_asan.so")

DRIVER = r'''
import ctypes, sys, os
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import emu_util
emu_util.SO = %(so)r
emu_util._lib = ctypes.CDLL(%(so)r)
from oracle import cpu_ref
pw = np.load(os.path.join(%(root)r, "tests", "golden", "default_power.npz"))
rng = np.random.RandomState(0)
for N in (8, 16, 32, 64, 128, 256, 512, 1024, 2048):
    for dt in (np.complex64, np.complex128):
        a = (rng.normal(size=(N, 64)) + 1j * rng.normal(size=(N, 64))).astype(dt)
        assert emu_util.col_fft(a, N, +1, 64, 64, 0, 64) == 0
for M in (8, 16, 32, 64, 128, 256, 512, 1024):
    for dt in (np.float32, np.float64):
        f = rng.normal(size=(8, 8, 2 * M)).astype(dt)
        spec = emu_util.r2c(f)
        back, s1, s2 = emu_util.c2r(spec)
        assert np.max(np.abs(back - f)) < 1e-3
for shape in ((8, 8, 16), (16, 32, 64), (64, 16, 32)):
    nx, ny, nz = shape
    xt, st = cpu_ref.sigma_table(pw["k"], pw["Pk"], nx, ny, nz, 2.5)
    noise = cpu_ref.reference_noise(1, nx * ny * (nz // 2 + 1))
    for dt in (np.complex64, np.complex128):
        emu_util.generate_kspace(nx, ny, nz, 2.5, xt, st, noise=noise, dtype=dt)
        emu_util.realise(nx, ny, nz, 2.5, xt, st, noise=noise, dtype=dt)
        emu_util.realise(nx, ny, nz, 2.5, xt, st, seed=3, dtype=dt)
    emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=3)
# the float32 generation pass of length 1024 on tile pairs (ColPair: parked registers, paired stores) and as two half transforms at 2048
for shape in ((1024, 8, 32), (2048, 8, 16)):
    nx, ny, nz = shape
    xt, st = cpu_ref.sigma_table(pw["k"], pw["Pk"], nx, ny, nz, 2.5)
    emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=5)
for shape in ((4, 6, 8), (40, 60, 80), (10, 14, 22), (2, 2, 2), (26, 34, 46)):        # the generic mixed-radix blocks
    for ct, rt in ((np.complex64, np.float32), (np.complex128, np.float64)):
        nx, ny, nz = shape
        ks = (rng.normal(size=(nx, ny, nz // 2 + 1)) + 1j * rng.normal(size=(nx, ny, nz // 2 + 1))).astype(ct)
        out, s1, s2 = emu_util.generic_c2r(ks)
        assert np.max(np.abs(out - np.fft.irfftn(ks.astype(np.complex128), s=shape, axes=(0, 1, 2)))) < 1e-3
        emu_util.generic_r2c(out)
        emu_util.generic_c2c(ks, True)
print("SANITIZED-OK")
'''


def test_emulator_under_asan_ubsan():
    src = os.path.join(CSRC, "emu", "rf_emu.cpp")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", SO, src])
    asan = subprocess.check_output(["g++", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", DRIVER % dict(root=ROOT, so=SO)], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0 and "SANITIZED-OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
