#!/usr/bin/env python3
"""GPU sweep over every supported axis length on every axis (float32 and float64): fast vs exact generation flavour,
native batch vs single realisation, r2c -> c2r and c2c round trips.  Exercises every kernel instantiation."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools      # noqa: E402

POWER = powertools.load_default_power()
COL = [8, 16, 32, 64, 128, 256, 512, 1024, 2048]
ROWZ = [16, 32, 64, 128, 256, 512, 1024, 2048]


def check(shape, ct):
    nx, ny, nz = shape
    rt = np.float32 if ct == np.complex64 else np.float64
    print('  plan', shape, np.dtype(ct).name, flush=True)
    p = _hip.DevicePlan(nx, ny, nz, ct)
    p.set_kgrid(*powertools.ksq_axes(nx, ny, nz, 2.5))
    p.set_power(*powertools.sigma_table(POWER, shape, 2.5))
    p.realise(seed=11)
    a = p.download_real()
    std = p.moments()[1]
    p.set_exact_generation(True)
    p.realise(seed=11)
    b = p.download_real()
    p.set_exact_generation(False)
    assert np.isfinite(a).all() and std > 0
    assert np.max(np.abs(a - b)) <= 3e-5 * std, (shape, ct, np.max(np.abs(a - b)) / std)
    p.realise_batch(np.array([4, 11], dtype=np.uint64), want_rms=False)
    assert np.array_equal(p.download_real(), a)
    p.execute_r2c()
    p.execute_c2r()
    c = p.download_real()
    assert np.max(np.abs(c - a)) <= (2e-5 if rt == np.float32 else 1e-11) * std, (shape, ct, "r2c round trip")
    p.close()
    if nz <= 1024:
        q = _hip.DevicePlan(nx, ny, nz, ct, unpacked=True)
        z = (a[..., ::1] + 1j * np.roll(a, 1, axis=0)).astype(ct)
        q.upload_c(z)
        q.execute_c2c(False)
        q.execute_c2c(True)
        assert np.max(np.abs(q.download_c() - z)) <= (3e-5 if rt == np.float32 else 1e-11) * std, (shape, ct, "c2c")
        q.close()


def main():
    shapes = set()
    for n in COL:
        shapes.add((n, 16, 32)); shapes.add((16, n, 32)); shapes.add((n, n, 16))
    for n in ROWZ:
        shapes.add((8, 16, n)); shapes.add((32, 8, n))
    shapes |= {(64, 32, 2048), (2048, 8, 1024), (8, 2048, 256), (256, 256, 256), (512, 64, 128), (128, 1024, 64)}
    n = 0
    for shape in sorted(shapes):
        for ct in (np.complex64, np.complex128):
            check(shape, ct)
            n += 1
        print("ok", shape, flush=True)
    print("size sweep ok: %d plans" % n)


if __name__ == "__main__":
    main()
