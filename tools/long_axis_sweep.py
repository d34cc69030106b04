#!/usr/bin/env python3
"""GPU sweep of the four-step form: axis lengths beyond one line of the LDS, on every axis, both dtypes --
packed c2r against numpy's irfftn, the r2c reverse plan, unpacked c2c both ways.  usage: long_axis_sweep.py [quick]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip                   # noqa: E402
from randomfield_amd.transform import Plan         # noqa: E402

LONG = {np.complex64: [8194, 8196, 8200, 9000, 10000, 12288, 15000, 16384, 20000, 24576, 30030, 32768, 50000, 65536, 100000, 131072,
                       250000, 524288, 1048576],
        np.complex128: [4098, 4100, 5000, 6144, 8192, 10000, 16384, 30030, 65536, 250000, 1048576]}


def check(shape, ct):
    rng = np.random.RandomState(sum(shape))
    nx, ny, nz = shape
    tol = 2e-6 if ct == np.complex64 else 1e-14
    plan = Plan(shape=shape, dtype_in=ct)
    assert plan.backend == "hip" and not plan.device.tiled
    rt = plan.data_out.dtype
    plan.data_in.view(rt).reshape(nx, ny, nz + 2)[:] = rng.normal(size=(nx, ny, nz + 2))
    ks = plan.data_in.copy()
    out = plan.execute().copy()
    ref = np.fft.irfftn(ks.astype(np.complex128), s=shape, axes=(0, 1, 2))
    e1 = np.max(np.abs(out - ref)) / ref.std()
    back = plan.create_reverse_plan().execute()
    kref = np.fft.rfftn(ref)
    e2 = np.max(np.abs(back - kref)) / np.abs(kref).std()
    plan.device.close()
    e3 = 0.0
    for inverse, fn in ((True, np.fft.ifftn), (False, np.fft.fftn)):
        c = Plan(shape=shape, dtype_in=ct, packed=False, inverse=inverse)
        c.data_in[:] = rng.normal(size=shape) + 1j * rng.normal(size=shape)
        src = c.data_in.copy()
        ref = fn(src.astype(np.complex128))
        e3 = max(e3, np.max(np.abs(c.execute() - ref)) / np.abs(ref).std())
        c.device.close()
    ok = e1 <= 40 * tol and e2 <= 80 * tol and e3 <= 40 * tol
    print("%-22s %-10s c2r %.2e  r2c %.2e  c2c %.2e  %s" % (shape, np.dtype(ct).name, e1, e2, e3, "ok" if ok else "FAIL"), flush=True)
    return ok


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    bad = n = 0
    for ct, lengths in LONG.items():
        for L in (lengths[::3] if quick else lengths):
            shapes = [(L, 2, 4), (2, L, 4)]
            if L % 4 == 0:
                shapes.append((2, 2, L))
                if _hip.shape_supported(2, 2, 2 * L, ct):
                    shapes.append((2, 2, 2 * L))             # (the packed plan's contiguous axis is long from nz / 2 > cap on)
            for s in shapes:
                if not _hip.shape_supported(*s, ct):
                    print("%-22s %-10s not supported (no two factors within the cap)" % (s, np.dtype(ct).name))
                    continue
                bad += 0 if check(s, ct) else 1
                n += 1
    print("long axis sweep: %d plans, %d failures" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
