#!/usr/bin/env python3
"""Timing of the generic (non-power-of-two) path: realisations on grids such as 1000^3 against the tiled 1024^3.
usage: generic_bench.py [--lib variant.so] [edge | NXxNYxNZ ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools      # noqa: E402

POWER = powertools.load_default_power()


def run(shape, ct=np.complex64, reps=3):
    nx, ny, nz = shape
    p = _hip.DevicePlan(nx, ny, nz, ct)
    p.set_kgrid(*powertools.ksq_axes(nx, ny, nz, 2.5))
    p.set_power(*powertools.sigma_table(POWER, shape, 2.5))
    p.realise(seed=1)
    p.sync()
    t0 = time.perf_counter()
    for i in range(reps):
        p.realise(seed=2 + i)
    p.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    k = p.kernel_ms() if p.tiled else None
    std = p.moments()[1]
    p.close()
    cells = float(nx) * ny * nz
    print("%-20s %-10s %s  %9.3f ms  %9.1f Mcells/s  rms %.4f %s" % (shape, np.dtype(ct).name, "tiled  " if k is not None else "generic", ms, cells / ms / 1e3, std,
                                                                "" if k is None else "kernels %s" % np.round(k, 3)), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--lib":
        _hip.LIB_PATH = os.path.abspath(args[1])
        args = args[2:]
    for a in args or ["500", "512", "1000", "1024"]:
        run(tuple(int(v) for v in a.split("x")) if "x" in a else (int(a),) * 3)
