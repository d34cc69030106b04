#!/usr/bin/env python3
"""Batch time per realisation for several settings of the y/z slab size (RF_FLAG_YZ_SLAB_PLANES) and of the transposed
intermediate (RF_FLAG_TRANSPOSED_INTERMEDIATE), in one process.  usage: slab_ab.py n [f32|f64] [variant.so] [B,B,...] [nbatch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dt = np.complex128 if len(sys.argv) > 2 and sys.argv[2] == "f64" else np.complex64
if len(sys.argv) > 3 and sys.argv[3] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[3])
Bs = [int(b) for b in sys.argv[4].split(",")] if len(sys.argv) > 4 and sys.argv[4] != "-" else [0, -1, 32, 64, 128, 0]
K = int(sys.argv[5]) if len(sys.argv) > 5 else 10
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, dt)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
seeds = np.arange(K, dtype=np.uint64)
ref = None
for xp in (0, 1):
    plan.set_transposed_intermediate(bool(xp))
    for B in Bs:
        plan.set_yz_slab_planes(B)
        rms = plan.realise_batch(seeds)
        t0 = time.perf_counter()
        plan.realise_batch(seeds, want_rms=False)
        plan.sync()
        ms = (time.perf_counter() - t0) * 1e3 / K
        f = plan.download_real(x0=0, x1=1)
        if ref is None:
            ref = (f.copy(), rms.copy())
        plan.realise(seed=3)
        plan.sync()
        km = [round(v, 3) for v in plan.kernel_ms()]
        print("n %d %s xposed %d slab planes %4d  batch ms/realisation %.4f  identical %s  rms dev %.1e  eager kernel_ms %s" % (
            n, np.dtype(dt).name, xp, B, ms, np.array_equal(ref[0], f), np.max(np.abs(rms - ref[1])), km), flush=True)
plan.close()
