#!/usr/bin/env python3
"""Batch time per realisation against the y/z slab size, several rounds (boxes drift).  usage: slab_scan.py n [f32|f64] B,B,... [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1])
dt = np.complex128 if sys.argv[2] == "f64" else np.complex64
Bs = [int(b) for b in sys.argv[3].split(",")]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
K = 20
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, dt)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
seeds = np.arange(K, dtype=np.uint64)
res = {b: [] for b in Bs}
for r in range(rounds):
    for B in Bs:
        plan.set_yz_slab_planes(B)
        plan.realise_batch(seeds[:5], want_rms=False)
        plan.sync()
        t0 = time.perf_counter()
        plan.realise_batch(seeds, want_rms=False)
        plan.sync()
        res[B].append((time.perf_counter() - t0) * 1e3 / K)
for B in Bs:
    print("n %d %s slab planes %4d  ms/realisation %s  min %.4f" % (n, np.dtype(dt).name, B, " ".join("%.4f" % v for v in res[B]), min(res[B])), flush=True)
plan.close()
