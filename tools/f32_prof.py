#!/usr/bin/env python3
"""per-kernel times of float32 realisations with a variant build of the library (kernel experiments)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if len(sys.argv) > 2:
    _hip.LIB_PATH = os.path.abspath(sys.argv[2])
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
for i in range(6):
    plan.realise(seed=100 + i)
    plan.sync()
    print("kernel_ms", [round(v, 3) for v in plan.kernel_ms()], "total", round(plan.elapsed_ms(), 3))
import time
seeds = np.arange(10, dtype=np.uint64)
plan.realise_batch(seeds)
t0 = time.perf_counter()
plan.realise_batch(seeds, want_rms=False)
plan.sync()
print("batch ms/realisation", round((time.perf_counter() - t0) * 100, 4))
plan.close()
