#!/usr/bin/env python3
"""per-kernel times of a float64 realisation (x main, y, z, reduce, x kz=0 tiles) + lognormal map"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools, cosmotools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if len(sys.argv) > 2:
    _hip.LIB_PATH = os.path.abspath(sys.argv[2])      # a variant build of the library (kernel experiments)
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex128)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
for i in range(4):
    plan.realise(seed=100 + i)
    plan.sync()
    print("kernel_ms", [round(v, 3) for v in plan.kernel_ms()], "total", round(plan.elapsed_ms(), 3))
mean, std = plan.moments()
g = np.exp(-0.5 * np.arange(n) / n)
import time
for i in range(3):
    plan.realise(seed=5)
    plan.sync()
    t0 = time.perf_counter()
    plan.lognormal(*cosmotools.lognormal_tables(g, std, n), std)
    plan.sync()
    print("lognormal ms", round((time.perf_counter() - t0) * 1e3, 3))
plan.close()
