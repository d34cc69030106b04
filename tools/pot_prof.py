#!/usr/bin/env python3
"""native realise_potential at 1024^3 a few times (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
import time
for i in range(6):
    t0 = time.perf_counter()
    plan.realise_potential(seed=100 + i)
    plan.sync()
    print("realise_potential %.3f ms" % ((time.perf_counter() - t0) * 1e3))
for i in range(3):
    t0 = time.perf_counter()
    plan.realise(seed=100 + i)
    plan.sync()
    print("realise %.3f ms" % ((time.perf_counter() - t0) * 1e3))
print(plan.moments())
plan.close()
