#!/usr/bin/env python3
"""realise_potential (the reference's default call, save_potential=True) at 1024^3 float64 and 2048^3 float32, a library variant
against the product (development tool).  usage: tools/pot_ab.py [variant.so | -]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
power = powertools.load_default_power()
for n, dt in ((1024, np.complex128), (2048, np.complex64)):
    plan = _hip.DevicePlan(n, n, n, dt)
    plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
    plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    for i in range(3):
        plan.realise_potential(seed=i)
    ts = []
    for i in range(4):
        plan.sync()
        t0 = time.perf_counter()
        plan.realise_potential(seed=10 + i)
        plan.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s  %d^3 %s realise_potential ms: %s  kernel_ms %s  std %.9f" % (sys.argv[1] if len(sys.argv) > 1 else "product", n, np.dtype(dt).name,
          " ".join("%.3f" % t for t in ts), np.round(plan.kernel_ms(), 3).tolist(), plan.moments()[1]), flush=True)
    plan.close()
