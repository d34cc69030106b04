#!/usr/bin/env python3
"""rf_lensing_potential at 1024^3 (float32 and float64): ms per call (development tool)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
n = 1024
power = powertools.load_default_power()
for dt in (np.complex64, np.complex128):
    plan = _hip.DevicePlan(n, n, n, dt)
    plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
    plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    plan.realise(seed=3)
    cot = 1.0 / (2.5 * (1 + np.arange(n)))
    ts = []
    for i in range(6):
        plan.sync()
        t0 = time.perf_counter()
        plan.lensing_potential(cot, 2.5, 2)
        plan.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    psi = plan.download_aux(x0=7, x1=8)
    print("%s lensing 1024^3 %s: %s ms  checksum %.12e" % (sys.argv[1] if len(sys.argv) > 1 else "product", np.dtype(dt).name,
          " ".join("%.3f" % t for t in ts), float(psi.astype(np.float64).sum())), flush=True)
    plan.close()
