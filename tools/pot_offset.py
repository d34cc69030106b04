#!/usr/bin/env python3
"""The default call (rf_realise_potential: two store streams) against the offset of the potential array inside its allocation
(RF_POT_OFFSET), in three allocation histories: how do the field's and the potential's store streams share the memory channels?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

os.environ["RANDOMFIELD_DEBUG"] = "1"
n = 1024
power = powertools.load_default_power()
tables = powertools.sigma_table(power, (n, n, n), 2.5)
axes = powertools.ksq_axes(n, n, n, 2.5)


def case(off):
    if off is None:
        os.environ.pop("RF_POT_OFFSET", None)
    else:
        os.environ["RF_POT_OFFSET"] = str(off)
    plan = _hip.DevicePlan(n, n, n, np.complex64)
    plan.set_kgrid(*axes)
    plan.set_power(*tables)
    for i in range(3):
        plan.realise_potential(seed=i)
    plan.sync()
    ts = []
    for i in range(5):
        t0 = time.perf_counter()
        plan.realise_potential(seed=10 + i)
        plan.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    plan.close()
    return float(np.median(ts))


offs = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1 << 20, 2 << 20, 3 << 20, (4 << 20) - 2048, None]
for history in ("fresh", "after f64 plan", "after mt plan"):
    if history == "after f64 plan":
        p = _hip.DevicePlan(n, n, n, np.complex128)
        p.set_kgrid(*axes); p.set_power(*tables); p.realise(seed=1); p.sync(); p.close()
    if history == "after mt plan":
        p = _hip.DevicePlan(n, n, n, np.complex64)
        p.set_kgrid(*axes); p.set_power(*tables); p.reference_noise(5, single=True); p.realise(noise="resident"); p.sync(); p.close()
    for off in offs:
        print("%-15s RF_POT_OFFSET %-8s default call %.3f ms" % (history, off, case(off)), flush=True)
