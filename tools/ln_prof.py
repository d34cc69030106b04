#!/usr/bin/env python3
"""fused against unfused float64 realisation + lognormal map (BASELINE config 5): wall times, for rocprofv3 --kernel-trace --stats too"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, cosmotools, powertools   # noqa: E402

if os.environ.get("RF_LIB"):
    _hip.LIB_PATH = os.path.abspath(os.environ["RF_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dt = np.complex64 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.complex128
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, dt)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
growth = np.exp(-0.5 * np.arange(n) / n)
plan.set_z_tables(growth)


def unfused(seed):
    plan.realise(seed=seed)
    mean, std = plan.moments()
    a_z, b_z = cosmotools.lognormal_tables(growth, std, n)
    plan.lognormal(a_z, b_z, std)


for name, fn in (("plain realisation", lambda s: plan.realise(seed=s)), ("unfused", unfused),
                 ("fused", lambda s: plan.realise_lognormal(seed=s, want_sigma=False))):
    fn(1)
    plan.sync()
    ts = []
    for i in range(5):
        t0 = time.perf_counter()
        fn(10 + i)
        plan.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s %s %d^3: %s ms" % (name, np.dtype(dt).name, n, " ".join("%.3f" % t for t in ts)), flush=True)
    if name != "unfused":
        print("   kernel_ms (x, y, z, reduce):", " ".join("%.3f" % t for t in plan.kernel_ms()), flush=True)
plan.close()
