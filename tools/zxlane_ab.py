#!/usr/bin/env python3
"""A/B of the z pass's last exchange through the cross-lane network (RF_Z_XLANE build) against the LDS row image: field
identity and batch time.  usage: zxlane_ab.py variant.so   (run twice under rocprofv3 --pmc for the LDS counters)"""
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
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
seeds = np.arange(20, dtype=np.uint64)
plan.realise_batch(seeds[:5], want_rms=False)
ts = []
for r in range(3):
    plan.sync()
    t0 = time.perf_counter()
    plan.realise_batch(seeds, want_rms=False)
    plan.sync()
    ts.append((time.perf_counter() - t0) * 1e3 / len(seeds))
plan.realise(seed=3)
plan.sync()
f = plan.download_real(x0=5, x1=7)
print("lib %s: batch ms/realisation %s; eager kernel_ms %s; checksum %.10e %.10e" % (
    os.path.basename(_hip.LIB_PATH), " ".join("%.4f" % t for t in ts), [round(v, 3) for v in plan.kernel_ms()],
    float(f.astype(np.float64).sum()), float((f.astype(np.float64) ** 2).sum())), flush=True)
np.save("gpurun_out/zxlane_%s.npy" % os.path.basename(_hip.LIB_PATH), f[:, ::16, ::16])
plan.close()
