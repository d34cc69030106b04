#!/usr/bin/env python3
"""A/B of the same-seed path (rng='reference', 1024^3 float32): python3 tools/ref_ab.py [variant library]
cases: default | the blocked intermediate (RF_FLAG_TRANSPOSED_INTERMEDIATE) | segments twice as long (half as many jumps, two
waves per SIMD in the replay's generation pass)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, mt19937, powertools   # noqa: E402

if len(sys.argv) > 1:
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
n = 1024
power = powertools.load_default_power()
ncells = n * n * (n // 2 + 1)
base_bps = mt19937.segment_blocks_for(ncells)
for name, xposed, bps in (("default", False, None), ("blocked intermediate", True, None), ("segments x2", False, 2 * base_bps),
                          ("segments x1.33", False, -(-4 * base_bps // 3))):
    plan = _hip.DevicePlan(n, n, n, np.complex64)
    plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
    plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    if xposed:
        plan.set_transposed_intermediate(True)
    if bps:
        t0 = time.perf_counter()
        plan.set_mt_segment_blocks(bps)
    rep, tot, kern = [], [], np.zeros(5)
    for i in range(5):
        plan.sync()
        t0 = time.perf_counter()
        plan.reference_noise(100 + i, single=True)
        plan.sync()
        t1 = time.perf_counter()
        plan.realise(noise="resident")
        plan.sync()
        t2 = time.perf_counter()
        if i:
            rep.append(t1 - t0)
            tot.append(t2 - t0)
            kern += np.array(plan.kernel_ms())
    print(json.dumps({"case": name, "bps": bps or base_bps, "replay_ms": round(float(np.median(rep)) * 1e3, 3),
                      "total_ms": round(float(np.median(tot)) * 1e3, 3), "kernel_ms[x,y,z,reduce,x_kz0]": [round(float(v), 3) for v in kern / 4],
                      "rms": round(plan.moments()[1], 6)}), flush=True)
    plan.close()
