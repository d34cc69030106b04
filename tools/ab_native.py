#!/usr/bin/env python3
"""A/B of library builds on the bench line's own workload (1024^3 float32, native generator): graph-replayed batches interleaved
between the builds in ONE process is not possible (one library per process), so: python3 tools/ab_native.py <lib or -> [steps]
prints ms per realisation of 3 batches + the eager per-pass event times."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
plan.realise_batch_prepare(steps)
plan.realise_batch(np.arange(5, dtype=np.uint64), want_rms=False)
plan.sync()
ts = []
for r in range(3):
    t0 = time.perf_counter()
    plan.realise_batch(np.arange(100 * r, 100 * r + steps, dtype=np.uint64), want_rms=False)
    plan.sync()
    ts.append((time.perf_counter() - t0) / steps * 1e3)
kern = np.zeros(5)
for i in range(5):
    plan.realise(seed=7 + i)
    plan.sync()
    kern += np.array(plan.kernel_ms())
print(json.dumps({"lib": os.path.basename(_hip.LIB_PATH), "ms_per_step": [round(t, 4) for t in ts],
                  "kernel_ms[x,y,z,reduce,x_kz0]": [round(float(v), 4) for v in kern / 5], "rms": round(plan.moments()[1], 6)}), flush=True)
plan.close()
