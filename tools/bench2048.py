#!/usr/bin/env python3
"""2048^3 float32 on one GPU (BASELINE config 4's per-GPU kernels at full axis length) and the per-rank slab
compute of the 8-GPU job with virtual ranks: per-kernel times (development tool)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])      # a variant build of the library (kernel experiments)
quick = len(sys.argv) > 2
n = 2048
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
plan.realise(seed=1)
plan.sync()
kern = np.zeros(5)
ts = []
for i in range(4):
    plan.sync()
    t0 = time.perf_counter()
    plan.realise(seed=10 + i)
    plan.sync()
    ts.append(time.perf_counter() - t0)
    kern += np.array(plan.kernel_ms())
t = float(np.median(ts))
sweep = 8.0 * n * n * (n // 2 + 1)
print(json.dumps({"case": "2048^3 f32 single GPU", "ms": round(t * 1e3, 3), "frac_hbm_peak": round(5 * sweep / t / 8e12, 4),
                  "kernel_ms[x,y,z,reduce,x_fix]": [round(float(v), 3) for v in kern / 4]}), flush=True)
plan.close()
if quick:
    sys.exit(0)
# per-rank slab compute, rank 0 and rank 3 of 8 (virtual ranks: forward = generation + x + y on the kz slab, backward = z)
for r in (0, 3):
    p = _hip.DevicePlan(n, n, n, np.complex64, nranks=8, rank=r)
    p.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
    p.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    p.slab_forward(seed=1)
    fw, bw = [], []
    for i in range(4):
        p.sync()
        t0 = time.perf_counter()
        p.slab_forward(seed=2 + i)
        t1 = time.perf_counter()
        p.slab_backward()
        t2 = time.perf_counter()
        fw.append(t1 - t0)
        bw.append(t2 - t1)
    print(json.dumps({"case": "2048^3 / 8 ranks, rank %d" % r, "forward_ms": round(float(np.median(fw)) * 1e3, 3),
                      "backward_ms": round(float(np.median(bw)) * 1e3, 3)}), flush=True)
    p.close()
