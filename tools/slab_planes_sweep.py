#!/usr/bin/env python3
"""Batch time at 1024^3 against the y / z slab size in x planes (RF_FLAG_YZ_SLAB_PLANES), sizes that do not divide nx included (the last
slab is smaller); merged z + y launches.  python3 tools/slab_planes_sweep.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n, steps = 1024, 20
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
ref = None
for B in (64, 32, 40, 48, 56, 64, 72, 80, 96, 128, 64):
    plan.set_yz_slab_planes(B)
    plan.realise_batch_prepare(steps)
    plan.realise_batch(np.arange(5, dtype=np.uint64), want_rms=False)
    plan.sync()
    ts = []
    for r in range(3):
        t0 = time.perf_counter()
        plan.realise_batch(np.arange(100, 100 + steps, dtype=np.uint64), want_rms=False)
        plan.sync()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    m = plan.moments()
    ref = ref or m
    print(json.dumps({"planes": B, "MB": B * 4, "slabs": plan.yz_slabs(), "ms_per_step": [round(t, 4) for t in ts], "same_field": m == ref}), flush=True)
plan.close()
