#!/usr/bin/env python3
"""A/B of the transposed x -> y intermediate (RF_FLAG_TRANSPOSED_INTERMEDIATE) in one process: per-kernel times and the
graph-replayed batch, flag on / off alternately.  usage: xp_ab.py [n] [variant.so] [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if len(sys.argv) > 2 and sys.argv[2] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
seeds = np.arange(10, dtype=np.uint64)
ref = None
for r in range(rounds):
    for on in (1, 0):
        plan.set_transposed_intermediate(bool(on))
        for i in range(3):
            plan.realise(seed=100 + i)
            plan.sync()
        km = [round(v, 3) for v in plan.kernel_ms()]
        rms = plan.moments()
        plan.realise_batch(seeds)
        t0 = time.perf_counter()
        plan.realise_batch(seeds, want_rms=False)
        plan.sync()
        dt = (time.perf_counter() - t0) * 100
        print("xposed=%d kernel_ms %s eager_total %.3f batch ms/realisation %.4f rms %r" % (on, km, plan.elapsed_ms(), dt, rms), flush=True)
        f = plan.download_real(x0=0, x1=2)
        if ref is None:
            ref = f.copy()
        else:
            assert np.array_equal(ref, f), "fields differ between the two layouts"
plan.close()
print("fields identical in both layouts")
