#!/usr/bin/env python3
"""1024^3 float32 pipeline of a library variant: graph-replayed batch time, eager per-pass times, rms (development tool).
usage: tools/col2_ab.py [variant.so | -] [rounds] [f32 | f64]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, K = 1024, 20
dt = np.complex128 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.complex64
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, dt)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
seeds = np.arange(K, dtype=np.uint64)
ts = []
for r in range(rounds):
    plan.realise_batch(seeds[:5], want_rms=False)
    plan.sync()
    t0 = time.perf_counter()
    plan.realise_batch(seeds, want_rms=False)
    plan.sync()
    ts.append((time.perf_counter() - t0) * 1e3 / K)
kern = np.zeros(5)
for i in range(5):
    plan.realise(seed=7)
    plan.sync()
    kern += np.array(plan.kernel_ms())
plan.realise(seed=7)
print("%s  ms/realisation %s  min %.4f  kernel_ms[x,y,z,reduce,x_fix] %s  moments %r" % (
    sys.argv[1] if len(sys.argv) > 1 else "product", " ".join("%.4f" % t for t in ts), min(ts), np.round(kern / 5, 4).tolist(), plan.moments()), flush=True)
sub = plan.download_real(x0=5, x1=6)[0, ::64, ::64].astype(np.float64)
print("   checksum %.9e" % float(np.sum(sub * np.arange(sub.size).reshape(sub.shape))), flush=True)
plan.close()
