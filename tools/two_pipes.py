#!/usr/bin/env python3
"""Two independent 1024^3 float32 pipelines (two plans, two streams) on one GPU: 2 x K graph-replayed realisations one plan after the
other against both at once.  Does the vector-bound generation pass of one overlap the y / z passes of the other?  usage: two_pipes.py [K] [variant.so]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2 and sys.argv[2] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[2])           # a variant build of the library (tools/bin/lib_*.so)
n = 1024
power = powertools.load_default_power()
plans = []
for i in range(2):
    p = _hip.DevicePlan(n, n, n, np.complex64)
    p.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
    p.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    p.realise_batch_prepare(K)
    p.realise_batch(np.arange(5, dtype=np.uint64), want_rms=False)
    p.sync()
    plans.append(p)
seeds = np.arange(100, 100 + K, dtype=np.uint64)
for rep in range(3):
    t0 = time.perf_counter()
    for p in plans:
        p.realise_batch(seeds, want_rms=False)
        p.sync()
    seq = (time.perf_counter() - t0) / (2 * K) * 1e3
    t0 = time.perf_counter()
    for p in plans:
        p.realise_batch(seeds, want_rms=False)
    for p in plans:
        p.sync()
    both = (time.perf_counter() - t0) / (2 * K) * 1e3
    print("one after the other %.3f ms per realisation, both at once %.3f ms per realisation" % (seq, both), flush=True)
print("rms", [round(p.moments()[1], 6) for p in plans])
for p in plans:
    p.close()
