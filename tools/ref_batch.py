#!/usr/bin/env python3
"""same-seed (rng='reference') realisations: one by one against rf_realise_batch_reference.  usage: ref_batch.py [n] [K]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
seeds = list(range(200, 200 + K))
plan.reference_noise(1, single=True)
plan.realise(noise="resident")
plan.sync()
for rep in range(2):
    t0 = time.perf_counter()
    for sd in seeds:
        plan.reference_noise(sd, single=True)
        plan.realise(noise="resident")
    plan.sync()
    one = (time.perf_counter() - t0) * 1e3 / K
    last = plan.download_real(x0=0, x1=1).copy()
    t0 = time.perf_counter()
    plan.realise_batch_reference(seeds, want_rms=False)
    plan.sync()
    bat = (time.perf_counter() - t0) * 1e3 / K
    same = np.array_equal(plan.download_real(x0=0, x1=1), last)
    print("n %d: one by one %.3f ms per realisation, batched %.3f ms (host seeding included), last field identical %s" % (n, one, bat, same), flush=True)
plan.close()
