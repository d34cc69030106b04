#!/usr/bin/env python3
"""One same-seed realisation per call: rf_noise_mt19937 + rf_realise (a host sync between the two) against
rf_realise_batch_reference with ONE seed (replay and passes queued back to back).  usage: ref_single.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
plan.reference_noise(1, single=True)
plan.realise(noise="resident")
plan.sync()
plan.realise_batch_reference([3], want_rms=False)
plan.sync()
for rep in range(3):
    ta, tb = [], []
    for sd in range(10, 16):
        t0 = time.perf_counter()
        plan.reference_noise(sd, single=True)
        plan.realise(noise="resident")
        m = plan.moments()
        ta.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        rms = plan.realise_batch_reference([sd], want_rms=True)
        tb.append((time.perf_counter() - t0) * 1e3)
        assert abs(rms[0] - m[1]) <= 1e-6 * m[1], (rms, m)
    print("n %d: two calls %.3f ms (min %.3f), batch of one %.3f ms (min %.3f)" % (n, np.median(ta), min(ta), np.median(tb), min(tb)), flush=True)
plan.close()
