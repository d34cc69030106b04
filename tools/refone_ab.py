#!/usr/bin/env python3
"""The one-call same-seed realisation (rf_realise_batch_reference with one seed) with a variant library: ms per call + field checksum.
usage: tools/refone_ab.py [variant.so]"""
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
plan.realise_batch_reference([1], want_rms=False)
plan.sync()
ts = []
for sd in range(10, 20):
    t0 = time.perf_counter()
    plan.realise_batch_reference([sd], want_rms=False)
    plan.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
rms = plan.realise_batch_reference([123], want_rms=True)
print("%s: one-call same-seed realisation median %.3f ms (min %.3f)  rms(seed 123) %.7f" % (sys.argv[1] if len(sys.argv) > 1 else "product", float(np.median(ts)), min(ts), rms[0]), flush=True)
plan.close()
