#!/usr/bin/env python3
"""MT19937 replay alone at 1024^3 (float32 pairs): ms per replay and a checksum of the field it produces.
usage: tools/mt_ab.py [variant.so]"""
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
ts = []
for i in range(8):
    plan.sync()
    t0 = time.perf_counter()
    plan.reference_noise(100 + i, single=True)
    plan.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
plan.reference_noise(123, single=True)
plan.realise(noise="resident")
print("%s replay ms: %s  median %.3f  field moments %r" % (sys.argv[1] if len(sys.argv) > 1 else "product", " ".join("%.3f" % t for t in ts),
                                                         float(np.median(ts[2:])), plan.moments()), flush=True)
plan.close()
