#!/usr/bin/env python3
"""rng='reference' at 1024^3 a few times (for rocprofv3 --kernel-trace --stats): MT19937 replay + exact-chain pipeline."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
single = len(sys.argv) > 2 and sys.argv[2] == "single"
pot = len(sys.argv) > 3 and sys.argv[3] == "pot"
if len(sys.argv) > 4:
    _hip.LIB_PATH = os.path.abspath(sys.argv[4])      # a variant build of the library (kernel experiments)
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
for i in range(4):
    plan.reference_noise(100 + i, single=single)
    (plan.realise_potential if pot else plan.realise)(noise="resident")
    plan.sync()
print(plan.moments())
plan.close()
