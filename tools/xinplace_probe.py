#!/usr/bin/env python3
"""How fast is a strided pass IN PLACE along x (64-byte segments one plane = 4 MiB apart) next to the y pass (one row = 4 KiB
apart)?  Run under rocprofv3 --kernel-trace: r2c then c2r on a 1024^3 plan; the c2r's first strided launch is the x pass in place."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
plan.realise(seed=3)
plan.sync()
for i in range(3):
    plan.execute_r2c()
    plan.sync()
    plan.execute_c2r()
    plan.sync()
    print("c2r kernel_ms", [round(v, 3) for v in plan.kernel_ms()], flush=True)
print("rms", plan.moments()[1])
plan.close()
