#!/usr/bin/env python3
"""Virtual-rank slab pipeline against the single-rank field on a few x-planes, for a list of shapes (development tool).
usage: slab_probe.py nx,ny,nz,P [nx,ny,nz,P ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref                                    # noqa: E402  (checker only)
from randomfield_amd import _hip, powertools                  # noqa: E402

power = powertools.load_default_power()
for arg in sys.argv[1:]:
    nx, ny, nz, P = [int(v) for v in arg.split(",")]
    tabs = cpu_ref.sigma_table(power["k"], power["Pk"], nx, ny, nz, 2.5)
    one = _hip.DevicePlan(nx, ny, nz, np.complex64)
    one.set_kgrid(*powertools.ksq_axes(nx, ny, nz, 2.5))
    one.set_power(*tabs)
    one.realise(seed=4)
    mean, std = one.moments()
    nxl = nx // P
    planes = sorted(set([0, nxl - 1, nxl, nx // 2 + 3, nx - 1]))
    ref = {x: one.download_real(x0=x, x1=x + 1).copy() for x in planes}
    one.close()
    plans = []
    for r in range(P):
        p = _hip.DevicePlan(nx, ny, nz, np.complex64, nranks=P, rank=r)
        p.set_kgrid(*powertools.ksq_axes(nx, ny, nz, 2.5))
        p.set_power(*tabs)
        plans.append(p)
    for p in plans:
        p.slab_forward(seed=4)
    _hip.DevicePlan.slab_exchange_local(plans)
    s1 = s2 = 0.0
    worst = 0.0
    for r, p in enumerate(plans):
        p.slab_backward()
        a, b = p.slab_stats()
        s1, s2 = s1 + a, s2 + b
        for x in planes:
            if r * nxl <= x < (r + 1) * nxl:
                got = p.download_real(x0=x - r * nxl, x1=x - r * nxl + 1)
                worst = max(worst, float(np.max(np.abs(got - ref[x])) / std))
    cells = float(nx) * ny * nz
    rms = np.sqrt(s2 / cells - (s1 / cells) ** 2)
    print("%s P=%d: std %.6f slab rms %.6f  worst plane diff %.3g * rms" % ((nx, ny, nz), P, std, rms, worst), flush=True)
    for p in plans:
        p.close()
