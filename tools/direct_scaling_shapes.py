#!/usr/bin/env python3
"""The grids `bench.py --gpus N` runs at N = 2, 4, 8 (weak scaling from 1024^3: (1024,1024,2048), (1024,2048,2048), 2048^3), as N
virtual ranks on ONE GPU: the direct exchange's field against the copy exchange's, bit for bit, with 1 and 4 sub-slabs.
usage: direct_scaling_shapes.py [N ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                       # noqa: E402
from randomfield_amd import _hip, powertools      # noqa: E402

POWER = powertools.load_default_power()


def run(plans, direct, seed):
    for p in plans:
        p.slab_forward(seed=seed)
    if not direct:
        _hip.DevicePlan.slab_exchange_local(plans)
    for p in plans:
        p.slab_backward()
    # every rank's exact (sum, sum of squares) of its whole slab + its first two and last two x planes
    return [p.slab_stats() for p in plans], [np.concatenate([p.download_real(x0=0, x1=2), p.download_real(x0=p.nx_local - 2, x1=p.nx_local)]) for p in plans]


def main():
    for n in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
        shape = bench.grid_for(n, 1024)
        nx, ny, nz = shape
        plans = []
        for r in range(n):
            p = _hip.DevicePlan(nx, ny, nz, np.complex64, nranks=n, rank=r)
            p.set_kgrid(*powertools.ksq_axes(nx, ny, nz, 2.5))
            p.set_power(*powertools.sigma_table(POWER, shape, 2.5))
            plans.append(p)
        for chunks in (1, 4):
            for p in plans:
                p.set_exchange_chunks(chunks)
            want = run(plans, False, 77)
            _hip.DevicePlan.slab_link_direct(plans)
            got = run(plans, True, 77)
            _hip.DevicePlan.slab_link_direct(plans, False)
            same = want[0] == got[0] and all(np.array_equal(a, b) for a, b in zip(want[1], got[1]))
            s1 = sum(s[0] for s in got[0]); s2 = sum(s[1] for s in got[0]); cells = float(nx) * ny * nz
            print("N = %d  %s  %d sub-slab(s): direct == copy exchange: %s   rms %.6f" % (n, shape, chunks, same, np.sqrt(s2 / cells - (s1 / cells) ** 2)), flush=True)
            assert same
        for p in plans:
            p.close()


if __name__ == "__main__":
    main()
