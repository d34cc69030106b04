#!/usr/bin/env python3
"""Forward half (generation + x + y) of one virtual rank of 2048^3 / 8 against RF_FLAG_EXCHANGE_CHUNKS (round 5):
with C sub-slabs of 128 / C kz planes the x pass of a sub-slab is followed at once by its y pass, which then finds the sub-slab
(4.3 GB / C) in the Infinity Cache when C >= 16.   python3 tools/chunk_fwd.py [rank]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 2048
power = powertools.load_default_power()
p = _hip.DevicePlan(n, n, n, np.complex64, nranks=8, rank=rank)
p.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
p.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
out = {"rank": rank}
for C in (1, 2, 4, 8, 16):
    try:
        p.set_exchange_chunks(C)
    except RuntimeError as e:
        out["C=%d" % C] = str(e)
        continue
    p.slab_forward(seed=1)
    p.slab_backward()
    fw, bw = [], []
    for i in range(4):
        p.sync()
        t0 = time.perf_counter()
        p.slab_forward(seed=2 + i)
        t1 = time.perf_counter()
        p.slab_backward()
        t2 = time.perf_counter()
        fw.append(t1 - t0)
        bw.append(t2 - t1)
    out["C=%d" % C] = {"forward_ms": round(float(np.median(fw)) * 1e3, 3), "backward_ms": round(float(np.median(bw)) * 1e3, 3)}
print(json.dumps(out), flush=True)
p.close()
