#!/usr/bin/env python3
"""What of the exchange's local traffic costs the passes beside it (round 5): one virtual rank of 2048^3 / 8 through the pipelined
batch with an unthrottled stand-in (128 workgroups) whose read and write shares vary -- reads alone, writes alone, half of both, all --
and the same with the sub-slab layout C = 4.  python3 tools/standin_attrib.py [rank]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, nreal = 2048, 8
power = powertools.load_default_power()
out = {"rank": rank}
for chunks in (1, 4):
    p = _hip.DevicePlan(n, n, n, np.complex64, nranks=8, rank=rank)
    p.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
    p.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    if chunks > 1:
        p.set_exchange_chunks(chunks)
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
    base = (float(np.median(fw)) + float(np.median(bw))) * 1e3
    res = {"forward_plus_backward_ms": round(base, 3)}
    for rd, wr in ((0, 0), (100, 0), (0, 100), (50, 50), (100, 100)):
        p.set_exchange_standin_ex(128, rd, wr)
        p.realise_batch(np.arange(3, dtype=np.uint64), want_rms=False)
        p.sync()
        ts = []
        for r in range(3):
            t0 = time.perf_counter()
            p.realise_batch(np.arange(100 * r, 100 * r + nreal, dtype=np.uint64), want_rms=False)
            p.sync()
            ts.append((time.perf_counter() - t0) / nreal * 1e3)
        res["read %d%% write %d%%" % (rd, wr)] = round(float(np.median(ts)), 3)
    out["C=%d" % chunks] = res
    p.close()
print(json.dumps(out), flush=True)
