#!/usr/bin/env python3
"""What the direct exchange costs on ONE GPU (round 6): one virtual rank of the 2048^3 / 8 job through the pipelined
batch with the y pass storing block h into segment h of its own receive buffers (rf_slab_set_direct_standin: the store pattern and
volume of the real direct exchange, without the links), next to the same rank's forward + backward alone and to the RCCL-style copy
stand-in of 128 workgroups.
    python3 tools/direct_bench.py [rank] [n_realisations]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402


def batch_ms(p, nreal, reps=3):
    p.realise_batch(np.arange(3, dtype=np.uint64), want_rms=False)
    p.sync()
    ts = []
    for r in range(reps):
        t0 = time.perf_counter()
        p.realise_batch(np.arange(100 * r, 100 * r + nreal, dtype=np.uint64), want_rms=False)
        p.sync()
        ts.append((time.perf_counter() - t0) / nreal * 1e3)
    return round(float(np.median(ts)), 3)


def measure(rank=0, nreal=8, n=2048, ranks=8, device=0, spacing=2.5, power=None, copy_width=128, chunk_counts=(1, 4)):
    power = powertools.load_default_power() if power is None else power
    p = _hip.DevicePlan(n, n, n, np.complex64, device=device, nranks=ranks, rank=rank)
    p.set_kgrid(*powertools.ksq_axes(n, n, n, spacing))
    p.set_power(*powertools.sigma_table(power, (n, n, n), spacing))
    out = {"rank": rank, "grid": [n, n, n], "ranks": ranks}
    for C in chunk_counts:
        p.set_direct_standin(False)
        p.set_exchange_standin(0)
        p.set_exchange_chunks(C)
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
        ent = {"forward_ms": round(float(np.median(fw)) * 1e3, 3), "backward_ms": round(float(np.median(bw)) * 1e3, 3)}
        ent["forward_plus_backward_ms"] = round(ent["forward_ms"] + ent["backward_ms"], 3)
        if copy_width:
            p.set_exchange_standin(copy_width)
            ent["copy stand-in, %d workgroups: pipelined ms per realisation" % copy_width] = batch_ms(p, nreal)
            p.set_exchange_standin(0)
        for overlap in (False, True):
            p.set_direct_standin(True, overlap=overlap)
            key = "direct stand-in, %s" % ("storing y pass on the exchange stream" if overlap else "one stream")
            ent[key + ": pipelined ms per realisation"] = batch_ms(p, nreal)
            t0 = time.perf_counter()
            p.realise(seed=5)
            p.sync()
            ent[key + ": one realisation ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        p.set_direct_standin(False)
        out["%d sub-slab%s" % (C, "" if C == 1 else "s")] = ent
    p.close()
    return out


if __name__ == "__main__":
    rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    nreal = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    print(json.dumps(measure(rank, nreal), indent=1), flush=True)
