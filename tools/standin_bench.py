#!/usr/bin/env python3
"""What the exchange costs the passes it runs beside, measured on ONE GPU (round 5): one virtual rank of the
2048^3 / 8 job (kernels and layouts of BASELINE config 4) through the REAL pipelined schedule of `rf_realise_batch` -- forward half
of realisation i + 1 on the compute stream while the exchange of realisation i runs on the exchange stream -- with the all-to-all
replaced by a copy kernel of fixed width (rf_slab_set_exchange_standin: 16 / 32 workgroups = RCCL's channel footprint) that reads the
3.76 GB the rank would send and writes the 3.76 GB it would receive.
    python3 tools/standin_bench.py [rank] [n_realisations] [chunks]
prints forward / backward alone, their sum, and the pipelined batch per realisation for each stand-in width."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402


def measure(rank=0, nreal=8, n=2048, ranks=8, widths=(16, 32, 64, 128, 256), device=0, spacing=2.5, power=None):
    power = powertools.load_default_power() if power is None else power
    p = _hip.DevicePlan(n, n, n, np.complex64, device=device, nranks=ranks, rank=rank)
    p.set_kgrid(*powertools.ksq_axes(n, n, n, spacing))
    p.set_power(*powertools.sigma_table(power, (n, n, n), spacing))
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
    out = {"rank": rank, "grid": [n, n, n], "ranks": ranks,
           "forward_ms": round(float(np.median(fw)) * 1e3, 3), "backward_ms": round(float(np.median(bw)) * 1e3, 3)}
    out["forward_plus_backward_ms"] = round(out["forward_ms"] + out["backward_ms"], 3)
    sweep = 8.0 * n * n * (n // 2 + 1)
    out["standin_bytes_read_and_written_each"] = (ranks - 1) / ranks * sweep / ranks
    pipe = {}
    for w in widths:
        p.set_exchange_standin(w)
        p.realise_batch(np.arange(3, dtype=np.uint64), want_rms=False)
        p.sync()
        ts = []
        for r in range(3):
            t0 = time.perf_counter()
            p.realise_batch(np.arange(100 * r, 100 * r + nreal, dtype=np.uint64), want_rms=False)
            p.sync()
            ts.append((time.perf_counter() - t0) / nreal * 1e3)
        # the exchange stand-in alone: one realisation, unpipelined = forward + stand-in + backward in sequence
        t0 = time.perf_counter()
        p.realise(seed=5)
        p.sync()
        seq = (time.perf_counter() - t0) * 1e3
        pipe["%d workgroups" % w] = {"pipelined_ms_per_realisation": round(float(np.median(ts)), 3),
                                     "slowdown_vs_forward_plus_backward": round(float(np.median(ts)) / out["forward_plus_backward_ms"], 4),
                                     "one_realisation_in_sequence_ms": round(seq, 3),
                                     "standin_alone_ms": round(seq - out["forward_plus_backward_ms"], 3)}
    out["exchange_standin"] = pipe
    p.set_exchange_standin(0)
    p.close()
    return out


if __name__ == "__main__":
    rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    nreal = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    print(json.dumps(measure(rank, nreal)), flush=True)
