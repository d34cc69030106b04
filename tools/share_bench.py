#!/usr/bin/env python3
"""Shared against replicated replay of the reference's stream on kz-slab ranks (virtual ranks on one GPU; development tool):
per-rank times of rf_mt_share_begin / pack / finish against rf_noise_mt19937_ex on the same multi-rank plan.
usage: tools/share_bench.py [n = 1024] [ranks = 8] [one: only rank 3's local replay + pack (a grid whose P plans do not fit one GPU)]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
power = powertools.load_default_power()
plans = []
if len(sys.argv) > 3:
    p = _hip.DevicePlan(n, n, n, np.complex64, nranks=P, rank=min(3, P - 1))
    p.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
    p.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    nseg, first, count = p.share_segments()
    for single in (True, False):
        ts = []
        for rep in range(3):
            p.sync()
            t0 = time.perf_counter()
            c = p.share_begin(5 + rep, single)
            t1 = time.perf_counter()
            # the other ranks' counts: this rank's own, repeated (same distribution; only the pack's addresses depend on them)
            allc = np.resize(c, nseg)
            allc[first:first + count] = c
            p.share_pack(allc)
            ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
        print(json.dumps({"case": "%d^3 / %d ranks, rank %d alone, %s transport" % (n, P, p.rank, "float32" if single else "float64"),
                          "segments": [nseg, first, count], "begin_ms": round(ts[-1][0], 2), "pack_ms": round(ts[-1][1], 2)}), flush=True)
    p.close()
    sys.exit(0)
for r in range(P):
    p = _hip.DevicePlan(n, n, n, np.complex64, nranks=P, rank=r)
    p.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
    p.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    plans.append(p)


def timed(fn):
    plans[0].sync()
    t0 = time.perf_counter()
    out = fn()
    return out, (time.perf_counter() - t0) * 1e3


for single in (True, False):
    for rep in range(2):                      # (the first round allocates)
        t_begin, counts = [], []
        for p in plans:
            c, t = timed(lambda: p.share_begin(123 + rep, single))
            counts.append(c)
            t_begin.append(t)
        allc = np.concatenate(counts)
        t_pack = [timed(lambda: p.share_pack(allc))[1] for p in plans]
        arr = (_hip.ctypes.c_void_p * P)(*[p._h.value for p in plans])
        _, t_x = timed(lambda: _hip.check(_hip.load().rf_mt_share_exchange_local(arr, P), "exchange"))
        t_fin = [timed(lambda: p.share_finish())[1] for p in plans]
    print(json.dumps({"case": "%d^3 / %d ranks, shared replay, %s transport" % (n, P, "float32" if single else "float64"),
                      "per_rank_ms": {"begin (jump + tree + generation pass)": [round(t, 2) for t in t_begin],
                                      "pack": [round(t, 2) for t in t_pack], "finish": [round(t, 2) for t in t_fin]},
                      "local_copies_ms (stand-in for the all-to-all)": round(t_x, 2),
                      "send_GB_per_rank": round((n * n * (n // 2 + 1) * (8 if single else 16) / P) / 1e9, 3)}), flush=True)
for single in (True, False):
    ts = []
    for rep in range(2):
        _, t = timed(lambda: plans[1].reference_noise(123, single=single))
        ts.append(t)
    print(json.dumps({"case": "%d^3 / %d ranks, replicated replay (every rank, whole stream), %s" % (n, P, "float32 runs" if single else "float64 + move"),
                      "per_rank_ms": round(ts[-1], 2)}), flush=True)
for p in plans:
    p.close()
