#!/usr/bin/env python3
"""One BASELINE configuration (or API path) a few times over, for rocprofv3 (tools/profile_cfg.sh):

    python3 tools/cfg_workload.py <case> [reps] [variant library]

cases: 512 | 1024 | ref (1024^3, rng='reference': MT19937 replay + generation pass reading the deviates) | refone | refbatch |
       f64 (1024^3 float64) | f64ln (config 5: float64 + lognormal, fused) | 2048 (2048^3 float32 on one GPU) |
       rank0 / rank3 (per-rank compute of the 2048^3 / 8 job, virtual ranks: forward + backward halves) |
       rank0direct / rank3direct (the same rank through the pipelined batch of the direct exchange, its stores into its own buffers)
Prints one JSON line: wall ms per call (median) and the plan's per-pass event times where the call records them."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "1024"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if len(sys.argv) > 3:
    _hip.LIB_PATH = os.path.abspath(sys.argv[3])
power = powertools.load_default_power()


def make(n, dtype, **kw):
    plan = _hip.DevicePlan(n, n, n, dtype, **kw)
    plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
    plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    return plan


def timed(plan, fn, reps, kernel_ms=True):
    fn(0)
    plan.sync()
    ts, kern = [], np.zeros(5)
    for i in range(reps):
        plan.sync()
        t0 = time.perf_counter()
        fn(1 + i)
        plan.sync()
        ts.append(time.perf_counter() - t0)
        if kernel_ms:
            kern += np.array(plan.kernel_ms())
    out = {"case": case, "ms": round(float(np.median(ts)) * 1e3, 3)}
    if kernel_ms:
        out["kernel_ms[x,y,z,reduce,x_kz0]"] = [round(float(v), 4) for v in kern / reps]
    return out


if case in ("512", "1024", "2048"):
    n = int(case)
    plan = make(n, np.complex64)
    res = timed(plan, lambda i: plan.realise(seed=100 + i), reps)
elif case == "ref":
    plan = make(1024, np.complex64)
    split = {}

    def f(i):
        t0 = time.perf_counter()
        plan.reference_noise(100 + i, single=True)
        plan.sync()
        split["replay_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        plan.realise(noise="resident")
    res = timed(plan, f, reps)
    res.update(split)
elif case == "refone":              # what Generator(rng='reference') runs: replay + passes as ONE device call (a batch of one seed)
    plan = make(1024, np.complex64)
    res = timed(plan, lambda i: plan.realise_batch_reference([300 + i], want_rms=False), reps, kernel_ms=False)
elif case == "refbatch":
    plan = make(1024, np.complex64)
    seeds = list(range(200, 200 + max(reps, 2)))
    plan.realise_batch_reference(seeds[:2], want_rms=False)
    plan.sync()
    t0 = time.perf_counter()
    plan.realise_batch_reference(seeds, want_rms=False)
    plan.sync()
    res = {"case": case, "ms": round((time.perf_counter() - t0) / len(seeds) * 1e3, 3)}
elif case == "f64":
    plan = make(1024, np.complex128)
    res = timed(plan, lambda i: plan.realise(seed=100 + i), reps)
elif case == "f64ln":
    plan = make(1024, np.complex128)
    plan.set_z_tables(np.exp(-0.5 * np.arange(1024) / 1024))
    res = timed(plan, lambda i: plan.realise_lognormal(seed=100 + i, want_sigma=False), reps, kernel_ms=False)
elif case in ("rank0", "rank3"):
    plan = make(2048, np.complex64, nranks=8, rank=int(case[4:]))
    plan.slab_forward(seed=1)
    fw, bw = [], []
    for i in range(reps):
        plan.sync()
        t0 = time.perf_counter()
        plan.slab_forward(seed=2 + i)
        t1 = time.perf_counter()
        plan.slab_backward()
        t2 = time.perf_counter()
        fw.append(t1 - t0)
        bw.append(t2 - t1)
    res = {"case": case, "forward_ms": round(float(np.median(fw)) * 1e3, 3), "backward_ms": round(float(np.median(bw)) * 1e3, 3)}
elif case in ("rank0direct", "rank3direct"):      # the same rank through the DIRECT exchange's schedule (rf_slab_set_direct_standin, one stream)
    plan = make(2048, np.complex64, nranks=8, rank=int(case[4]))
    plan.set_direct_standin(True, overlap=False)
    plan.realise_batch(np.arange(2, dtype=np.uint64), want_rms=False)
    plan.sync()
    n = max(reps, 2) * 2
    t0 = time.perf_counter()
    plan.realise_batch(np.arange(10, 10 + n, dtype=np.uint64), want_rms=False)
    plan.sync()
    res = {"case": case, "ms": round((time.perf_counter() - t0) / n * 1e3, 3), "note": "per realisation of a pipelined batch; storing y pass = col2_direct_kernel"}
else:
    sys.exit("unknown case %r" % case)
print(json.dumps(res), flush=True)
plan.close()
