#!/usr/bin/env python3
"""EXPERIMENT (round 5): what a plane pitch that is not a power of two would give the strided passes.  RF_XPAD_CELLS (complex cells of
padding behind every x plane, read once by the library) is applied to the x pass's row stride and the y pass's plane stride only -- the
z pass still reads dense rows, so the field is WRONG; only times are meaningful.  One process per pad value:
    RF_XPAD_CELLS=544 python3 tools/xpad_probe.py [n]
prints per-pass event times of the native and of the same-seed (deviate-reading) pipeline and the graph-replayed batch time."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
power = powertools.load_default_power()
plan = _hip.DevicePlan(n, n, n, np.complex64)
plan.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5))
plan.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
out = {"pad_cells": int(os.environ.get("RF_XPAD_CELLS", "0")), "n": n}
plan.realise(seed=1)
plan.sync()
kern = np.zeros(5)
for i in range(5):
    plan.realise(seed=7 + i)
    plan.sync()
    kern += np.array(plan.kernel_ms())
out["native kernel_ms [x,y,z,reduce,x_kz0]"] = [round(float(v), 4) for v in kern / 5]
steps = 10
plan.realise_batch_prepare(steps)
plan.realise_batch(np.arange(3, dtype=np.uint64), want_rms=False)
plan.sync()
ts = []
for r in range(3):
    t0 = time.perf_counter()
    plan.realise_batch(np.arange(100 * r, 100 * r + steps, dtype=np.uint64), want_rms=False)
    plan.sync()
    ts.append((time.perf_counter() - t0) / steps * 1e3)
out["native batch ms_per_step"] = [round(t, 4) for t in ts]
kern = np.zeros(5)
for i in range(4):
    plan.reference_noise(500 + i, single=True)
    plan.realise(noise="resident")
    plan.sync()
    if i:
        kern += np.array(plan.kernel_ms())
out["same-seed kernel_ms [x,y,z,reduce,x_kz0]"] = [round(float(v), 4) for v in kern / 3]
ts = []
for r in range(3):
    t0 = time.perf_counter()
    plan.realise_batch_reference([900 + r], want_rms=False)
    plan.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
out["same-seed one call ms"] = [round(t, 3) for t in ts]
print(json.dumps(out), flush=True)
plan.close()
