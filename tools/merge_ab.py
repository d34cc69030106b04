#!/usr/bin/env python3
"""Graph-replayed realisations with and without the merged z + y launches (rf_k_yz.hip), per grid:
    python3 tools/merge_ab.py [edge ...]        (default 512 1024 2048)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools   # noqa: E402


def run(n, reps=5, spacing=2.5):
    power = powertools.load_default_power()
    nb = 20 if n <= 512 else (10 if n <= 1024 else 2)
    out = {"grid": n, "realisations_per_graph": nb}
    plan = _hip.DevicePlan(n, n, n, np.complex64)
    plan.set_kgrid(*powertools.ksq_axes(n, n, n, spacing))
    plan.set_power(*powertools.sigma_table(power, (n, n, n), spacing))
    out["yz_slabs"] = plan.yz_slabs()
    for rnd in range(2):
        for mode in (0, 1):
            plan.set_merged_yz(mode)
            plan.realise_batch_prepare(nb)
            plan.realise_batch(np.arange(nb, dtype=np.uint64), want_rms=False)
            plan.sync()
            ts = []
            for r in range(reps):
                t0 = time.perf_counter()
                plan.realise_batch(np.arange(100 * r, 100 * r + nb, dtype=np.uint64), want_rms=False)
                plan.sync()
                ts.append((time.perf_counter() - t0) / nb * 1e3)
            out["round %d, %s" % (rnd, "merged" if mode else "one launch per pass")] = round(float(np.median(ts)), 4)
    plan.close()
    return out


if __name__ == "__main__":
    edges = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]
    for n in edges:
        print(json.dumps(run(n)), flush=True)
