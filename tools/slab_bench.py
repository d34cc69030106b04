#!/usr/bin/env python3
"""Per-rank compute time of the 8-GPU 2048^3 job, measured on ONE GPU with a virtual-rank plan
(DevicePlan(nranks=8, rank=r)): forward = generation + x pass + y pass on the kz slab, backward = gathering
z pass on the x slab.  The all-to-all itself cannot be measured on a 1-GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools

def run(shape, nranks, rank):
    nx, ny, nz = shape
    power = powertools.load_default_power()
    p = _hip.DevicePlan(nx, ny, nz, np.complex64, nranks=nranks, rank=rank)
    p.set_kgrid(*powertools.ksq_axes(nx, ny, nz, 2.5))
    p.set_power(*powertools.sigma_table(power, (nx, ny, nz), 2.5))
    p.slab_forward(seed=1); p.slab_backward()
    tf, tb = [], []
    for i in range(5):
        p.sync(); t0 = time.perf_counter(); p.slab_forward(seed=10 + i); t1 = time.perf_counter()
        p.slab_backward(); t2 = time.perf_counter()
        tf.append(t1 - t0); tb.append(t2 - t1)
    p.close()
    print(json.dumps({"shape": shape, "nranks": nranks, "rank": rank, "forward_ms": round(float(np.median(tf)) * 1e3, 3),
                      "backward_ms": round(float(np.median(tb)) * 1e3, 3)}), flush=True)

run((2048, 2048, 2048), 8, 0)
run((2048, 2048, 2048), 8, 3)
run((1024, 2048, 2048), 4, 1)
run((1024, 1024, 2048), 2, 1)
