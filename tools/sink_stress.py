#!/usr/bin/env python3
"""Stress of the host sink's pin / unpin life cycle next to pageable downloads (debugging)."""
import gc
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools, transform, Generator   # noqa: E402

power = powertools.load_default_power()
mode = sys.argv[1] if len(sys.argv) > 1 else "sink"
for it in range(60):
    n = (64, 128, 256)[it % 3]
    if mode == "sink":
        gen = Generator(n, n, n, 2.5, rng="native")
        a = gen.generate_delta_field(seed=it, save_potential=(it % 2 == 0))
        chk = float(a.std())
        del gen, a
    plans = []
    for r in range(2):
        p = _hip.DevicePlan(64, 64, 64, np.complex64, nranks=2, rank=r)
        p.set_kgrid(*powertools.ksq_axes(64, 64, 64, 2.5))
        p.set_power(*powertools.sigma_table(power, (64, 64, 64), 2.5))
        plans.append(p)
    for p in plans:
        p.slab_forward(seed=it)
    _hip.DevicePlan.slab_exchange_local(plans)
    parts = []
    for p in plans:
        p.slab_backward()
        parts.append(p.download_real())
    for p in plans:
        p.close()
    junk = [np.empty((32, 64, 64), np.float32) for _ in range(8)]
    one = _hip.DevicePlan(64, 64, 64, np.complex64)
    one.set_kgrid(*powertools.ksq_axes(64, 64, 64, 2.5))
    one.set_power(*powertools.sigma_table(power, (64, 64, 64), 2.5))
    one.realise(seed=it)
    ref = one.download_real()
    assert np.max(np.abs(np.concatenate(parts, axis=0) - ref)) <= 1e-5 * ref.std()
    one.close()
    if it % 10 == 9:
        gc.collect()
        print("iteration", it, "ok", flush=True)
print("done", mode)
