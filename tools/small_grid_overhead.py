#!/usr/bin/env python3
"""Host-side cost of one Generator call on small grids, where the GPU work is tens of microseconds: wall time per call of
generate_delta_field (field left on the device / returned as a numpy array) against the GPU time of the realisation itself.
usage: small_grid_overhead.py [edge ...]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import Generator      # noqa: E402


def run(e, rng, reps=50):
    gen = Generator(e, e, e, 2.5, backend="hip", rng=rng)
    gen.generate_delta_field(seed=1, save_potential=False, download=False)
    out = {}
    for label, kw in (("device", dict(download=False)), ("numpy", dict(download=True))):
        t0 = time.perf_counter()
        for i in range(reps):
            gen.generate_delta_field(seed=2 + i, save_potential=False, **kw)
        gen.plan_c2r.device.sync()
        out[label] = (time.perf_counter() - t0) / reps * 1e3
    dev = gen.plan_c2r.device
    dev.realise(seed=5)
    dev.sync()
    gpu = dev.elapsed_ms()
    print("%4d^3 rng=%-9s per call: %.3f ms (field on the device), %.3f ms (numpy array); GPU time of a realisation %.3f ms" % (e, rng, out["device"], out["numpy"], gpu), flush=True)
    return gen


if __name__ == "__main__":
    edges = [int(a) for a in sys.argv[1:]] or [64, 128, 256]
    for e in edges:
        for rng in ("native", "reference"):
            gen = run(e, rng)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(50):
        gen.generate_delta_field(seed=100 + i, save_potential=False, download=False)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
