#!/usr/bin/env python3
"""does the default call slow down after other big plans were created and destroyed in the process? (development probe)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import Generator, powertools, _hip
power = powertools.load_default_power()
s = iter(range(1, 10000))
def timed(fn, sync, reps=5, warm=3):
    for _ in range(warm): fn()
    sync(); ts = []
    for _ in range(reps):
        sync(); t0 = time.perf_counter(); fn(); sync(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 3)
def gen_case(tag):
    gen = Generator(1024, 1024, 1024, 2.5, power=power, rng="native")
    dev = gen.plan_c2r.device
    a = timed(lambda: gen.generate_delta_field(seed=next(s), save_potential=True, download=False), dev.sync)
    b = timed(lambda: gen.generate_delta_field(seed=next(s), save_potential=False, download=False), dev.sync)
    print(tag, "save_potential=True", a, "False", b, flush=True)
    dev.close()
if not os.environ.get("SKIP_FRESH"): gen_case("fresh      ")
p = _hip.DevicePlan(1024, 1024, 1024, np.complex128)
p.set_kgrid(*powertools.ksq_axes(1024, 1024, 1024, 2.5)); p.set_power(*powertools.sigma_table(power, (1024,) * 3, 2.5))
p.realise(seed=1); p.sync(); p.close()
gen_case("after f64  ")
if not os.environ.get("SKIP_FRESH"):
    p = _hip.DevicePlan(1024, 1024, 1024, np.complex64)
    p.set_kgrid(*powertools.ksq_axes(1024, 1024, 1024, 2.5)); p.set_power(*powertools.sigma_table(power, (1024,) * 3, 2.5))
    p.reference_noise(5, single=True); p.realise(noise="resident"); p.sync(); p.close()
    gen_case("after mt   ")
