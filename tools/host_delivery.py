#!/usr/bin/env python3
"""PCIe-inclusive time of the drop-in call at 1024^3: Generator.generate_delta_field(seed) returning a numpy array, delivered slab by slab
behind the z pass (rf_set_host_sink) against realisation-then-download.   python3 tools/host_delivery.py [n]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import Generator   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = {}
for rng in ("native", "reference"):
    gen = Generator(n, n, n, 2.5, rng=rng)
    dev = gen.plan_c2r.device
    t0 = time.perf_counter()
    gen.generate_delta_field(seed=1, save_potential=False)
    first = time.perf_counter() - t0                       # includes pinning the host buffer
    sink, serial = [], []
    for i in range(4):
        t0 = time.perf_counter()
        a = gen.generate_delta_field(seed=10 + i, save_potential=False)
        sink.append(time.perf_counter() - t0)
        chk = float(a[::97, ::89, ::83].std())
        t0 = time.perf_counter()
        gen.generate_delta_field(seed=10 + i, save_potential=False, download=False)
        b = gen.download_field()
        serial.append(time.perf_counter() - t0)
        assert float(b[::97, ::89, ::83].std()) == chk
    out[rng] = {"first_call_ms": round(first * 1e3, 1), "delivered_behind_z_ms": [round(t * 1e3, 2) for t in sink],
                "realise_then_download_ms": [round(t * 1e3, 2) for t in serial]}
    dev.close()
print(json.dumps(out, indent=1))
