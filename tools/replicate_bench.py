"""GPU probe: one virtual rank of a P-rank plan in replicated-generation mode (weak scaling: 1024^3 cells per rank)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools
power = powertools.load_default_power()
for P, shape in ((2, (1024, 1024, 2048)), (4, (1024, 2048, 2048)), (8, (2048, 2048, 2048))):
    nx, ny, nz = shape
    p = _hip.DevicePlan(nx, ny, nz, np.complex64, nranks=P, rank=1)
    p.set_kgrid(*powertools.ksq_axes(nx, ny, nz, 2.5)); p.set_power(*powertools.sigma_table(power, shape, 2.5))
    p.set_replicated_generation(True)
    p.realise(seed=1); p.sync()
    p.realise(seed=2); p.sync()
    k = p.kernel_ms()
    t0 = time.perf_counter()
    p.realise_batch(np.arange(10, 15, dtype=np.uint64), want_rms=False); p.sync()
    t = (time.perf_counter() - t0) / 5
    print("P=%d %s: %.2f ms per realisation per rank (x %.2f + %.2f, y %.2f, z+reduce %.2f) -> %.0f Gcells/s for the job" %
          (P, shape, t * 1e3, k[0], k[4], k[1], k[2], nx * ny * nz / t / 1e9), flush=True)
    p.close()
