"""GPU probe: the on-device MT19937 + polar replay against numpy, value by value."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip
for shape, seed in (((64, 64, 64), 123), ((128, 128, 256), 7), ((256, 256, 256), 2024), ((512, 512, 512), 123)):
    nx, ny, nz = shape
    p = _hip.DevicePlan(nx, ny, nz)
    p.reference_noise(seed); p.sync()
    t0 = time.time(); p.reference_noise(seed); p.sync(); t1 = time.time()
    n = 2 * nx * ny * (nz // 2 + 1)
    ref = np.random.RandomState(seed).normal(size=n)
    got = p.download_noise()
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
    print(shape, "time %.4f s  max rel %.3g  identical %.4f  >1ulp %d" % (t1 - t0, rel.max(), np.mean(got == ref), np.sum(rel > 2.3e-16)), flush=True)
    p.close()
