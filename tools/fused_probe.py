"""GPU probe: fused x+y kernel against the separate passes, cell by cell."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
power = powertools.load_default_power()
def field(fused):
    os.environ["RANDOMFIELD_FUSED"] = "1" if fused else "0"
    p = _hip.DevicePlan(n, n, n)
    p.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5)); p.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
    p.realise(seed=7)
    d = p.download_real().copy(); m = p.moments(); p.close()
    return d, m
a, ma = field(False)
b, mb = field(True)
print("std unfused %.6f fused %.6f" % (ma[1], mb[1]))
diff = np.abs(a - b)
print("max diff", diff.max(), "cells differing", int((diff > 1e-6).sum()), "of", diff.size)
# the difference in k space tells which modes are wrong
fa, fb = np.fft.rfftn(a), np.fft.rfftn(b)
dk = np.abs(fa - fb)
bad = dk > 1e-3 * np.abs(fa).max()
print("bad modes", int(bad.sum()))
if bad.any():
    ix, iy, iz = np.nonzero(bad)
    print("kz values with bad modes:", np.unique(iz)[:40], "count", len(np.unique(iz)))
    print("ky values:", np.unique(iy)[:20], "count", len(np.unique(iy)))
    print("kx values:", np.unique(ix)[:20], "count", len(np.unique(ix)))
