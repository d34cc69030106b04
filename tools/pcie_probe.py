"""GPU probe: device -> host download of a 1024^3 float32 field into pageable vs pinned (hipHostRegister) numpy memory."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools
n = 1024
power = powertools.load_default_power()
p = _hip.DevicePlan(n, n, n)
p.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5)); p.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
p.realise(seed=1); p.sync()
out = np.empty((n, n, n + 2), np.float32)
out[:] = 0
for rep in range(2):
    t0 = time.perf_counter(); p.download_real(out, padded=True); t = time.perf_counter() - t0
    print("pageable, padded layout : %.1f ms  %.1f GB/s" % (t * 1e3, 4 * n ** 3 / t / 1e9), flush=True)
hip = ctypes.CDLL("libamdhip64.so")
rc = hip.hipHostRegister(ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(out.nbytes), 0)
print("hipHostRegister rc", rc)
for rep in range(2):
    t0 = time.perf_counter(); p.download_real(out, padded=True); t = time.perf_counter() - t0
    print("pinned,   padded layout : %.1f ms  %.1f GB/s" % (t * 1e3, 4 * n ** 3 / t / 1e9), flush=True)
hip.hipHostUnregister(ctypes.c_void_p(out.ctypes.data))
