"""GPU probe: cost of the exact-chain x pass (reference dtype chain) with native and with resident noise."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools
n = 1024
power = powertools.load_default_power()
p = _hip.DevicePlan(n, n, n)
p.set_kgrid(*powertools.ksq_axes(n, n, n, 2.5)); p.set_power(*powertools.sigma_table(power, (n, n, n), 2.5))
def t(label, **kw):
    p.realise(**kw); p.sync(); p.realise(**kw); p.sync()
    print(label, [round(v, 3) for v in p.kernel_ms()], flush=True)
t("fast native          ", seed=1)
p.set_exact_generation(True)
t("exact chain, native  ", seed=1)
p.reference_noise(5)
t("exact chain, resident", noise="resident")
