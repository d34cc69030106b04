#!/usr/bin/env python3
"""
ORACLE TOOLING -- build container only (needs /root/reference; never runs on the GPU box).

Summary fixtures of the REFERENCE's own path at the BASELINE sizes (256^3, 512^3, 1024^3; complex64, and complex128 where the
container's memory allows), made exactly like ``oracle/make_golden.py`` makes the small ones: the reference's
``transform.py`` / ``powertools.py`` / ``random.py`` loaded by file path and driven in the order of
``generate.py:191-199,218-219`` (fill_with_log10k -> tabulate_sigmas -> randomize -> symmetrize -> Plan.execute -> np.std),
default P(k), spacing 2.5, seed 123.  Only subsamples and statistics are kept (DATA, a few hundred KB):

    sub            delta[::s, ::s, ::s], s = n / 16  -> 4096 values
    first, last    delta[0, 0, :4], delta[-1, -1, -4:]          (the SURVEY section 8c spot values)
    rms, mean, min, max, sumsq
    plane0_sub, nyq_sub   the two Hermitian planes of k space after symmetrize, [::s, ::s]
    kspace_sub     k space after symmetrize [::s, ::s, 1::(n/2)//8]

usage: python oracle/make_golden_large.py [256 512 1024] [--c128]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import load_reference, OUT, SEED, SPACING   # noqa: E402


def summarise(ref, n, dtype, power):
    transform, powertools, rf_random, _ = ref
    t0 = time.time()
    plan = transform.Plan(shape=(n, n, n), dtype_in=dtype)
    B = plan.data_in
    powertools.fill_with_log10k(B, spacing=SPACING, packed=True)
    powertools.tabulate_sigmas(B, power=powertools.filter_power(power, 0.0), spacing=SPACING, packed=True)
    rf_random.randomize(B, seed=SEED)
    transform.symmetrize(B, packed=True)
    s = n // 16
    out = dict(shape=np.array((n, n, n)), spacing=SPACING, seed=SEED,
               plane0_sub=B[::s, ::s, 0].copy(), nyq_sub=B[::s, ::s, n // 2].copy(),
               kspace_sub=B[::s, ::s, 1::max(1, (n // 2) // 8)].copy())
    delta = plan.execute()
    out["rms"] = np.asarray(np.std(delta.flat))                      # generate.py:219, as the reference computes it
    out["sub"] = delta[::s, ::s, ::s].copy()
    out["first"] = delta[0, 0, :4].copy()
    out["last"] = delta[-1, -1, -4:].copy()
    out["min"] = np.asarray(delta.min())
    out["max"] = np.asarray(delta.max())
    acc = np.zeros(2, np.float64)
    for ix in range(n):                                              # float64 moments plane by plane (no 8 GB temporary)
        p = delta[ix].astype(np.float64)
        acc += (p.sum(), (p * p).sum())
    out["mean"] = np.asarray(acc[0] / delta.size)
    out["sumsq"] = np.asarray(acc[1])
    out["seconds"] = np.asarray(time.time() - t0)
    return out


def main():
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [256, 512, 1024]
    dtypes = [(np.complex64, "c64")] + ([(np.complex128, "c128")] if "--c128" in sys.argv else [])
    ref = load_reference()
    power = ref[1].load_default_power()
    for n in sizes:
        for dtype, tag in dtypes:
            st = summarise(ref, n, dtype, power)
            name = os.path.join(OUT, "summary_%d_%s.npz" % (n, tag))
            np.savez_compressed(name, **st)
            print("%s: %.0f s, rms %.7f first %s" % (os.path.basename(name), float(st["seconds"]), float(st["rms"]), st["first"]), flush=True)


if __name__ == "__main__":
    main()
