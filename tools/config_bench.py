#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations (development tool): 512^3 single realisation (config 1),
1024^3 float64 + lognormal map (config 4/5), 2048^3 on one GPU.  Prints one JSON line per case."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, cosmotools, powertools   # noqa: E402


def run(n, dtype, lognormal=False, reps=5):
    spacing = 2.5
    power = powertools.load_default_power()
    plan = _hip.DevicePlan(n, n, n, dtype)
    plan.set_kgrid(*powertools.ksq_axes(n, n, n, spacing))
    plan.set_power(*powertools.sigma_table(power, (n, n, n), spacing))
    plan.realise(seed=1)
    plan.sync()
    times, kern = [], np.zeros(5)
    growth = np.exp(-0.5 * np.arange(n) / n)
    for i in range(reps):
        plan.sync()
        t0 = time.perf_counter()
        plan.realise(seed=10 + i)
        if lognormal:
            mean, std = plan.moments()
            a_z, b_z = cosmotools.lognormal_tables(growth, std, n)
            plan.lognormal(a_z, b_z, std)
        plan.sync()
        times.append(time.perf_counter() - t0)
        if not lognormal:
            kern += np.array(plan.kernel_ms())
    t = float(np.median(times))
    itemsize = 8 if dtype == np.complex64 else 16
    sweep = itemsize * n * n * (n // 2 + 1)
    alg = 5 * sweep + (2 * (itemsize // 2) * n ** 3 if lognormal else 0)
    out = {"case": "%d^3 %s%s" % (n, "f32" if dtype == np.complex64 else "f64", " + lognormal" if lognormal else ""),
           "ms": round(t * 1e3, 3), "Mcells_s": round(n ** 3 / t / 1e6, 1), "algorithmic_GBs": round(alg / t / 1e9, 1),
           "frac_hbm_peak": round(alg / t / 8e12, 4)}
    if not lognormal:
        out["kernel_ms"] = [round(float(v), 3) for v in kern / reps]
    plan.close()
    print(json.dumps(out), flush=True)


def run_reference(n, reps=3):
    """rng='reference': numpy's MT19937 + polar stream replayed on the GPU, then the exact-chain pipeline."""
    spacing = 2.5
    power = powertools.load_default_power()
    plan = _hip.DevicePlan(n, n, n, np.complex64)
    plan.set_kgrid(*powertools.ksq_axes(n, n, n, spacing))
    plan.set_power(*powertools.sigma_table(power, (n, n, n), spacing))
    plan.reference_noise(1)
    plan.realise(noise="resident")
    plan.sync()
    t_rng, t_all = [], []
    for i in range(reps):
        t0 = time.perf_counter()
        plan.reference_noise(100 + i)
        plan.sync()
        t1 = time.perf_counter()
        plan.realise(noise="resident")
        plan.sync()
        t_rng.append(t1 - t0)
        t_all.append(time.perf_counter() - t0)
    print(json.dumps({"case": "%d^3 f32 rng=reference" % n, "ms": round(float(np.median(t_all)) * 1e3, 3),
                      "ms_mt19937_replay": round(float(np.median(t_rng)) * 1e3, 3),
                      "Mcells_s": round(n ** 3 / float(np.median(t_all)) / 1e6, 1),
                      "kernel_ms": [round(float(v), 3) for v in plan.kernel_ms()]}), flush=True)
    plan.close()


def run_lensing(n, reps=3):
    """rf_lensing_potential on an n^3 float32 field."""
    plan = _hip.DevicePlan(n, n, n, np.complex64)
    plan.upload_real(np.random.RandomState(1).normal(size=(n, n, n)).astype(np.float32))
    cot = np.ones(n)
    cot[1:] = 1.0 / (np.arange(1, n) * 2.5)
    plan.lensing_potential(cot, 2.5, n // 32)
    times = []
    for i in range(reps):
        plan.sync()
        t0 = time.perf_counter()
        plan.lensing_potential(cot, 2.5, n // 32)
        plan.sync()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    print(json.dumps({"case": "%d^3 f32 lensing potential" % n, "ms": round(t * 1e3, 3),
                      "GBs_read_plus_write": round(2 * 4 * n ** 3 / t / 1e9, 1)}), flush=True)
    plan.close()


if __name__ == "__main__":
    run(512, np.complex64)
    run(1024, np.complex64)
    run(1024, np.complex128)
    run(1024, np.complex128, lognormal=True)
    run_reference(1024)
    run_lensing(1024)
    run(2048, np.complex64, reps=3)
