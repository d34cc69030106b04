#!/usr/bin/env python3
"""Randomised C-ABI stress on the GPU: random shapes / dtypes / call sequences, every result checked against the
oracle or numpy.  usage: tools/fuzz_api.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref                                    # noqa: E402  (checker)
from randomfield_amd import _hip, cosmotools, powertools      # noqa: E402

SPACING = 2.5
POWER = powertools.load_default_power()


def make(shape, ct):
    nx, ny, nz = shape
    p = _hip.DevicePlan(nx, ny, nz, ct)
    p.set_kgrid(*powertools.ksq_axes(nx, ny, nz, SPACING))
    p.set_power(*powertools.sigma_table(POWER, shape, SPACING))
    return p


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, it, counts = time.time(), 0, {}
    k, Pk = POWER["k"], POWER["Pk"]
    while time.time() - t0 < budget:
        it += 1
        if rng.rand() < 0.35:      # any even shape (nz a multiple of 4 for the generation rows): the generic mixed-radix kernels
            shape = (2 * int(rng.randint(1, 40)), 2 * int(rng.randint(1, 40)), 4 * int(rng.randint(1, 24)))
        else:
            shape = tuple(int(2 ** rng.randint(3, 8)) for _ in range(2)) + (int(2 ** rng.randint(4, 9)),)
        nx, ny, nz = shape
        if nx * ny * nz > 2 ** 22:
            continue
        ct = np.complex64 if rng.rand() < 0.6 else np.complex128
        rt = np.float32 if ct == np.complex64 else np.float64
        tol = 1e-5 if ct == np.complex64 else 1e-11
        p = make(shape, ct)
        M = nx * ny * (nz // 2 + 1)
        field = None
        for _ in range(rng.randint(2, 7)):
            op = rng.choice(["ext", "native", "gen", "r2c", "lognormal", "affine", "potential", "rpot", "batch", "mt", "mt32", "lens", "regen"])
            counts[op] = counts.get(op, 0) + 1
            if op == "ext":
                seed = int(rng.randint(1, 10 ** 6))
                noise = cpu_ref.reference_noise(seed, M)
                p.realise(noise=noise)
                ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, SPACING, k, Pk, noise=noise, dtype=ct, double_fft=True)
                field = p.download_real()
                assert np.max(np.abs(field - ref)) <= tol * rms, ("ext", shape, ct)
                assert abs(p.moments()[1] - rms) <= tol * rms
            elif op == "native":
                seed = int(rng.randint(1, 2 ** 31))
                p.realise(seed=seed)
                field = p.download_real()
                noise = cpu_ref.native_noise(seed, nx, ny, nz, ct)
                ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, SPACING, k, Pk, noise=noise, dtype=ct, double_fft=True)
                assert np.max(np.abs(field - ref)) <= 2e-5 * rms, ("native", shape, ct)
            elif op == "gen":
                seed = int(rng.randint(1, 2 ** 31))
                p.generate(seed=seed)
                ks = p.download_k()
                assert cpu_ref.is_hermitian_packed(ks, rtol=0, atol=0) and ks[0, 0, 0] == 0
                p.execute_c2r()
                field = p.download_real()
                ref = np.fft.irfftn(ks.astype(np.complex128), s=shape, axes=(0, 1, 2))
                assert np.max(np.abs(field - ref)) <= tol * max(ref.std(), 1e-30) * 3
            elif op == "r2c":
                f = rng.normal(size=shape).astype(rt)
                p.upload_real(f)
                p.execute_r2c()
                spec = p.download_k()
                ref = np.fft.rfftn(f.astype(np.float64), axes=(0, 1, 2))
                t = (3e-6 if ct == np.complex64 else 1e-13) * np.sqrt(f.size) * 3
                assert np.max(np.abs(spec - ref)) <= t, ("r2c", shape, ct)
                p.execute_c2r()
                field = p.download_real()
                assert np.max(np.abs(field - f)) <= (3e-5 if ct == np.complex64 else 1e-12)
            elif op == "lognormal" and field is not None:
                growth = np.exp(-0.5 * np.arange(nz) / nz)
                sigma = float(np.std(field.astype(np.float64)))
                if 0 < sigma < 5 and np.max(np.abs(field)) < 20:      # keep exp() in range: this is a stress test, not a physics run
                    a_z, b_z = cosmotools.lognormal_tables(growth, sigma, nz)
                    p.lognormal(a_z, b_z, sigma)
                    got = p.download_real()
                    ref = cpu_ref.lognormal(field.copy(), growth.astype(rt), sigma=rt(sigma))
                    assert np.allclose(got, ref, rtol=2e-5 if ct == np.complex64 else 1e-11, atol=0), ("lognormal", shape)
                    field = got
            elif op == "affine" and field is not None:
                mul = 1.0 + 0.1 * rng.rand(nz)
                add = float(rng.rand())
                p.affine_z(mul, add)
                got = p.download_real()
                ref = (field * mul.astype(rt) + rt(add)).astype(rt)
                assert np.allclose(got, ref, rtol=3e-6 if ct == np.complex64 else 1e-13, atol=1e-6 if ct == np.complex64 else 1e-13)
                field = got
            elif op == "rpot":
                # the reference API's default call: field + delta(k)/k^2 in one go (fused for native noise on float32 plans)
                seed = int(rng.randint(1, 2 ** 31))
                p.realise_potential(seed=seed)
                field = p.download_real()
                noise = cpu_ref.native_noise(seed, nx, ny, nz, ct)
                ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, SPACING, k, Pk, noise=noise, dtype=ct, double_fft=True)
                assert np.max(np.abs(field - ref)) <= 2e-5 * rms, ("rpot field", shape, ct)
                p.load_potential(1.0)
                got = p.download_k()
                kref = cpu_ref.generate_kspace(nx, ny, nz, SPACING, k, Pk, noise=noise, dtype=np.complex128)
                pref = cpu_ref.potential_kspace(kref, SPACING)
                assert np.max(np.abs(got - pref)) <= 2e-5 * max(np.max(np.abs(pref)), 1e-30), ("rpot potential", shape, ct)
            elif op == "regen":
                # calculate_newtonian_potential with the potential formed again inside the generation pass (+ the light-cone
                # factor in the z pass's store), against the oracle's delta(k)/k^2 transformed by numpy
                seed = int(rng.randint(1, 2 ** 31))
                if p.can_regenerate_potential(None):
                    scale = -float(rng.rand() + 0.5)
                    fz = 1.0 / (1.0 + 0.01 * np.arange(nz)) if rng.rand() < 0.5 else None
                    p.realise_scaled_potential(seed=seed, scale=scale, factor_z=fz)
                    field = p.download_real()
                    noise = cpu_ref.native_noise(seed, nx, ny, nz, ct)
                    kref = cpu_ref.generate_kspace(nx, ny, nz, SPACING, k, Pk, noise=noise, dtype=np.complex128)
                    want = np.fft.irfftn(scale * cpu_ref.potential_kspace(kref, SPACING), s=shape, axes=(0, 1, 2))
                    if fz is not None:
                        want = want * fz
                    assert np.max(np.abs(field - want)) <= 3e-5 * max(want.std(), 1e-30), ("regen", shape, ct)
            elif op == "potential":
                seed = int(rng.randint(1, 10 ** 6))
                noise = cpu_ref.reference_noise(seed, M)
                p.generate(noise=noise)
                ks = p.download_k()
                p.save_potential()
                p.load_potential(-2.0)
                got = p.download_k()
                ref = cpu_ref.potential_kspace(ks, SPACING) * ct(-2.0)
                scale = max(np.max(np.abs(ref)), 1e-30)
                assert np.max(np.abs(got - ref)) <= (1e-6 if ct == np.complex64 else 1e-14) * scale, ("potential", shape)
                p.execute_c2r()
                field = p.download_real()
            elif op == "batch":
                n = int(rng.randint(1, 5))
                seeds = rng.randint(1, 2 ** 31, size=n).astype(np.uint64)
                rms = p.realise_batch(seeds)
                last = p.download_real()
                p.realise(seed=int(seeds[-1]))
                assert np.array_equal(p.download_real(), last)
                assert abs(p.moments()[1] - rms[-1]) <= 1e-12 * rms[-1]
                field = last
            elif op == "mt":
                seed = int(rng.randint(0, 2 ** 31))
                p.reference_noise(seed)
                got = p.download_noise()
                ref = np.random.RandomState(seed).normal(size=2 * M)
                assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)) <= 1e-15, ("mt", shape, seed)
                p.realise(noise="resident")
                field = p.download_real()
            elif op == "mt32":
                # the same stream kept as float32 pairs (a request: plans without the fast generation pass keep float64),
                # then the reference's default call from the resident deviates
                seed = int(rng.randint(0, 2 ** 31))
                p.reference_noise(seed, single=True)
                noise = cpu_ref.reference_noise(seed, M)
                ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, SPACING, k, Pk, noise=noise, dtype=ct, double_fft=True)
                p.realise_potential(noise="resident")
                field = p.download_real()
                assert np.max(np.abs(field - ref)) <= tol * rms, ("mt32 field", shape, ct)
                p.load_potential(1.0)
                pref = cpu_ref.potential_kspace(cpu_ref.generate_kspace(nx, ny, nz, SPACING, k, Pk, noise=noise, dtype=np.complex128), SPACING)
                assert np.max(np.abs(p.download_k() - pref)) <= 2e-5 * max(np.max(np.abs(pref)), 1e-30), ("mt32 potential", shape, ct)
            elif op == "lens" and field is not None:
                DA = np.arange(nz) * SPACING * (1 + 0.02 * np.arange(nz) / nz)
                i_min = int(rng.randint(0, nz))
                cot = cpu_ref.cot_k(np.arange(nz) * SPACING, DA, 0.0)
                p.lensing_potential(cot, SPACING, i_min)
                psi = p.download_aux()
                ref = cpu_ref.lensing_potential(field, np.arange(nz) * SPACING, DA, i_min=i_min)
                assert np.max(np.abs(psi - ref)) <= (3e-6 if ct == np.complex64 else 1e-12) * max(np.max(np.abs(ref)), 1e-30)
        p.close()
        if it % 10 == 0:
            print("iteration %d  %.0f s  %s" % (it, time.time() - t0, shape), flush=True)
    print("fuzz ok: %d plans, ops %s" % (it, counts))


if __name__ == "__main__":
    main()
