#!/usr/bin/env python3
"""
ORACLE TOOLING -- build container only (needs /root/reference; never runs on the GPU box).

Generates the golden fixtures under ``tests/golden/`` by driving the
reference's OWN functions (loaded read-only, by file path, from
``/root/reference/randomfield``) in the order of ``generate.py:191-199,218-219``:

    fill_with_log10k -> tabulate_sigmas -> randomize -> symmetrize -> Plan.execute -> np.std

The reference cannot be imported as a package here (it needs astropy and is
python-2-only: implicit relative imports), so its three hot-path modules are
loaded under the bare names they import each other by, after adding the
numpy-1 aliases they use (``np.obj2sctype``, ``np.float_`` ...).  pyFFTW is not
installed, so ``Plan`` selects its numpy backend (``transform.py:262-270``):
the fixtures are numpy 2.2.6 pocketfft results (single precision for c64).

Fixtures hold DATA only: inputs (shape, spacing, seed, P(k) table) and the
reference's outputs.  Usage:  python oracle/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/randomfield"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SEED = 123
SPACING = 2.5


def load_reference():
    sys.dont_write_bytecode = True
    if not hasattr(np, "obj2sctype"):
        np.obj2sctype = lambda rep, default=None: np.dtype(rep).type
    for old, new in [("float_", np.float64), ("complex_", np.complex128), ("float", float),
                     ("int", int), ("longfloat", np.longdouble), ("clongfloat", np.clongdouble)]:
        if not hasattr(np, old):
            setattr(np, old, new)
    for stub in ("astropy", "astropy.cosmology", "astropy.units"):
        sys.modules.setdefault(stub, types.ModuleType(stub))

    def load(name, alias=None):
        spec = importlib.util.spec_from_file_location(alias or name, f"{REF}/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[alias or name] = mod
        spec.loader.exec_module(mod)
        return mod

    transform = load("transform")
    powertools = load("powertools")
    rf_random = load("random", alias="rf_random")
    cosmotools = load("cosmotools", alias="rf_cosmotools")
    return transform, powertools, rf_random, cosmotools


def gaussian_power(n, spacing):
    """The P(k) of the reference's variance test (tests/test_generate.py:41-52):
    linear k grid, 100 rows, Gaussian."""
    kmin = (2 * np.pi) / (spacing * n)
    kmax = np.pi / spacing
    sigma = 2.5 * spacing
    power = np.empty(100, dtype=[("k", float), ("Pk", float)])
    power["k"] = np.linspace(kmin, np.sqrt(3) * kmax, len(power))
    power["Pk"] = 1.23 * np.exp(-0.5 * (power["k"] * sigma) ** 2)
    return power


def run_stages(ref, shape, power, dtype, seed=SEED, spacing=SPACING, smoothing=0.0):
    transform, powertools, rf_random, _ = ref
    plan = transform.Plan(shape=shape, dtype_in=dtype)
    B = plan.data_in
    powertools.fill_with_log10k(B, spacing=spacing, packed=True)
    log10k = B.real.copy()
    smoothed = powertools.filter_power(power, smoothing)
    powertools.tabulate_sigmas(B, power=smoothed, spacing=spacing, packed=True)
    sigma = B.real.copy()
    rf_random.randomize(B, seed=seed)
    randomized = B.copy()
    transform.symmetrize(B, packed=True)
    kspace = B.copy()
    delta = plan.execute()
    rms = np.std(delta.flat)
    return dict(log10k=log10k, sigma=sigma, randomized=randomized, kspace=kspace,
                delta=np.ascontiguousarray(delta), rms=np.asarray(rms),
                smoothed_Pk=smoothed["Pk"].copy())


AXIS2048_SHAPES = [(2048, 16, 64), (16, 2048, 64), (16, 16, 2048)]


def axis2048_strides(shape):
    """subsampling strides of the real field (x, y, z) and of k space (kx, ky, kz) for the long-axis fixtures: 128 samples along
    the 2048-point axis, 8 / 16 along the short ones; every kz where nz is short, else every 16th (0 and nz/2 included)"""
    sd = tuple(16 if n == 2048 else (2 if n == 16 else 4) for n in shape)
    sk = (sd[0], sd[1], 16 if shape[2] == 2048 else 1)
    return sd, sk


def summarise_axis2048(st, shape):
    sd, sk = axis2048_strides(shape)
    d = st["delta"]
    d64 = d.astype(np.float64)
    return dict(shape=np.array(shape), spacing=SPACING, seed=SEED, stride_delta=np.array(sd), stride_k=np.array(sk),
                sub=d[::sd[0], ::sd[1], ::sd[2]].copy(), first=d[0, 0, :4].copy(), last=d[-1, -1, -4:].copy(), rms=st["rms"],
                mean=np.asarray(d64.mean()), min=np.asarray(d.min()), max=np.asarray(d.max()), sumsq=np.asarray((d64 ** 2).sum()),
                kspace_sub=st["kspace"][::sk[0], ::sk[1], ::sk[2]].copy(), kspace_absmax=np.asarray(np.max(np.abs(st["kspace"]))),
                sigma_sub=st["sigma"][::sk[0], ::sk[1], ::sk[2]].copy())


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    transform, powertools, rf_random, cosmotools = ref
    power = powertools.load_default_power()

    # the P(k) table itself (data, 500 x 2 float64)
    np.savez_compressed(os.path.join(OUT, "default_power.npz"), k=power["k"], Pk=power["Pk"])

    # --- full stage-by-stage fixtures at small shapes ---------------------
    for shape in [(4, 6, 8), (6, 4, 12), (16, 16, 16), (32, 32, 32), (16, 32, 64)]:
        for dtype, tag in [(np.complex64, "c64"), (np.complex128, "c128")]:
            if tag == "c128" and shape not in [(4, 6, 8), (16, 16, 16), (32, 32, 32)]:
                continue
            st = run_stages(ref, shape, power, dtype)
            name = "stages_%dx%dx%d_%s.npz" % (shape + (tag,))
            np.savez_compressed(os.path.join(OUT, name), shape=np.array(shape), spacing=SPACING,
                                seed=SEED, **{k: v for k, v in st.items() if k != "smoothed_Pk"})

    # --- smoothing (filter_power) + Gaussian table of the variance test ----
    st = run_stages(ref, (16, 16, 16), power, np.complex64, smoothing=3.0)
    np.savez_compressed(os.path.join(OUT, "smoothed_16_c64.npz"), shape=np.array((16, 16, 16)),
                        spacing=SPACING, seed=SEED, smoothing=3.0, smoothed_Pk=st["smoothed_Pk"],
                        kspace=st["kspace"], delta=st["delta"], rms=st["rms"])
    gp = gaussian_power(16, SPACING)
    st = run_stages(ref, (16, 16, 16), gp, np.complex64)
    np.savez_compressed(os.path.join(OUT, "gaussian_16_c64.npz"), shape=np.array((16, 16, 16)),
                        spacing=SPACING, seed=SEED, k=gp["k"], Pk=gp["Pk"], sigma=st["sigma"],
                        kspace=st["kspace"], delta=st["delta"], rms=st["rms"])

    # --- larger grids: subsampled + summary statistics ---------------------
    for n in (64, 128):
        for dtype, tag in [(np.complex64, "c64"), (np.complex128, "c128")]:
            st = run_stages(ref, (n, n, n), power, dtype)
            d = st["delta"]
            d64 = d.astype(np.float64)
            np.savez_compressed(
                os.path.join(OUT, "summary_%d_%s.npz" % (n, tag)), shape=np.array((n, n, n)),
                spacing=SPACING, seed=SEED, sub=d[::8, ::8, ::8].copy(), first=d[0, 0, :4].copy(),
                last=d[-1, -1, -4:].copy(), rms=st["rms"], mean=np.asarray(d64.mean()),
                min=np.asarray(d.min()), max=np.asarray(d.max()), sumsq=np.asarray((d64 ** 2).sum()),
                kspace_plane0=st["kspace"][:, :, 0].copy(),
                kspace_nyq=st["kspace"][:, :, n // 2].copy(),
                kspace_sub=st["kspace"][::8, ::8, 1::7].copy())

    # --- one axis of 2048 points (the longest the tiled kernels serve; BASELINE config 4's axis length), the other two short:
    # the reference's run, subsampled (a value of delta depends on every mode of its line, so a subsample pins the whole transform)
    for shape in AXIS2048_SHAPES:
        st = run_stages(ref, shape, power, np.complex64)
        np.savez_compressed(os.path.join(OUT, "axis2048_%dx%dx%d_c64.npz" % shape), **summarise_axis2048(st, shape))

    # --- variance test of tests/test_generate.py:24-62 through the reference
    n = 64
    gp = gaussian_power(n, SPACING)
    variances = []
    for trial in range(10):
        st = run_stages(ref, (n, n, n), gp, np.complex64, seed=SEED + trial)
        variances.append(np.var(st["delta"]))
    np.savez_compressed(os.path.join(OUT, "variance_64.npz"), k=gp["k"], Pk=gp["Pk"],
                        variances=np.array(variances, np.float64), spacing=SPACING, seed=SEED)

    # --- noise stream pin: first values of RandomState(123).normal ---------
    np.savez_compressed(os.path.join(OUT, "normals_seed123.npz"),
                        normals=np.random.RandomState(SEED).normal(size=4096))

    # --- lognormal map (cosmotools.py:206-221; inputs as tests/test_cosmotools.py:87-94)
    np.random.seed(SEED)
    for rt, tag in [(np.float32, "f32"), (np.float64, "f64")]:
        delta = np.empty((16, 16, 32), dtype=rt)
        delta[:] = 2.5 * np.random.normal(size=delta.shape)
        inp = delta.copy()
        growth_z = np.exp(-0.5 * np.arange(32) / 32.0)
        out_scalar = cosmotools.apply_lognormal_transform(inp.copy(), 0.3, sigma=2.5)
        out_vec = cosmotools.apply_lognormal_transform(inp.copy(), growth_z, sigma=rt(np.std(inp)))
        np.savez_compressed(os.path.join(OUT, "lognormal_%s.npz" % tag), delta=inp, growth_z=growth_z,
                            out_scalar=out_scalar, out_vec=out_vec, sigma_vec=np.asarray(rt(np.std(inp))))

    # --- save_potential branch (generate.py:200-217), reproduced with the
    # reference's own helper create_ksq_grids + the same numpy calls
    shape = (16, 16, 16)
    plan = transform.Plan(shape=shape, dtype_in=np.complex64)
    B = plan.data_in
    powertools.fill_with_log10k(B, spacing=SPACING, packed=True)
    powertools.tabulate_sigmas(B, power=power, spacing=SPACING, packed=True)
    rf_random.randomize(B, seed=SEED)
    transform.symmetrize(B, packed=True)
    pot = np.empty_like(B)
    pot.imag = 0.0
    kx2, ky2, kz2 = powertools.create_ksq_grids(pot, spacing=SPACING, packed=True)
    np.add(kx2, ky2, out=pot.real)
    pot.real += kz2
    with np.errstate(divide="ignore"):
        np.reciprocal(pot.real, out=pot.real)
    pot[0, 0, 0] = 0.0
    pot *= B
    np.savez_compressed(os.path.join(OUT, "potential_16_c64.npz"), shape=np.array(shape),
                        spacing=SPACING, seed=SEED, kspace=B.copy(), potential=pot)

    # --- r2c forward / round trip pins (transform.py:199-206,270)
    np.random.seed(SEED)
    f = np.random.normal(size=(8, 16, 32)).astype(np.float32)
    planf = transform.Plan(shape=f.shape, dtype_in=np.float32, inverse=False, packed=True)
    planf.data_in[:] = f
    spec = planf.execute().copy()
    np.savez_compressed(os.path.join(OUT, "r2c_8x16x32_f32.npz"), field=f, spectrum=spec)

    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("wrote fixtures to %s (%.1f KB)" % (os.path.normpath(OUT), total / 1024.0))


if __name__ == "__main__":
    main()
