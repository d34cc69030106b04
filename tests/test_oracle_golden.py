"""The oracle (oracle/cpu_ref.py) pinned against outputs of the reference's own
functions (tests/golden/*.npz, produced by oracle/make_golden.py) and against
the known-answer values recorded in SURVEY.md section 8c."""
import numpy as np
import pytest

from conftest import golden
from oracle import cpu_ref

STAGE_FILES = [
    "stages_4x6x8_c64.npz", "stages_6x4x12_c64.npz", "stages_16x16x16_c64.npz",
    "stages_32x32x32_c64.npz", "stages_16x32x64_c64.npz",
    "stages_4x6x8_c128.npz", "stages_16x16x16_c128.npz", "stages_32x32x32_c128.npz",
]


@pytest.mark.parametrize("name", STAGE_FILES)
def test_stages_bit_exact_through_symmetrize(name, default_power):
    g = golden(name)
    nx, ny, nz = (int(v) for v in g["shape"])
    dtype = g["kspace"].dtype
    spacing, seed = float(g["spacing"]), int(g["seed"])
    k, Pk = default_power["k"], default_power["Pk"]

    data = cpu_ref.fill_log10k(nx, ny, nz, spacing, dtype)
    assert np.array_equal(data.real, g["log10k"])           # includes -inf at DC
    assert np.isneginf(data.real[0, 0, 0]) and not data.imag.any()

    cpu_ref.tabulate_sigmas(data, k, Pk, spacing)
    assert np.array_equal(data.real, g["sigma"])
    assert data.real[0, 0, 0] == 0

    cpu_ref.randomize(data, seed=seed)
    assert np.array_equal(data, g["randomized"])

    cpu_ref.symmetrize_packed(data)
    assert np.array_equal(data, g["kspace"])
    assert cpu_ref.is_hermitian_packed(data)

    delta = cpu_ref.c2r(data)
    rms = float(g["rms"])
    tol = 1e-6 if dtype == np.complex64 else 1e-13
    assert np.max(np.abs(delta - g["delta"])) <= tol * rms
    assert abs(np.std(delta) - rms) <= tol * rms
    # f64-FFT-then-round variant (numpy 1.x behaviour) agrees within f32 FFT error
    delta64 = cpu_ref.c2r(data, double_fft=True)
    assert np.max(np.abs(delta64 - g["delta"])) <= 3e-6 * rms


@pytest.mark.parametrize("n,tag", [(64, "c64"), (128, "c64"), (64, "c128"), (128, "c128")])
def test_summary_grids(n, tag, default_power):
    g = golden("summary_%d_%s.npz" % (n, tag))
    dtype = np.complex64 if tag == "c64" else np.complex128
    data = cpu_ref.generate_kspace(n, n, n, 2.5, default_power["k"], default_power["Pk"],
                                   seed=123, dtype=dtype)
    assert np.array_equal(data[:, :, 0], g["kspace_plane0"])
    assert np.array_equal(data[:, :, n // 2], g["kspace_nyq"])
    assert np.array_equal(data[::8, ::8, 1::7], g["kspace_sub"])
    delta = cpu_ref.c2r(data)
    rms = float(g["rms"])
    tol = 2e-6 if tag == "c64" else 1e-13
    assert np.max(np.abs(delta[::8, ::8, ::8] - g["sub"])) <= tol * rms
    assert abs(np.std(delta) - rms) <= tol * rms
    assert abs(delta.astype(np.float64).mean() - float(g["mean"])) < 1e-6


@pytest.mark.parametrize("shape", [(2048, 16, 64), (16, 2048, 64), (16, 16, 2048)])
def test_2048_point_axes(shape, default_power):
    """the oracle against the reference's own run with one axis of 2048 points (subsampled fixtures of oracle/make_golden.py):
    bit exact through symmetrise, <= 2e-6 * rms after the float32 FFT"""
    g = golden("axis2048_%dx%dx%d_c64.npz" % shape)
    nx, ny, nz = shape
    sd, sk = (tuple(int(v) for v in g[k]) for k in ("stride_delta", "stride_k"))
    data = cpu_ref.fill_log10k(nx, ny, nz, float(g["spacing"]), np.complex64)
    cpu_ref.tabulate_sigmas(data, default_power["k"], default_power["Pk"], float(g["spacing"]))
    assert np.array_equal(data.real[::sk[0], ::sk[1], ::sk[2]], g["sigma_sub"])
    cpu_ref.randomize(data, seed=int(g["seed"]))
    cpu_ref.symmetrize_packed(data)
    assert np.array_equal(data[::sk[0], ::sk[1], ::sk[2]], g["kspace_sub"])
    assert float(np.max(np.abs(data))) == float(g["kspace_absmax"])
    delta = cpu_ref.c2r(data)
    rms = float(g["rms"])
    assert np.max(np.abs(delta[::sd[0], ::sd[1], ::sd[2]] - g["sub"])) <= 2e-6 * rms
    assert np.max(np.abs(delta[0, 0, :4] - g["first"])) <= 2e-6 * rms and np.max(np.abs(delta[-1, -1, -4:] - g["last"])) <= 2e-6 * rms
    assert abs(np.std(delta) - rms) <= 2e-6 * rms
    assert abs(float((delta.astype(np.float64) ** 2).sum()) - float(g["sumsq"])) <= 1e-5 * float(g["sumsq"])


def test_survey_known_answers(default_power):
    """SURVEY.md 8c spot values (default P(k), spacing 2.5, seed 123, c64)."""
    expect = {
        64: ([-0.87743145, -2.2504132, -2.4692194, -1.7827865],
             [-1.2413877, 0.30402526, 1.9525096, 3.7410448], 2.3093588),
        128: ([-0.28138322, -0.53682196, 0.38453567, -1.8259681],
              [-0.1237278, -2.8513167, -2.2769353, 3.2357025], 2.3160193),
    }
    for n, (first, last, std) in expect.items():
        delta, rms = cpu_ref.generate_delta_field(n, n, n, 2.5, default_power["k"],
                                                  default_power["Pk"], seed=123)
        assert delta.dtype == np.float32 and delta.shape == (n, n, n)
        assert np.allclose(delta[0, 0, :4], first, rtol=0, atol=5e-6)
        assert np.allclose(delta[-1, -1, -4:], last, rtol=0, atol=5e-6)
        assert abs(rms - std) < 5e-6
    # at 32^3: sigma range and log10k range quoted in the survey
    d = cpu_ref.fill_log10k(32, 32, 32, 2.5)
    lk = d.real[np.isfinite(d.real)]
    assert abs(lk.min() - (-1.10491)) < 1e-5 and abs(lk.max() - 0.33777) < 1e-5
    cpu_ref.tabulate_sigmas(d, default_power["k"], default_power["Pk"], 2.5)
    s = d.real[d.real > 0]
    assert abs(s.min() - 106.334) < 1e-3 and abs(s.max() - 3025.888) < 1e-3


def test_noise_stream_definition():
    """MT19937 + polar restatement == numpy RandomState == golden fixture."""
    g = golden("normals_seed123.npz")["normals"]
    assert np.allclose(g[:4], [-1.0856306, 0.99734545, 0.2829785, -1.50629471], atol=1e-7)
    assert np.array_equal(np.random.RandomState(123).normal(size=4096), g)
    mine = cpu_ref.legacy_normals(123, 1500)
    assert np.array_equal(mine, g[:1500])


def test_smoothed_and_gaussian_tables(default_power):
    g = golden("smoothed_16_c64.npz")
    Pk = default_power["Pk"] * np.exp(-(default_power["k"] * float(g["smoothing"])) ** 2)
    assert np.array_equal(Pk, g["smoothed_Pk"])                     # powertools.py:121
    data = cpu_ref.generate_kspace(16, 16, 16, 2.5, default_power["k"], Pk, seed=123)
    assert np.array_equal(data, g["kspace"])
    g = golden("gaussian_16_c64.npz")                                # linear-k table (non-uniform in log k)
    data = cpu_ref.generate_kspace(16, 16, 16, 2.5, g["k"], g["Pk"], seed=123)
    assert np.array_equal(data, g["kspace"])
    assert np.max(np.abs(cpu_ref.c2r(data) - g["delta"])) <= 1e-6 * float(g["rms"])


def test_lognormal_matches_reference():
    for tag, tol in (("f32", 0), ("f64", 0)):
        g = golden("lognormal_%s.npz" % tag)
        out = cpu_ref.lognormal(g["delta"].copy(), 0.3, sigma=2.5)
        assert np.array_equal(out, g["out_scalar"])
        out = cpu_ref.lognormal(g["delta"].copy(), g["growth_z"], sigma=g["sigma_vec"][()])
        assert np.array_equal(out, g["out_vec"])
        assert np.all(out > 0)


def test_potential_matches_reference():
    g = golden("potential_16_c64.npz")
    pot = cpu_ref.potential_kspace(g["kspace"], 2.5)
    assert np.array_equal(pot, g["potential"])
    assert pot[0, 0, 0] == 0


def test_r2c_matches_reference():
    g = golden("r2c_8x16x32_f32.npz")
    spec = cpu_ref.r2c(g["field"])
    assert spec.dtype == np.complex64
    assert np.allclose(spec, g["spectrum"], rtol=0, atol=1e-4)


def test_variance_fixture_matches_analytic():
    """tests/test_generate.py:24-62 through the reference: mean variance within
    1 % of the analytic erf^3 expression."""
    from scipy.special import erf
    g = golden("variance_64.npz")
    spacing, n = 2.5, 64
    kmin, kmax, sigma = (2 * np.pi) / (spacing * n), np.pi / spacing, 2.5 * spacing
    calc = 1.23 / (2 * np.pi) ** 1.5 / sigma ** 3 * (
        erf(kmax * sigma / np.sqrt(2)) ** 3 - erf(kmin * sigma / np.sqrt(2)) ** 3)
    assert abs(g["variances"].mean() - calc) < 0.01 * calc


def test_philox_known_answer():
    """Philox4x32 known-answer vectors from the Random123 distribution (kat_vectors): 10 rounds (the library default) and
    7 rounds (the native stream, cpu_ref.NATIVE_PHILOX_ROUNDS), counter / key all zero, all ones, and the digits of pi."""
    assert cpu_ref.NATIVE_PHILOX_ROUNDS == 7
    pi_lo, pi_hi = np.array([0x85A308D3243F6A88], np.uint64), np.array([0x0370734413198A2E], np.uint64)
    z, f = np.array([0], np.uint64), np.array([0xFFFFFFFFFFFFFFFF], np.uint64)
    for rounds, args, want in ((7, (z, z, 0, 0), [0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48]),
                               (7, (f, f, 0xFFFFFFFF, 0xFFFFFFFF), [0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662]),
                               (7, (pi_lo, pi_hi, 0xA4093822, 0x299F31D0), [0x4DFCCABA, 0x190A87F0, 0xC47362BA, 0xB6B5242A]),
                               (10, (pi_lo, pi_hi, 0xA4093822, 0x299F31D0), [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1])):
        assert [int(v[0]) for v in cpu_ref.philox4x32(*args, rounds=rounds)] == want
    w = cpu_ref.philox4x32_10(np.array([0], np.uint64), np.array([0], np.uint64), 0, 0)
    assert [int(v[0]) for v in w] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    ff = np.array([0xFFFFFFFFFFFFFFFF], np.uint64)
    w = cpu_ref.philox4x32_10(ff, ff, 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(v[0]) for v in w] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    re, im = cpu_ref.philox_normals(123, np.arange(200000, dtype=np.uint64))
    for v in (re, im):
        assert abs(v.mean()) < 0.01 and abs(v.std() - 1) < 0.01
    assert abs(np.mean(re * im)) < 0.01


def test_simps_restatement_against_scipy():
    """The oracle's restatement of scipy.integrate.simps(even='avg') (what generate.py:405 called; removed from
    current scipy): the odd-count branch equals today's scipy.integrate.simpson, the even-count branch equals its
    published definition -- the average of (Simpson on the first N-1 samples + trapezoid on the last interval) and
    (trapezoid on the first interval + Simpson on the last N-1) -- built from scipy's own odd-count rule."""
    from scipy import integrate
    rng = np.random.RandomState(2)
    for N in range(1, 14):
        x = np.cumsum(0.5 + rng.rand(N))                     # non-uniform abscissae too
        y = rng.normal(size=(3, N))
        got = cpu_ref.simps_avg(y, x)
        if N % 2:
            ref = integrate.simpson(y, x=x) if N > 1 else np.zeros(3)
        else:
            a = (integrate.simpson(y[:, :-1], x=x[:-1]) if N > 2 else 0.0) + 0.5 * (x[-1] - x[-2]) * (y[:, -1] + y[:, -2])
            b = 0.5 * (x[1] - x[0]) * (y[:, 0] + y[:, 1]) + (integrate.simpson(y[:, 1:], x=x[1:]) if N > 2 else 0.0)
            ref = 0.5 * (a + b)
        assert np.allclose(got, ref, rtol=1e-13, atol=1e-13), N


@pytest.mark.parametrize("K", [0.0, 2e-6, -3e-6])
@pytest.mark.parametrize("coeffs", [[1.5], [0.2, -0.01], [1.0, 0.03, -2e-4]])
def test_lensing_oracle_closed_form(K, coeffs):
    """The oracle's slice loop + Simpson restatement against analytic integrals (tests/lensing_closed_form.py):
    constant, linear and quadratic potentials, flat and curved models, odd and even sample counts -- a pin of
    generate.py:397-411 that does not rest on this repo's reading of scipy.integrate.simps."""
    import lensing_closed_form as lcf
    nz, h = 48, 2.5
    for i_min in (1, 2, 7):
        DC, DA = lcf.tables(nz, h, K)
        phi = np.polynomial.Polynomial(coeffs)(DC)
        field = np.broadcast_to(phi, (2, 3, nz)).copy()
        psi = cpu_ref.lensing_potential(field, DC, DA, K=K, i_min=i_min)
        want = lcf.expected(coeffs, nz, h, i_min)
        scale = np.max(np.abs(want))
        assert np.max(np.abs(psi - want[None, None, :])) <= 1e-11 * scale
        assert np.all(psi[:, :, :i_min] == 0)
