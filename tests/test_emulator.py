"""CPU emulation of the HIP kernels' phase functions (rf_fft.h / rf_core.h)
against numpy and the oracle.  Runs in the build container (no GPU): catches
index / twiddle / packing / symmetrise mistakes before any GPU minute is spent."""
import numpy as np
import pytest

import emu_util
from conftest import golden
from oracle import cpu_ref

COL_SIZES = [8, 16, 32, 64, 128, 256, 512, 1024, 2048]
ROW_SIZES = [8, 16, 32, 64, 128, 256, 512, 1024]


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("N", COL_SIZES)
def test_col_fft_all_sizes(N, dtype):
    rng = np.random.RandomState(N)
    ncols = 64
    a = (rng.normal(size=(N, ncols)) + 1j * rng.normal(size=(N, ncols))).astype(dtype)
    tol = 3e-6 if dtype == np.complex64 else 1e-13
    for direction in (+1, -1):
        work = a.copy()
        # contiguous columns: element (row, C) at row*ncols + C
        assert emu_util.col_fft(work, N, direction, ncols, ncols, 0, ncols) == 0
        ref = (np.fft.ifft(a.astype(np.complex128), axis=0) * N if direction > 0
               else np.fft.fft(a.astype(np.complex128), axis=0))
        assert np.max(np.abs(work - ref)) <= tol * np.sqrt(N) * 2


def test_col_fft_y_geometry():
    """y-pass addressing: columns are (ix, kz) pairs, rows are iy."""
    nx, ny, nzc = 4, 32, 16
    rng = np.random.RandomState(5)
    a = (rng.normal(size=(nx, ny, nzc)) + 1j * rng.normal(size=(nx, ny, nzc))).astype(np.complex64)
    work = a.copy()
    assert emu_util.col_fft(work, ny, +1, nx * nzc, nzc, ny * nzc, nzc) == 0
    ref = np.fft.ifft(a.astype(np.complex128), axis=1) * ny
    assert np.max(np.abs(work - ref)) < 1e-4


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("M", ROW_SIZES)
def test_row_c2r_all_sizes(M, dtype):
    """c2r along z for every supported nz = 2M (x, y kept at the minimum 8)."""
    nx = ny = 8
    nz = 2 * M
    rng = np.random.RandomState(M)
    ks = (rng.normal(size=(nx, ny, M + 1)) + 1j * rng.normal(size=(nx, ny, M + 1))).astype(dtype)
    cpu_ref.symmetrize_packed(ks)
    out, s1, s2 = emu_util.c2r(ks)
    ref = np.fft.irfftn(ks.astype(np.complex128), s=(nx, ny, nz), axes=(0, 1, 2))
    tol = 4e-6 if dtype == np.complex64 else 1e-13
    assert np.max(np.abs(out - ref)) <= tol * ref.std()
    assert abs(s1 - out.astype(np.float64).sum()) < 1e-6 * out.size
    assert abs(s2 - (out.astype(np.float64) ** 2).sum()) < 1e-6 * s2


@pytest.mark.parametrize("shape", [(8, 8, 16), (16, 16, 16), (32, 32, 32), (16, 32, 64), (64, 32, 16)])
@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_generation_and_fused_realisation(shape, dtype, default_power):
    nx, ny, nz = shape
    k, Pk = default_power["k"], default_power["Pk"]
    xt, st = cpu_ref.sigma_table(k, Pk, nx, ny, nz, 2.5)
    noise = cpu_ref.reference_noise(123, nx * ny * (nz // 2 + 1))
    ref = cpu_ref.generate_kspace(nx, ny, nz, 2.5, k, Pk, seed=123, dtype=dtype)
    ks = emu_util.generate_kspace(nx, ny, nz, 2.5, xt, st, noise=noise, dtype=dtype)
    # identical up to the last ulp of log10 (libm here, numpy's SIMD loop in the oracle)
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(ks - ref)) <= (5e-7 if dtype == np.complex64 else 1e-15) * scale
    assert ks[0, 0, 0] == 0 and cpu_ref.is_hermitian_packed(ks, rtol=0, atol=0)
    # exact structure of the symmetrised planes: same zeros, same conjugate pairs
    assert np.array_equal(ks.imag == 0, ref.imag == 0)
    dref = cpu_ref.c2r(ref, double_fft=True)
    rms = dref.std()
    out, s1, s2 = emu_util.realise(nx, ny, nz, 2.5, xt, st, noise=noise, dtype=dtype)
    tol = 3e-6 if dtype == np.complex64 else 1e-12
    assert np.max(np.abs(out - dref)) <= tol * rms
    n = out.size
    assert abs(np.sqrt(s2 / n - (s1 / n) ** 2) - rms) <= 1e-6 * rms


def test_native_noise_matches_oracle_philox(default_power):
    """Native (Philox + Box-Muller) mode: the emulator's k-space equals the oracle
    chain fed with the oracle's restatement of the same counter-based stream."""
    nx, ny, nz = 16, 16, 32
    k, Pk = default_power["k"], default_power["Pk"]
    xt, st = cpu_ref.sigma_table(k, Pk, nx, ny, nz, 2.5)
    for dtype, tol in ((np.complex128, 1e-12), (np.complex64, 2e-6)):
        noise = cpu_ref.native_noise(321, nx, ny, nz, dtype)
        ref = cpu_ref.generate_kspace(nx, ny, nz, 2.5, k, Pk, noise=noise, dtype=dtype)
        ks = emu_util.generate_kspace(nx, ny, nz, 2.5, xt, st, seed=321, dtype=dtype)
        assert np.max(np.abs(ks - ref)) <= tol * np.max(np.abs(ref))


def test_linear_k_table_and_edges():
    """Non-uniform (linear-k) table of the reference's variance test.  Its first
    knot equals the grid's fundamental |k| exactly, so for the six fundamental
    modes log10|k| sits within one float32 ulp of the table edge: whether they get
    sigma = 0 (outside) or sigma(edge) depends on the last ulp of log10f (numpy's
    SIMD loop, libm and OCML all differ).  Everything else must agree."""
    g = golden("gaussian_16_c64.npz")
    xt, st = cpu_ref.sigma_table(g["k"], g["Pk"], 16, 16, 16, 2.5)
    noise = cpu_ref.reference_noise(123, 16 * 16 * 9)
    ks = emu_util.generate_kspace(16, 16, 16, 2.5, xt, st, noise=noise)
    lk = cpu_ref.fill_log10k(16, 16, 16, 2.5).real.astype(np.float64)
    edge = (np.abs(lk - xt[0]) < 3e-7) | (np.abs(lk - xt[-1]) < 3e-7)
    assert 0 < edge.sum() <= 8
    ref = g["kspace"]
    assert np.max(np.abs(ks - ref)[~edge]) <= 5e-7 * np.max(np.abs(ref))
    # on the edge cells: either the reference's value (0 here) or the edge sigma times the same deviate
    sig_edge = np.float32(st[0])
    nz3 = noise.reshape(16, 16, 9, 2)
    for ix, iy, iz in zip(*np.nonzero(edge)):
        v = ks[ix, iy, iz]
        ok_zero = abs(v - ref[ix, iy, iz]) <= 1e-6 * sig_edge
        ok_edge = abs(abs(v.real) - abs(np.float32(sig_edge * nz3[ix, iy, iz, 0]))) <= 1e-5 * sig_edge or iz in (0, 8)
        assert ok_zero or ok_edge


@pytest.mark.parametrize("shape", [(16, 16, 32), (32, 64, 16), (64, 64, 64)])
def test_fast_native_generation_matches_oracle(shape, default_power):
    """The fast float32 generation path (float32 sigma lookup through per-bin records,
    one Philox call per cell pair) against the oracle's exact chain fed with the
    oracle's restatement of the native stream."""
    nx, ny, nz = shape
    k, Pk = default_power["k"], default_power["Pk"]
    xt, st = cpu_ref.sigma_table(k, Pk, nx, ny, nz, 2.5)
    out, s1, s2 = emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=99)
    noise = cpu_ref.native_noise(99, nx, ny, nz, np.complex64)
    ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, 2.5, k, Pk, noise=noise, double_fft=True)
    assert np.max(np.abs(out - ref)) <= 2e-5 * rms      # float32 Box-Muller angle rounding dominates
    # float64 plans use the same float32 generation, widened: equal to the float32 plan up to FFT rounding, and
    # within the same tolerance of the oracle's float64 restatement of the stream
    out64, t1, t2 = emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=99, dtype=np.float64)
    assert np.max(np.abs(out64 - out)) <= 2e-6 * rms
    noise64 = cpu_ref.native_noise(99, nx, ny, nz, np.complex128)
    ref64, rms64 = cpu_ref.generate_delta_field(nx, ny, nz, 2.5, k, Pk, noise=noise64, dtype=np.complex128)
    assert np.max(np.abs(out64 - ref64)) <= 2e-5 * rms64
    # a linear-k Gaussian table (non-uniform in log k, many knots per decade at high k); its first knot is
    # put below the grid's fundamental mode so that no cell sits on the table edge (see the test above)
    if shape == (64, 64, 64):
        kmin, kmax, sig = (2 * np.pi) / (2.5 * 64), np.pi / 2.5, 2.5 * 2.5
        kk = np.linspace(0.9 * kmin, np.sqrt(3) * kmax, 100)
        pk = 1.23 * np.exp(-0.5 * (kk * sig) ** 2)
        xt, st = cpu_ref.sigma_table(kk, pk, 64, 64, 64, 2.5)
        out, s1, s2 = emu_util.realise_fast(64, 64, 64, 2.5, xt, st, seed=5)
        noise = cpu_ref.native_noise(5, 64, 64, 64, np.complex64)
        ref, rms = cpu_ref.generate_delta_field(64, 64, 64, 2.5, kk, pk, noise=noise, double_fft=True)
        assert np.max(np.abs(out - ref)) <= 2e-5 * rms


@pytest.mark.parametrize("shape", [(2048, 8, 16), (8, 2048, 16), (2048, 8, 64)])
def test_length_2048_passes_as_two_half_transforms(shape, default_power):
    """Float32 x / y passes of length 2048 run as two 1024-point transforms per tile (Col2 in rf_fft.h: even rows
    first, last-pass outputs parked in registers, odd rows second, one radix-2 combine): generation's row -> mode
    mapping, the parked combine and the half-table twiddle fetch, against the oracle chain."""
    nx, ny, nz = shape
    k, Pk = default_power["k"], default_power["Pk"]
    xt, st = cpu_ref.sigma_table(k, Pk, nx, ny, nz, 2.5)
    out, s1, s2 = emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=41)
    noise = cpu_ref.native_noise(41, nx, ny, nz, np.complex64)
    ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, 2.5, k, Pk, noise=noise, double_fft=True)
    assert np.max(np.abs(out - ref)) <= 3e-5 * rms        # (same 2.2e-5 at (1024, 16, 64): Box-Muller angle rounding)
    n = out.size
    assert abs(np.sqrt(s2 / n - (s1 / n) ** 2) - rms) <= 1e-5 * rms


def test_float64_length_1024_generation_as_two_half_transforms(default_power):
    """float64 plans run the length-1024 generation pass as two 512-point transforms per tile (Col2 with FastGenColIO64<…, XS = 2>:
    64 KB tiles, two workgroups per CU instead of one): against the oracle's float64 restatement of the native stream."""
    nx, ny, nz = 1024, 8, 32
    k, Pk = default_power["k"], default_power["Pk"]
    xt, st = cpu_ref.sigma_table(k, Pk, nx, ny, nz, 2.5)
    out, s1, s2 = emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=17, dtype=np.float64)
    noise = cpu_ref.native_noise(17, nx, ny, nz, np.complex128)
    ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, 2.5, k, Pk, noise=noise, dtype=np.complex128)
    assert np.max(np.abs(out - ref)) <= 3e-5 * rms      # (float32 Box-Muller / sigma arithmetic, widened: as the float32 plans)
    out32, t1, t2 = emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=17)
    assert np.max(np.abs(out - out32)) <= 3e-6 * rms    # the same deviates through the float32 transform
    n = out.size
    assert abs(np.sqrt(s2 / n - (s1 / n) ** 2) - rms) <= 1e-5 * rms


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("M", ROW_SIZES)
def test_row_r2c_all_sizes(M, dtype):
    """Forward r2c (rows + forward x / y passes + unpack) for every supported nz = 2M against np.fft.rfftn."""
    nx, ny, nz = 8, 16, 2 * M
    rng = np.random.RandomState(M)
    f = rng.normal(size=(nx, ny, nz)).astype(dtype)
    spec = emu_util.r2c(f)
    ref = np.fft.rfftn(f.astype(np.float64), axes=(0, 1, 2))
    tol = 2e-6 if dtype == np.float32 else 1e-13
    assert np.max(np.abs(spec - ref)) <= tol * np.sqrt(f.size) * 3
    # round trip through the emulated c2r
    back, s1, s2 = emu_util.c2r(spec)
    assert np.max(np.abs(back - f)) <= (5e-6 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("shape", [(8, 8, 8), (16, 8, 32), (8, 64, 16), (32, 16, 128), (16, 16, 1024), (128, 8, 8)])
def test_unpacked_c2c_against_numpy(shape):
    """Plan(packed=False) (transform.py:207-213,266-270): forward = np.fft.fftn, inverse = np.fft.ifftn,
    through the strided x / y passes and the plain row pass (RowC2C) of the device code."""
    rng = np.random.RandomState(11)
    for ct, tol in ((np.complex64, 2e-6), (np.complex128, 1e-14)):
        a = (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(ct)
        fwd = emu_util.c2c(a, inverse=False)
        ref = np.fft.fftn(a.astype(np.complex128))
        assert np.max(np.abs(fwd - ref)) <= tol * np.sqrt(a.size) * 4
        inv = emu_util.c2c(a, inverse=True)
        ref = np.fft.ifftn(a.astype(np.complex128))
        assert np.max(np.abs(inv - ref)) <= 4 * tol
        back = emu_util.c2c(fwd, inverse=True)
        assert np.max(np.abs(back - a)) <= 20 * tol


def test_fast_sigma_lookup_error_bound(default_power):
    """Row T of the fast native generation (powertools.py:125-164 semantics): the float32 per-bin records against the
    exact float64 piecewise-linear interpolation over the whole |k|^2 range of a 1024^3 and a 64^3 grid, including
    values next to bin edges and table knots.  This is the table part of the native path's error budget; the
    hardware transcendentals are bounded by the GPU tests.  Bound: 1e-6 relative; measured 6.0e-7 (the bin
    coordinate u = a log2|k|^2 + b is a float32 of magnitude <= 512, i.e. resolved to 3e-5 of a bin, times the
    <= 2 % change of sigma across a bin)."""
    from oracle import cpu_ref
    k, Pk = default_power["k"], default_power["Pk"]
    for n in (1024, 64):
        xt, st = cpu_ref.sigma_table(k, Pk, n, n, n, 2.5)
        k0 = 2 * np.pi / 2.5
        xlo, xhi = np.log10(k0 / n) - 0.01, np.log10(k0 * np.sqrt(3) / 2) + 0.01
        rng = np.random.RandomState(n)
        # all |k|^2 of the form dk^2 * (integer), a log-uniform sample, and points hugging the table knots
        top = 3 * (n // 2) ** 2                                  # |k|^2 / dk^2 never exceeds this on the grid
        ints = np.unique(np.concatenate([np.arange(1, min(4096, top + 1)), rng.randint(1, top + 1, 200000)]))
        k2 = [(k0 / n) ** 2 * ints, 10 ** rng.uniform(2 * (xlo + 0.011), 2 * (xhi - 0.011), 200000)]
        knots = xt[(xt > xlo + 0.02) & (xt < xhi - 0.02)]
        k2.append(np.concatenate([10 ** (2 * knots) * (1 + e) for e in (-3e-7, -1e-7, 0.0, 1e-7, 3e-7)]))
        k2 = np.concatenate(k2).astype(np.float32)
        fast, exact, nb = emu_util.fast_sigma(xt, st, xlo, xhi, k2)
        assert nb <= 512                                        # the records fit the kernel's LDS table
        rel = np.abs(fast - exact) / np.maximum(np.abs(exact), 1e-300)
        assert np.max(rel) <= 1e-6, "fast sigma lookup off by %.3g relative" % np.max(rel)


GENERIC_SHAPES = [(4, 6, 8), (6, 4, 12), (40, 60, 80), (10, 14, 22), (2, 2, 2), (12, 18, 6), (26, 34, 46), (24, 8, 16),
                  (4096, 2, 4), (2, 6000, 4), (2, 4, 8192)]          # (axes beyond the tiled kernels: whole lines of up to 8192 complex64)


@pytest.mark.parametrize("shape", GENERIC_SHAPES)
def test_generic_mixed_radix_transforms_against_numpy(shape):
    """The non-power-of-two path (csrc/rf_generic.h): the block functions the generic kernels run, executed by one
    "thread" per block.  c2r = np.fft.irfftn incl. its treatment of non-Hermitian DC / Nyquist bins (the reference's
    backend, transform.py:314), r2c = np.fft.rfftn, c2c = fftn / ifftn; line and row counts that do not divide the
    block sizes are part of the case."""
    rng = np.random.RandomState(5)
    nx, ny, nz = shape
    for tile, (ct, rt, tol) in [(t, c) for t in (3, 1) for c in ((np.complex64, np.float32, 3e-6), (np.complex128, np.float64, 3e-14))]:
        emu_util.lib().emu_set_generic_tile(tile)            # 1: a thread stays on one line (the kernels' walk), 3: walks by index
        ks = (rng.normal(size=(nx, ny, nz // 2 + 1)) + 1j * rng.normal(size=(nx, ny, nz // 2 + 1))).astype(ct)
        out, s1, s2 = emu_util.generic_c2r(ks)
        ref = np.fft.irfftn(ks.astype(np.complex128), s=shape, axes=(0, 1, 2))
        assert np.max(np.abs(out - ref)) <= tol * ref.std()
        assert abs(s1 - ref.sum()) <= 10 * tol * ref.std() * ref.size and abs(s2 - (ref ** 2).sum()) <= 10 * tol * (ref ** 2).sum()
        f = rng.normal(size=shape).astype(rt)
        spec = emu_util.generic_r2c(f)
        ref = np.fft.rfftn(f.astype(np.float64))
        assert np.max(np.abs(spec - ref)) <= tol * np.abs(ref).std() * 4
        a = (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(ct)
        for inverse, fn in ((False, np.fft.fftn), (True, np.fft.ifftn)):
            ref = fn(a.astype(np.complex128))
            assert np.max(np.abs(emu_util.generic_c2c(a, inverse) - ref)) <= tol * np.abs(ref).std() * 4
    emu_util.lib().emu_set_generic_tile(3)


@pytest.mark.parametrize("name", ["stages_4x6x8_c64.npz", "stages_6x4x12_c64.npz", "stages_4x6x8_c128.npz"])
def test_generic_c2r_reproduces_reference_fields(name):
    """The reference's own k-space -> delta(x) pairs at its test shapes (tests/test_transform.py:11)."""
    g = golden(name)
    out, s1, s2 = emu_util.generic_c2r(g["kspace"])
    tol = 1e-6 if g["kspace"].dtype == np.complex64 else 1e-14
    assert np.max(np.abs(out - g["delta"])) <= tol * float(g["rms"])
    n = out.size
    assert abs(np.sqrt(s2 / n - (s1 / n) ** 2) - float(g["rms"])) <= 10 * tol * float(g["rms"])


def test_row_table_of_resident_float32_deviates():
    """rng='reference' on complex64 plans: the generation pass reads the deviate pairs where the replay's segments left them.  A
    table with one entry per row (ix, iy) of the stream (rf_core.h make_rowloc, built on the device by mt_rowtab_kernel) says
    where the row's nz/2 + 1 consecutive cells start: `nfirst` of them in segment `seg` from slot `off`, the rest at the start of
    the next segment.  Against numpy.searchsorted cell by cell: rows inside a segment, rows cut by a boundary at every
    position, a short last segment, a single segment; rows beyond the stream point somewhere harmless; segments shorter than a
    row are flagged."""
    import ctypes
    lib = emu_util.lib()
    rng = np.random.RandomState(4)
    u64p = ctypes.POINTER(ctypes.c_ulonglong)
    for nseg, mean, last, nzh in ((4300, 125463, 30000, 513), (7, 3000, 2500, 1025), (1, 5000, 5000, 33), (3, 700, 650, 129)):
        counts = rng.binomial(int(mean / 0.7854) + 1, 0.7854, size=nseg).astype(np.uint64)
        counts[-1] = last
        first = np.ascontiguousarray(np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64))
        cap = int(mean / 0.7854) + 8
        total = int(first[-1])
        nrows = total // nzh
        # rows whose cells straddle every segment boundary, plus random ones, plus the first and the last whole row
        near = np.concatenate([np.arange(int(b) // nzh - 1, int(b) // nzh + 2) for b in first[1:-1]]) if nseg > 1 else np.array([], np.int64)
        rows = np.unique(np.concatenate([rng.randint(0, nrows, size=300), near, [0, nrows - 1]]))
        rows = rows[(rows >= 0) & (rows < nrows)]
        starts = np.ascontiguousarray((rows * nzh).astype(np.uint64))
        out = np.zeros(len(rows) * nzh, np.uint64)
        bad = lib.emu_row_lookup(first.ctypes.data_as(u64p), nseg, ctypes.c_ulonglong(cap), nzh, starts.ctypes.data_as(u64p), len(rows),
                                 out.ctypes.data_as(u64p))
        assert bad == 0
        cells = (starts[:, None] + np.arange(nzh, dtype=np.uint64)[None, :]).reshape(-1)
        seg = np.searchsorted(first, cells, side="right") - 1
        want = seg.astype(np.uint64) * np.uint64(cap) + (cells - first[seg])
        assert np.array_equal(out, want)
        assert nseg == 1 or len(np.unique(seg.reshape(len(rows), nzh).max(1) - seg.reshape(len(rows), nzh).min(1))) == 2   # both kinds of row were tested
        # a row that reaches beyond the last accepted pair (a failed replay: the host reports it): in bounds, at the start of the runs
        beyond = np.array([total - nzh + 1], np.uint64)
        o2 = np.zeros(nzh, np.uint64)
        lib.emu_row_lookup(first.ctypes.data_as(u64p), nseg, ctypes.c_ulonglong(cap), nzh, beyond.ctypes.data_as(u64p), 1, o2.ctypes.data_as(u64p))
        assert np.array_equal(o2, np.arange(nzh, dtype=np.uint64))
    # segments shorter than a row: flagged (the library then keeps the float64 form, or refuses)
    first = np.ascontiguousarray(np.array([0, 40, 70, 130, 400], np.uint64))
    out = np.zeros(100, np.uint64)
    starts = np.array([0], np.uint64)
    assert lib.emu_row_lookup(first.ctypes.data_as(u64p), 4, ctypes.c_ulonglong(300), 100, starts.ctypes.data_as(u64p), 1, out.ctypes.data_as(u64p)) == 1


@pytest.mark.parametrize("shape,dtype", [((64, 64, 64), np.complex64), ((128, 64, 64), np.complex64), ((64, 128, 128), np.complex64),
                                         ((256, 256, 32), np.complex64), ((64, 64, 128), np.complex128), ((128, 128, 64), np.complex128), ((32, 32, 32), np.complex128),
                                         ((512, 16, 64), np.complex64), ((512, 512, 64), np.complex64), ((1024, 1024, 64), np.complex64)])
def test_transposed_intermediate_is_bit_identical(shape, dtype, default_power):
    """The blocked intermediate X [x block][kz tile][ny][rb][TC] (x pass stores contiguous chunks, y pass in place on X, z pass
    gathers X -> W; rf_capi.hip queue_xyz, RF_FLAG_TRANSPOSED_INTERMEDIATE) only moves data: fields are bit-identical to the
    passes on the plain layout."""
    nx, ny, nz = shape
    L = emu_util.lib()
    old = L.emu_set_xposed(1)                    # (off by default, like the product's flag)
    applies = L.emu_xpose_applies(int(dtype == np.complex128), nx, ny, nz)
    L.emu_set_xposed(old)
    k, Pk = default_power["k"], default_power["Pk"]
    xt, st = cpu_ref.sigma_table(k, Pk, nx, ny, nz, 2.5)
    rng = np.random.RandomState(nx + ny + nz)
    ks = (rng.normal(size=(nx, ny, nz // 2 + 1)) + 1j * rng.normal(size=(nx, ny, nz // 2 + 1))).astype(dtype)
    cpu_ref.symmetrize_packed(ks)
    res = {}
    for on, rb in ((1, 64), (0, 64), (2, 16), (3, 0)):          # 64 = the product's block of x rows; 0 = unblocked
        old, oldrb = L.emu_set_xposed(int(on > 0)), L.emu_set_rowblock(rb)
        try:
            res[on] = (emu_util.c2r(ks),) if nx >= 512 and ny >= 512 else (
                emu_util.c2r(ks), emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=7,
                                                        dtype=np.float32 if dtype == np.complex64 else np.float64))
        finally:
            L.emu_set_xposed(old)
            L.emu_set_rowblock(oldrb)
    # (512, 16, 64): x and y tiles of different widths; (32, 32, 32) float64: fewer x rows than the z pass's rows per workgroup
    assert applies == (0 if shape in ((512, 16, 64), (32, 32, 32)) else 1)
    for v in (1, 2, 3):
        for a, b in zip(res[v], res[0]):
            assert np.array_equal(a[0], b[0])
            # (the moments are summed in a different order: the z pass deals its rows to threads differently when it gathers; float32
            # fields accumulate 16 - 32 values per thread in float32 before the float64 reduction: rf_fft.h MomAcc)
            f32 = dtype == np.complex64
            assert abs(a[1] - b[1]) <= (1e-6 if f32 else 1e-9) * max(1.0, a[0].size ** 0.5) * float(np.abs(a[0]).max())
            assert abs(a[2] - b[2]) <= (1e-6 if f32 else 1e-12) * b[2]
    ref = np.fft.irfftn(ks.astype(np.complex128), s=(nx, ny, nz), axes=(0, 1, 2))
    assert np.max(np.abs(res[1][0][0] - ref)) <= (4e-6 if dtype == np.complex64 else 1e-13) * ref.std()


@pytest.mark.parametrize("M,nx,ny,tc,rb,dtype", [(512, 16, 4, 8, 8, np.complex64), (512, 64, 2, 8, 64, np.complex64), (1024, 8, 4, 8, 4, np.complex64),
                                                (256, 32, 2, 16, 16, np.complex64), (512, 8, 2, 8, 8, np.complex128), (128, 64, 2, 8, 32, np.complex64),
                                                (512, 64, 2, 32, 64, np.complex64)])
def test_gathering_z_pass(M, nx, ny, tc, rb, dtype):
    """RowC2R with XGatherRowIO (z pass reading the blocked intermediate, threads dealt so that a wave reads whole chunks of
    NRT segments) against the plain z pass on the same rows: identical outputs."""
    import ctypes
    L = emu_util.lib()
    rng = np.random.RandomState(M + nx)
    Wk = (rng.normal(size=(nx, ny, M)) + 1j * rng.normal(size=(nx, ny, M))).astype(dtype)      # packed rows (slot 0 = DC + i Nyquist)
    # X[xb][kt][iy][r][c] = Wk[xb * rb + r][iy][kt * tc + c]
    X = np.ascontiguousarray(Wk.reshape(nx // rb, rb, ny, M // tc, tc).transpose(0, 3, 2, 1, 4))
    rt = np.float32 if dtype == np.complex64 else np.float64
    out = np.zeros((nx, ny, 2 * M), rt)
    s1, s2 = ctypes.c_double(), ctypes.c_double()
    rc = L.emu_row_c2r_xgather(int(dtype == np.complex128), M, nx, ny, tc, rb, X.ctypes.data_as(ctypes.c_void_p),
                               out.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(0.5), ctypes.byref(s1), ctypes.byref(s2))
    assert rc == 0, rc
    # reference: the c2r of a row of packed data = irfft of the half spectrum with (DC, Nyquist) unpacked, unnormalised * scale
    half = np.empty((nx, ny, M + 1), np.complex128)
    half[..., 1:M] = Wk[..., 1:]
    half[..., 0] = Wk[..., 0].real
    half[..., M] = Wk[..., 0].imag
    ref = np.fft.irfft(half, n=2 * M, axis=-1) * (2 * M) * 0.5
    assert np.max(np.abs(out - ref)) <= (2e-6 if dtype == np.complex64 else 1e-13) * np.abs(ref).max()
    assert abs(s1.value - out.astype(np.float64).sum()) <= 1e-6 * out.size * np.abs(out).max()
    assert abs(s2.value - (out.astype(np.float64) ** 2).sum()) <= 1e-6 * s2.value


@pytest.mark.parametrize("shape,dtype", [((16, 16, 32), np.complex64), ((32, 16, 64), np.complex128), ((64, 64, 64), np.complex64),
                                         ((8, 8, 1024), np.complex128)])     # (rows of 512 complex128: the exp table in the tile's pad slots)
def test_fused_lognormal_pipeline(shape, dtype):
    """The phase functions behind rf_realise_lognormal on the CPU: the accumulating y pass (AccColIO) gives, by Parseval, the rms
    of the field the z pass is about to produce, and the z pass's epilogue (LognormalRowIO) maps it like the reference's
    apply_lognormal_transform followed by the density factor (cosmotools.py:206-221, generate.py:273)."""
    import ctypes
    nx, ny, nz = shape
    L = emu_util.lib()
    rng = np.random.RandomState(9)
    ks = (rng.normal(size=(nx, ny, nz // 2 + 1)) + 1j * rng.normal(size=(nx, ny, nz // 2 + 1))).astype(dtype) * 40
    cpu_ref.symmetrize_packed(ks)
    delta, _, _ = emu_util.c2r(ks)
    growth = np.exp(-0.5 * np.arange(nz) / nz)
    dens = 0.5 + np.arange(nz) / nz
    rt = np.float32 if dtype == np.complex64 else np.float64
    out = np.zeros((nx, ny, nz), rt)
    sig, s1, s2 = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    dp = ctypes.POINTER(ctypes.c_double)
    rc = L.emu_c2r_lognormal(int(dtype == np.complex128), nx, ny, nz, ks.ctypes.data_as(ctypes.c_void_p), growth.ctypes.data_as(dp),
                             dens.ctypes.data_as(dp), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(sig), ctypes.byref(s1), ctypes.byref(s2))
    assert rc == 0, rc
    std = delta.astype(np.float64).std()
    assert abs(delta.astype(np.float64).mean()) < 1e-6 * std              # DC mode 0 -> mean 0: the Parseval sum is the variance
    assert abs(sig.value - std) <= (2e-6 if rt == np.float32 else 1e-13) * std
    want = cpu_ref.scale_z(cpu_ref.lognormal(delta.copy(), growth, sigma=rt(sig.value)), dens)
    assert np.all(out > 0) and np.max(np.abs(out - want) / want) <= (3e-6 if rt == np.float32 else 1e-12)
    assert abs(s1.value - out.astype(np.float64).sum()) <= 1e-6 * out.size * np.abs(out).max()


def test_tile_pairs_give_the_single_tile_field(default_power):
    """ColPair (rf_fft_col.h; the product's float32 generation pass of length 1024): two adjacent tiles per workgroup, the first tile's
    last-pass outputs parked in registers, both tiles stored row by row from the same lane -- and the second tile's first pass built
    without the kz = 0 repair.  Same arithmetic per tile as ColFFT: the emulated field through pairs is bit for bit the field through
    single tiles (this file also runs under ASan / UBSan: the parked store's indexing is checked there)."""
    nx, ny, nz = 1024, 8, 64
    k, Pk = default_power["k"], default_power["Pk"]
    xt, st = cpu_ref.sigma_table(k, Pk, nx, ny, nz, 2.5)
    old = emu_util.lib().emu_set_pairs(1)
    try:
        pairs, s1, s2 = emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=23)
        emu_util.lib().emu_set_pairs(0)
        single, t1, t2 = emu_util.realise_fast(nx, ny, nz, 2.5, xt, st, seed=23)
    finally:
        emu_util.lib().emu_set_pairs(old)
    assert np.array_equal(pairs, single) and (s1, s2) == (t1, t2) and pairs.std() > 0
    noise = cpu_ref.native_noise(23, nx, ny, nz, np.complex64)
    ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, 2.5, k, Pk, noise=noise, double_fft=True)
    assert np.max(np.abs(pairs - ref)) <= 3e-5 * rms


@pytest.mark.parametrize("L,S", [(128, 16), (128, 32), (64, 16), (256, 16), (16, 16)])
def test_sigma_share_row_mapping_is_a_bijection(L, S):
    """FastGenColIOT::share_row: the L butterflies of the generation pass are dealt to the lanes so that butterfly q and its mirror
    L - q (the rows -ix; butterfly 0 pairs with L/2, both their own mirrors) sit half a wave apart and can trade sigma values through
    ds_bpermute.  Every butterfly must be taken exactly once, partners must be S/2 slots apart in the same wave -- for the
    whole-column form (XS = 1) and for both phases of the two-half-transform form (odd phase: mirror L - 1 - q, nobody its own)."""
    f = emu_util.lib().emu_share_row
    h = S // 2
    for phase in (-1, 0, 1):
        rows = [f(jl, L, S, phase) for jl in range(L)]
        assert sorted(rows) == list(range(L)), (phase, rows[:8])
        for jl in range(L):
            sl = jl % S
            if sl < h:
                q, partner = rows[jl], rows[jl + h]
                if phase == 1:
                    assert partner == L - 1 - q
                else:
                    assert partner == (L // 2 if q == 0 else L - q)


def test_generic_in_place_positions_and_fast_division():
    """rf_generic.h: an axis whose radices are all among 2..5 is transformed in place on a line stored in digit-reversed order --
    generic_pos must be a permutation of [0, n) for such n (and the identity otherwise); and the divisions of the stage loops are
    multiply-high by a host-formed reciprocal (FastDiv), exact for every numerator the loops can produce (a < n * TC <= 2**17,
    a * d < 2**32)."""
    import ctypes
    lib = emu_util.lib()
    lib.emu_fastdiv_first_error.restype = ctypes.c_longlong
    for n in (2, 4, 6, 8, 10, 12, 60, 96, 100, 250, 500, 768, 960, 1000, 1024, 3000, 4096, 6000, 8192):
        out = (ctypes.c_int * n)()
        assert lib.emu_generic_positions(n, out) == 1, n
        assert sorted(out) == list(range(n)), n
    for n in (14, 22, 26, 1019 * 2, 7 * 64):
        out = (ctypes.c_int * n)()
        assert lib.emu_generic_positions(n, out) == 0 and list(out) == list(range(n)), n
    for d in (1, 2, 3, 4, 5, 7, 8, 9, 16, 25, 125, 241, 500, 683, 1000, 1024, 4095, 4096, 4097, 8191, 8192):
        amax = min(1 << 17, (1 << 32) // d)
        assert lib.emu_fastdiv_first_error(d, amax) == -1, d


@pytest.mark.parametrize("shape,cap", [((40, 60, 80), 16), ((36, 10, 24), 8), ((6, 100, 16), 12), ((4, 6, 192), 16), ((64, 8, 16), 8), ((8, 8, 128), 16)])
def test_axes_too_long_for_one_line_take_the_four_step_form(shape, cap):
    """Axes longer than a line the LDS holds (the library's cap: 8192 complex64 / 4096 complex128; lowered here so that small grids
    qualify) run as a four-step transform through global memory -- n = n1 n2: sub-line transforms of length n1, twiddles on load,
    sub-line transforms of length n2 stored in natural order -- with the SAME sequences and block functions as the library
    (rf_generic.h generic_*_seq, generic_lines_block, generic_untangle_at / generic_tangle_at): c2r = np.fft.irfftn, r2c = rfftn,
    c2c = fftn / ifftn, every axis long in turn, with line counts that do not divide the block sizes."""
    rng = np.random.RandomState(7)
    nx, ny, nz = shape
    old = emu_util.lib().emu_set_generic_cap(cap)
    try:
        # (tile 3: lines per block that divide nothing, threads walking by index; tile 1: the kernels' own walk -- a thread stays on
        # one line, loads four elements per trip and advances the twiddle index instead of reducing e q mod n per element)
        for tile, (ct, rt, tol) in [(t, c) for t in (3, 1) for c in ((np.complex64, np.float32, 4e-6), (np.complex128, np.float64, 4e-14))]:
            emu_util.lib().emu_set_generic_tile(tile)
            ks = (rng.normal(size=(nx, ny, nz // 2 + 1)) + 1j * rng.normal(size=(nx, ny, nz // 2 + 1))).astype(ct)
            out, s1, s2 = emu_util.generic_c2r(ks)
            ref = np.fft.irfftn(ks.astype(np.complex128), s=shape, axes=(0, 1, 2))
            assert np.max(np.abs(out - ref)) <= tol * ref.std()
            assert abs(s1 - ref.sum()) <= 10 * tol * ref.std() * ref.size and abs(s2 - (ref ** 2).sum()) <= 10 * tol * (ref ** 2).sum()
            f = rng.normal(size=shape).astype(rt)
            spec = emu_util.generic_r2c(f)
            ref = np.fft.rfftn(f.astype(np.float64))
            assert np.max(np.abs(spec - ref)) <= tol * np.abs(ref).std() * 4
            a = (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(ct)
            for inverse, fn in ((False, np.fft.fftn), (True, np.fft.ifftn)):
                ref = fn(a.astype(np.complex128))
                assert np.max(np.abs(emu_util.generic_c2c(a, inverse) - ref)) <= tol * np.abs(ref).std() * 4
    finally:
        emu_util.lib().emu_set_generic_cap(old)
        emu_util.lib().emu_set_generic_tile(3)
