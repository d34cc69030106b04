// rf_fft_gen.h -- the generation-fused IOs of the x pass: exact chain (GenColIO), fast float32 generation (FastGenColIOT), its float64 form (FastGenColIO64) (part of rf_fft.h: include that)
#pragma once
#include "rf_fft.h"

namespace rf {

// x pass fused with generation (rows K,T,R,S): load() synthesises the packed
// k-space cell instead of reading memory.  Columns are the flattened (iy, kz).
// If `kspace` is non-null the cell is read from an API-layout array
// [nx][ny][nz/2+1] instead (unfused c2r of uploaded / separately generated data).
template <typename T, bool WIDE = false> struct GenColIO {
  cplx<T>* base;           // destination W
  ColGeom g;               // x-pass geometry: inner = ny*nzc, row_stride = ny*nzc
  GenParams gp;
  const cplx<T>* kspace;   // optional source in API layout
  int kz0, nzl;            // this rank's kz slab [kz0, kz0 + nzl) of the nz/2 packed planes
  RF_HD V16<T> load(long long C0, int cl, int rb, int ro) const {
    V16<T> v;
    const long long C = C0 + cl;
    const int nzc = gp.nz / 2;
    const uint64_t seed = gp.seed;
    const int ix = rb + ro;
#pragma unroll
    for (int c = 0; c < V16<T>::CPL; ++c) {
      const long long Cc = C + c;
      const int iy = (int)(Cc / nzl), kz = kz0 + (int)(Cc % nzl);
      if (kspace) {
        const cplx<T>* p = kspace + ((long long)ix * gp.ny + iy) * gp.zpitch;      // rows of this rank's planes + Nyquist
        cplx<T> a = p[kz - gp.zoff];
        if (kz == 0) {
          // The planes kz = 0 and kz = nz/2 travel as ONE complex plane (a + i n), which needs both to be 2-D
          // Hermitian.  np.fft.irfftn (transform.py:314) accepts anything there and, by discarding the imaginary part
          // after the x and y transforms, in effect uses the Hermitian part of each plane: so that is what is packed.
          // (Hermitian input, e.g. after symmetrize(), is reproduced bit for bit: (a + conj a*)/2 with a == conj a*.)
          const int mx = (gp.nx - ix) % gp.nx, my = (gp.ny - iy) % gp.ny;
          const cplx<T>* pm = kspace + ((long long)mx * gp.ny + my) * gp.zpitch;
          const cplx<T> am = pm[0], n0 = p[gp.zpitch - 1], nm = pm[gp.zpitch - 1];
          const cplx<T> ah = mk<T>((T)0.5 * (a.x + am.x), (T)0.5 * (a.y - am.y));
          const cplx<T> nh = mk<T>((T)0.5 * (n0.x + nm.x), (T)0.5 * (n0.y - nm.y));
          a = mk<T>(ah.x - nh.y, ah.y + nh.x);
        }
        v.c[c] = a;
      } else {
        v.c[c] = gen_packed<T>(gp, seed, ix, iy, kz);
      }
    }
    return v;
  }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<T>& v) const {
    v16_store<T>(g.at<WIDE>(base, C0, cl, rb, ro), v);
  }
  static constexpr int FIX_MODE = 0;
  RF_HD bool needs_fix(long long) const { return false; }
  RF_HD cplx<T> fix_value(long long, int, int) const { return cplx<T>(); }
  static constexpr int LDS_EXTRA = 0;
  RF_HD void prologue(int, int, void*) {}
  // the kernel calls this once before any load(): a seed kept in device memory (graph replay) is read once, through
  // the scalar unit, instead of once per cell
  RF_HD void bind_seed() { if (gp.seed_dev) { gp.seed = pin_uniform(*gp.seed_dev); gp.seed_dev = nullptr; } }
  RF_HD static void sched_fence(int = 0) {}
  // the exact-chain generation body (float64 lookups, libm-grade log10 / sin / cos) is far too big to be
  // replicated R times: the load loop stays rolled and parks its values in the thread's own LDS slots
  static constexpr bool ROLLED_LOAD = true;
  RF_HD long long remap_tile(long long t) const { return t; }
  static constexpr bool HAS_FINISH = false;
};

// x pass fused with the fast float32 native generation (one Philox call per lane load)
// SLAB: 1 = only rows [x0, x1) are stored (replicated-generation mode); a separate instantiation so that the
// guard costs the ordinary kernel nothing.
// FIX: 1 = this kernel repairs the kz = 0 slot itself (the owning lane, rolled loop through LDS: short passes, and the emulator's
// reference form); 3 = it takes the repaired slots from a side buffer [ny][nx] that fix_fill_kernel (rf_kernels.h: one thread per
// mode, every lane busy, the same fix_value() arithmetic) has filled just before -- 8 extra loads per owning lane instead of two
// Philox calls, Box-Muller pairs and sigma lookups per row in a kernel that then needs 128 - 244 registers and runs its tiles
// 2.5 - 19x slower than an ordinary one (rounds 1 - 3: FIX = 1, then FIX = 2 = the values computed by all lanes in a phase of
// their own); 0 = it does not repair (the tiles that hold kz = 0 are run by a FIX = 1 / 3 launch first).
// POT: 2 = the pass transforms pscale * delta(k) / k^2 instead of delta(k) (the saved potential regenerated on demand: each
// cell rounded as the stored one and its scaled copy would be); 1 = the pass also stores delta(k) / k^2 (0 at DC) of every generated cell into `pot`, an API-layout array
// [nx][ny][nz/2+1] -- the save_potential=True branch of generate_delta_field (generate.py:200-217) without ever
// materialising delta(k) itself.  (Rows of nz/2+1 cells are only 8-byte aligned: two 8-byte stores per lane.)
// SRC: 0 = native Philox + Box-Muller draws; 1 = deviates resident in device memory as float64 (2 = as float32 pairs;
// the reference's numpy stream,
// rng='reference'): same float32 |k| and sigma arithmetic, the draw replaced by two 16-byte loads per lane.  The field
// then differs from the exact-chain kernel's by the float32 sigma rounding only (<= 1e-6 relative, far inside the
// 1e-5 * rms parity tolerance) and the pass is HBM-bound (12.9 GB) instead of latency-bound on table lookups.
// XS: 1 = row r of the pass is mode ix = r; 2 = the pass is one HALF of a transform of twice its length (Col2 below: rows of the
// even / odd modes ix = 2 r + xp, xp = the phase set_phase() selects) -- native generation without the potential store only.
template <int FIX = 1, int SLAB = 0, int POT = 0, int SRC = 0, int XS = 1>
struct FastGenColIOT {
  static_assert(XS == 1 || (XS == 2 && SLAB == 0 && POT != 1 && SRC != 1), "half-transform rows: native generation or float32 deviate pairs, no potential store");
  static constexpr int NOISE_SRC = SRC;      // (0 native, 1 float64 deviates, 2 float32 pairs in the replay's runs)
  int xp = 0;
  RF_HD void set_phase(int p) { xp = p; }
  cplx<float>* base;
  ColGeom g;
  FastGenParams gp;
  cplx<float>* pot = nullptr;
  int kz0, nzl;
  int x0 = 0, x1 = 1 << 30;  // replicated-generation mode (multi-GPU without an exchange): only rows [x0, x1) are stored,
                             // and `base` has been moved back by x0 rows so that row x0 lands on the local array's row 0
  RF_HD int nzl_shift() const { return 31 - __builtin_clz((unsigned)nzl); }
  const FastRec* rec;      // set by prologue(): LDS copy of the sigma records (or the global one)
  static constexpr int LDS_EXTRA = FAST_LDS_BINS * (int)sizeof(FastRec);
  // (a scheduling fence between the R generation bodies of a butterfly was measured every 1, 2 and 4 rows: no gain with the max-ILP
  // strategy this file is compiled with; the hook stays because ColFFT calls it on every IO)
  RF_HD static void sched_fence(int = 0) {}
  // stage the sigma records in LDS (every thread copies its share; the kernel barriers afterwards)
  // (the host only selects this kernel when nbins <= FAST_LDS_BINS, so `rec` is always an LDS pointer
  // and the lookups compile to ds_read_b128, not flat loads)
  RF_HD void prologue(int tid, int nthreads, void* lds_extra) {
    FastRec* l = reinterpret_cast<FastRec*>(lds_extra);
    for (int i = tid; i < gp.nbins && i < FAST_LDS_BINS; i += nthreads) l[i] = gp.rec[i];
    rec = l;
  }
  RF_HD void bind_seed() { if (gp.seed_dev) { gp.seed = pin_uniform(*gp.seed_dev); gp.seed_dev = nullptr; } }
  // Cell pair (kz, kz + 1) of column (ix = rb + ro, iy).  What does not depend on m (= ro / L) is a common
  // subexpression of the R unrolled loads, and what does not depend on the lane runs on the scalar ALU: the
  // Philox counter is (lane part) + (uniform part), two vector adds per load instead of a 64-bit multiply chain.
  RF_HD V16<float> load_impl(long long C0, int cl, int rb, int ro, const V16<float>* raw) const {
    V16<float> v;
    const long long C = C0 + cl;
    const uint64_t seed = gp.seed;                                                      // bind_seed() ran first
    // nzl = (nz/2) / ranks is a power of two (the launcher checks it): shift and mask instead of a 64-bit division
    const int iy = (int)((unsigned)C >> nzl_shift()), kz = kz0 + (int)((unsigned)C & (unsigned)(nzl - 1));   // lane, m-invariant
    const uint64_t half_plane = ((uint64_t)gp.ny * (uint64_t)(gp.nz / 2)) >> 1;        // counters per unit of ix
    // mode index ix = XS (rb + ro) + xp = (lane part rbt) + (uniform part rot)
    const int rbt = XS * rb, rot = XS * ro + (XS == 2 ? xp : 0);
    const uint64_t ctr_l = (uint64_t)rbt * half_plane + (((uint64_t)iy * (uint64_t)(gp.nz / 2) + (uint64_t)kz) >> 1);
    const uint64_t ctr_u = pin_uniform((uint64_t)rot * half_plane);
    // signed fftfreq index: XS rb < XS L <= nx/2 and XS ro is a multiple of XS L, so the wrap depends on ro alone
    const int ro_s = XS * ro >= (gp.nx >> 1) ? rot - gp.nx : rot;
    const float kx = (float)(rbt + ro_s) * gp.dkx, ky = (float)fast_signed_index(iy, gp.ny) * gp.dky;
    const float kxy = fmaf(kx, kx, ky * ky);                                           // == fast_kxy2(gp, rb + ro, iy)
    const float k2a = fast_k2(gp, kxy, kz), k2b = fast_k2(gp, kxy, kz + 1);
    if (SRC == 0) {
      fast_gen_pair_at(gp, rec, seed, ctr_l + ctr_u, k2a, k2b, v.c[0], v.c[1]);
    } else if (SRC == 1) {
      // cells (ix, iy, kz) and (ix, iy, kz + 1) are adjacent in the reference's order: 4 doubles, 32 contiguous bytes
      const int nzp = gp.zpitch;
      const double* d = (gp.noise + 2LL * ro * gp.ny * nzp) + 2u * (uint32_t)((rb * gp.ny + iy) * nzp + (kz - gp.zoff));
      const V16<double> ga = v16_load<double>(d), gb = v16_load<double>(d + 2);     // one complex128 = one deviate pair
      const double sa = (double)fast_sigma(gp, rec, k2a), sb = (double)fast_sigma(gp, rec, k2b);
      v.c[0] = mk<float>((float)(sa * ga.c[0].x), (float)(sa * ga.c[0].y));
      v.c[1] = mk<float>((float)(sb * gb.c[0].x), (float)(sb * gb.c[0].y));
    } else {
      // float32 pairs where the one-pass replay left them: the row's entry of the row table (rf_core.h RowLoc; index iy nx + ix =
      // a lane part that is the same for all R rows of a butterfly + the uniform row offset) says where its cells start; cells kz
      // and kz + 1 are neighbours unless a segment ends between them
      RowLoc e;
      if (raw) { e.off = 0; e.seg_n = 0; }
      else e = load_rowloc((gp.rowtab + rot) + (uint32_t)(iy * gp.nx + rbt));
      cplx<float> ga, gb;
      if (raw) { ga = raw->c[0]; gb = raw->c[1]; }                                       // (loaded by preload() at the top of the kernel)
      else { ga = load_pair_global(row_pair(gp, e, kz)); gb = load_pair_global(row_pair(gp, e, kz + 1)); }
      const float sa = fast_sigma(gp, rec, k2a), sb = fast_sigma(gp, rec, k2b);
      v.c[0] = mk<float>(sa * ga.x, sa * ga.y);
      v.c[1] = mk<float>(sb * gb.x, sb * gb.y);
    }
    if (POT == 2) {
      const float ra = fast_rcp(k2a), rb2 = fast_rcp(k2b), ps = (float)gp.pscale;      // (slot kz = 0, where k2a may be 0, is replaced by fix_value())
      v.c[0] = mk<float>((v.c[0].x * ra) * ps, (v.c[0].y * ra) * ps);
      v.c[1] = mk<float>((v.c[1].x * rb2) * ps, (v.c[1].y * rb2) * ps);
    }
    if (POT == 1) {
      // row (ix, iy) of the API layout; the slot kz = 0 (Hermitian planes) is written by fix_value() instead
      // rows of gp.ppitch (even) cells: the pair (kz even, kz + 1) is one aligned 16-byte store
      const int nzp = gp.ppitch, sl = kz - gp.zoff;
      cplx<float>* row = (pot + (long long)ro * gp.ny * nzp) + (uint32_t)((rb * gp.ny + iy) * nzp);
      const float ra = fast_rcp(k2a), rb2 = fast_rcp(k2b);
      V16<float> q;
      q.c[0] = mk<float>(v.c[0].x * ra, v.c[0].y * ra);
      q.c[1] = mk<float>(v.c[1].x * rb2, v.c[1].y * rb2);
      if (FIX != 0 && kz == 0) row[sl + 1] = q.c[1];          // slot kz = 0 itself: written by fix_value()
      else v16_store<float>(row + sl, q);
    }
    return v;
  }
  // ---- sigma shared between the rows +-ix (round 5) ------------------------------------------------------------------------------
  // |k|^2 of a cell depends on kx^2 only, and the R rows j + m L of a first-pass butterfly are the mirror images (nx - ix) of the
  // rows of butterfly L - j: thread j's rows m >= R/2 need exactly the sigmas thread L - j computes for its rows R - 1 - m < R/2.
  // ColFFT::pass_first deals the butterflies to the lanes so that the two sit in the same wave 32 lanes apart (share_row), each
  // computes the sigma of its first R/2 rows only and the halves change places through the wave's cross-lane network (ds_bpermute,
  // no LDS space, no barrier): R sigma lookups (11 vector instructions and one 16-byte LDS read each) become R/2 + R/2 exchanges.
  // The values are bit for bit those of the unshared kernel: kx enters through its square.  Butterfly 0 (rows m L, mirror R - m, row
  // R/2 L its own mirror) and butterfly L/2 (its own mirror image) take no partner: they source from themselves.  XS = 2, odd phase
  // (rows of the modes 2 r + 1): the mirror of row r is N1 - 1 - r, i.e. butterfly L - 1 - j, and no butterfly is its own partner.
  // (measured on MI355X, profiles/r05_ab/r05_c_*: the whole-column kernels gain -- x pass of 1024^3 1.186 -> 1.15 ms, 1131 instead of 1198
  // vector instructions per wave -- the two-phase Col2 form, whose register budget is full with the parked half, loses 4 %: 10.87 -> 11.29 ms
  // per 2048^3; so XS = 1 only)
  static constexpr bool SIGMA_SHARE = SRC == 0 && SLAB == 0 && XS == 1;
  // butterfly (row base) of slot jl = tid / LPR when there are S slots per wave: the first S/2 slots of a wave take q = (S/2) w + s, the
  // others its partner
  RF_HD int share_row(int jl, int L, int S) const {
    const int h = S >> 1, w = jl / S, sl = jl & (S - 1), q = h * w + (sl & (h - 1));
    if (!(sl & h)) return q;
    if (XS == 2 && xp == 1) return L - 1 - q;
    return q == 0 ? (L >> 1) : L - q;
  }
  RF_HD bool share_self(int j, int L) const { return !(XS == 2 && xp == 1) && (j == 0 || j == (L >> 1)); }
  // all R rows j + m L of the lane's cell pair; `lane` = the lane's index in its wave
  template <int R> RF_HD void load_rows(long long C0, int cl, int j, int L, int lane, V16<float>* out) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const long long C = C0 + cl;
    const uint64_t seed = gp.seed;
    const int iy = (int)((unsigned)C >> nzl_shift()), kz = kz0 + (int)((unsigned)C & (unsigned)(nzl - 1));
    const uint64_t half_plane = ((uint64_t)gp.ny * (uint64_t)(gp.nz / 2)) >> 1;
    const int rbt = XS * j;
    const uint64_t ctr_l = (uint64_t)rbt * half_plane + (((uint64_t)iy * (uint64_t)(gp.nz / 2) + (uint64_t)kz) >> 1);
    const float ky = (float)fast_signed_index(iy, gp.ny) * gp.dky, ky2 = ky * ky;
    float sa[R], sb[R];
#pragma unroll
    for (int m = 0; m < R / 2; ++m) {                      // rows below nx / 2: the mode index is the row's own
      const float kx = (float)(rbt + XS * m * L + (XS == 2 ? xp : 0)) * gp.dkx;
      const float kxy = fmaf(kx, kx, ky2);
      sa[m] = fast_sigma(gp, rec, fast_k2(gp, kxy, kz));
      sb[m] = fast_sigma(gp, rec, fast_k2(gp, kxy, kz + 1));
    }
    const int src = (share_self(j, L) ? lane : lane ^ 32) << 2;
#pragma unroll
    for (int m = 0; m < R / 2; ++m) {
      sa[R - 1 - m] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sa[m])));
      sb[R - 1 - m] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sb[m])));
    }
    if (j == 0 && !(XS == 2 && xp == 1)) {                 // rows m L: the mirror of row m is row R - m, row R/2 (mode nx / 2) is its own
      const float kx = (float)(XS * (R / 2) * L) * gp.dkx;
      const float kxy = fmaf(kx, kx, ky2);
      sa[R / 2] = fast_sigma(gp, rec, fast_k2(gp, kxy, kz));
      sb[R / 2] = fast_sigma(gp, rec, fast_k2(gp, kxy, kz + 1));
#pragma unroll
      for (int m = R / 2 + 1; m < R; ++m) { sa[m] = sa[R - m]; sb[m] = sb[R - m]; }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int ro = m * L, rot = XS * ro + (XS == 2 ? xp : 0);
      const uint64_t ctr_u = pin_uniform((uint64_t)rot * half_plane);
      const PhiloxOut o = philox_native(ctr_l + ctr_u, 0, seed);
      float g0, g1;
      BoxMuller<float>::run_scaled(o.w[0], o.w[1], sa[m], g0, g1);
      out[m].c[0] = mk<float>(g0, g1);
      BoxMuller<float>::run_scaled(o.w[2], o.w[3], sb[m], g0, g1);
      out[m].c[1] = mk<float>(g0, g1);
      if (POT != 0) {
        const int ro_s = XS * ro >= (gp.nx >> 1) ? rot - gp.nx : rot;
        const float kx = (float)(rbt + ro_s) * gp.dkx, kxy = fmaf(kx, kx, ky2);
        const float ra = fast_rcp(fast_k2(gp, kxy, kz)), rb2 = fast_rcp(fast_k2(gp, kxy, kz + 1));
        if (POT == 2) {
          const float ps = (float)gp.pscale;
          out[m].c[0] = mk<float>((out[m].c[0].x * ra) * ps, (out[m].c[0].y * ra) * ps);
          out[m].c[1] = mk<float>((out[m].c[1].x * rb2) * ps, (out[m].c[1].y * rb2) * ps);
        } else {
          const int nzp = gp.ppitch, sl = kz - gp.zoff;
          cplx<float>* row = (pot + (long long)ro * gp.ny * nzp) + (uint32_t)((j * gp.ny + iy) * nzp);
          V16<float> q;
          q.c[0] = mk<float>(out[m].c[0].x * ra, out[m].c[0].y * ra);
          q.c[1] = mk<float>(out[m].c[1].x * rb2, out[m].c[1].y * rb2);
          if (FIX != 0 && kz == 0) row[sl + 1] = q.c[1];
          else v16_store<float>(row + sl, q);
        }
      }
    }
#else
    for (int m = 0; m < R; ++m) out[m] = load(C0, cl, j, m * L);      // (the emulator has no lanes: the same values row by row)
#endif
  }
  // SRC = 2: the memory half of load() -- the row's table entry, then its two deviate pairs -- for the kernel to issue before it
  // stages any table (col_kernel): three dependent round trips (records, row table, pairs) become two that overlap the staging
  static constexpr bool HAS_PRELOAD = (SRC == 2);
  RF_HD V16<float> preload(long long C0, int cl, int rb, int ro) const {
    V16<float> v;
    const long long C = C0 + cl;
    const int iy = (int)((unsigned)C >> nzl_shift()), kz = kz0 + (int)((unsigned)C & (unsigned)(nzl - 1));
    const RowLoc e = load_rowloc((gp.rowtab + (XS * ro + (XS == 2 ? xp : 0))) + (uint32_t)(iy * gp.nx + XS * rb));
    v.c[0] = load_pair_global(row_pair(gp, e, kz));
    v.c[1] = load_pair_global(row_pair(gp, e, kz + 1));
    return v;
  }
  RF_HD V16<float> load_pre(long long C0, int cl, int rb, int ro, const V16<float>& raw) const { return load_impl(C0, cl, rb, ro, &raw); }
  RF_HD V16<float> load(long long C0, int cl, int rb, int ro) const { return load_impl(C0, cl, rb, ro, nullptr); }
  // the lane that owns slot kz = 0 replaces its provisional first cell of every row by the packed,
  // symmetrised (kz=0, kz=nz/2) pair (cold path: one lane in four of one tile in nz/16)
  static constexpr bool ROLLED_LOAD = false;
  RF_HD long long remap_tile(long long t) const { return t; }
  static constexpr bool HAS_FINISH = false;
  static constexpr int FIX_MODE = FIX;
  template <int F2> using with_fix = FastGenColIOT<F2, SLAB, POT, SRC, XS>;
  using fill_io = FastGenColIOT<1, SLAB, POT, SRC, 1>;      // the IO whose fix_value() fix_fill_kernel evaluates (mode index = row)
  const cplx<float>* fixbuf = nullptr;                          // FIX = 3: [ny][nx] repaired slots kz = 0, left by fix_fill_kernel
  RF_HD bool needs_fix(long long C) const { return FIX != 0 && kz0 + (int)((unsigned)C & (unsigned)(nzl - 1)) == 0; }
  // FIX = 3: the repaired slot of mode (XS (rb + ro) + xp, iy) from the side buffer (lane part + uniform part, as load())
  RF_HD cplx<float> fix_load(long long C, int rb, int ro) const {
    const int iy = (int)((unsigned)C >> nzl_shift());
    return load_pair_global((fixbuf + (XS * ro + (XS == 2 ? xp : 0))) + (uint32_t)(iy * gp.nx + XS * rb));
  }
  RF_HD cplx<float> fix_value(long long C, int rb, int ro) const {
    if (FIX == 3) return fix_load(C, rb, ro);
    const uint64_t seed = gp.seed;                                                      // bind_seed() ran first
    const int iy = (int)((unsigned)C >> nzl_shift());
    cplx<float> p0, pn;
    const int ixm = XS * (rb + ro) + (XS == 2 ? xp : 0);                                  // the row's mode index
    const cplx<float> packed = SRC != 0 ? fast_fix_kz0_noise<SRC == 0 ? 1 : SRC>(gp, rec, ixm, iy, p0, pn)
                                        : fast_fix_kz0(gp, rec, seed, ixm, iy, p0, pn);
    if (POT == 1) {
      cplx<float>* row = pot + ((long long)(rb + ro) * gp.ny + iy) * gp.ppitch;   // only the rank with kz0 = 0 gets here: slot 0 = plane 0
      row[0] = p0;
      row[gp.zpitch - 1] = pn;
    }
    if (POT == 2) {           // (plane kz = 0) + i (plane kz = nz/2) of the scaled potential
      const float ps = (float)gp.pscale;
      return mk<float>(p0.x * ps - pn.y * ps, p0.y * ps + pn.x * ps);
    }
    return packed;
  }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<float>& v) const {
    // x0, x1 are multiples of the last pass's row stride L (the launcher checks it) and rb < L: the test is uniform
    if (SLAB && (ro < x0 || ro >= x1)) return;
    v16_store<float>(g.at<false>(base, C0, cl, rb, ro), v);
  }
};
using FastGenColIO = FastGenColIOT<1>;   // (emulator)

// The same fast generation for float64 plans (native generator only: parity / reference-noise mode keeps the exact
// float64 chain of GenColIO).  The deviates and sigma are formed in float32 -- hardware log / sin / cos, LDS
// records -- and widened; the transform itself is float64.  One complex128 per lane (CPL = 1).
template <int FIX = 1, int SLAB = 0, int POT = 0, int XS = 1>
struct FastGenColIO64 {
  static_assert(XS == 1 || (XS == 2 && SLAB == 0 && POT != 1), "half-transform rows: no x-slab restriction, no potential store");
  int xp = 0;
  RF_HD void set_phase(int p) { xp = p; }
  cplx<double>* base;
  ColGeom g;
  FastGenParams gp;
  cplx<double>* pot = nullptr;   // POT = 1: where delta(k) / k^2 goes (API-layout rows of gp.zpitch cells, 16-byte aligned)
  int kz0, nzl;
  int x0 = 0, x1 = 1 << 30;  // replicated-generation mode: see FastGenColIOT
  RF_HD int nzl_shift() const { return 31 - __builtin_clz((unsigned)nzl); }
  const FastRec* rec;
  static constexpr int LDS_EXTRA = FAST_LDS_BINS * (int)sizeof(FastRec);
  RF_HD static void sched_fence(int = 0) {}
  RF_HD void prologue(int tid, int nthreads, void* lds_extra) {
    FastRec* l = reinterpret_cast<FastRec*>(lds_extra);
    for (int i = tid; i < gp.nbins && i < FAST_LDS_BINS; i += nthreads) l[i] = gp.rec[i];
    rec = l;
  }
  RF_HD void bind_seed() { if (gp.seed_dev) { gp.seed = pin_uniform(*gp.seed_dev); gp.seed_dev = nullptr; } }
  // the cell of row (rb, ro) in column C: its native noise index (lane part + uniform part) and |k|^2
  RF_HD void cell_of(long long C, int rb, int ro, int& iy, int& kz, uint64_t& ci_l, uint64_t& ci_u, float& k2) const {
    iy = (int)((unsigned)C >> nzl_shift());
    kz = kz0 + (int)((unsigned)C & (unsigned)(nzl - 1));                                // lane, m-invariant
    const uint64_t plane = (uint64_t)gp.ny * (uint64_t)(gp.nz / 2);                    // noise cells per unit of ix
    // mode index ix = XS (rb + ro) + xp = (lane part rbt) + (uniform part rot), as in FastGenColIOT
    const int rbt = XS * rb, rot = XS * ro + (XS == 2 ? xp : 0);
    ci_l = (uint64_t)rbt * plane + (uint64_t)iy * (uint64_t)(gp.nz / 2) + (uint64_t)kz;
    ci_u = pin_uniform((uint64_t)rot * plane);
    const int ro_s = XS * ro >= (gp.nx >> 1) ? rot - gp.nx : rot;
    const float kx = (float)(rbt + ro_s) * gp.dkx, ky = (float)fast_signed_index(iy, gp.ny) * gp.dky;
    k2 = fast_k2(gp, fmaf(kx, kx, ky * ky), kz);
  }
  // the cell's value from its two Philox words (Box-Muller, sigma) + the potential variants
  RF_HD V16<double> cell_from_words(uint32_t wa, uint32_t wb, float k2, int iy, int kz, int rb, int ro) const {
    float g0, g1;
    BoxMuller<float>::run_scaled(wa, wb, fast_sigma(gp, rec, k2), g0, g1);
    const cplx<float> c = mk<float>(g0, g1);
    V16<double> v;
    v.c[0] = mk<double>((double)c.x, (double)c.y);
    if (POT == 2) {
      const float r = fast_rcp(k2);
      v.c[0] = mk<double>((double)(c.x * r) * gp.pscale, (double)(c.y * r) * gp.pscale);
    }
    if (POT == 1 && !(FIX != 0 && kz == 0)) {          // (slot kz = 0: the two Hermitian planes, written by fix_value())
      const float r = fast_rcp(k2);
      V16<double> q;
      q.c[0] = mk<double>((double)(c.x * r), (double)(c.y * r));
      v16_store<double>((pot + (long long)ro * gp.ny * gp.ppitch) + (uint32_t)((rb * gp.ny + iy) * gp.ppitch + (kz - gp.zoff)), q);
    }
    return v;
  }
  RF_HD V16<double> load(long long C0, int cl, int rb, int ro) const {
    int iy, kz;
    uint64_t ci_l, ci_u;
    float k2;
    cell_of(C0 + cl, rb, ro, iy, kz, ci_l, ci_u, k2);
    const uint64_t ci = ci_l + ci_u;
    const PhiloxOut o = philox_native(ci >> 1, 0, gp.seed);                             // bind_seed() ran first
    const bool odd = (ci & 1u) != 0;
    return cell_from_words(odd ? o.w[2] : o.w[0], odd ? o.w[3] : o.w[1], k2, iy, kz, rb, ro);
  }
  // Two rows at once.  A Philox call serves the cell pair (kz even, kz + 1) of one row, and with one complex128 per lane that pair sits
  // in the lane pair (2l, 2l + 1): load() has both lanes run the same call and keep half of it.  Here the even lane runs row A's call
  // and the odd lane row B's; each sends the half its neighbour needs across (one quad-permute DPP move per word) -- one call per lane
  // and two rows instead of two (230 -> 109 v_mad_u64_u32 per thread in the 1024-point kernel).  Same words, same field.
  static constexpr bool HAS_LOAD_PAIR = true;
  RF_HD void load_pair(long long C0, int cl, int rb, int roA, int roB, V16<double>& a, V16<double>& b) const {
#if defined(__HIP_DEVICE_COMPILE__)
    int iy, kz, iy2, kz2;
    uint64_t ci_l, cuA, cuB, ci_l2;
    float k2A, k2B;
    cell_of(C0 + cl, rb, roA, iy, kz, ci_l, cuA, k2A);
    cell_of(C0 + cl, rb, roB, iy2, kz2, ci_l2, cuB, k2B);
    const bool odd = (kz & 1) != 0;                                                      // (the rest of the noise index is even: nz / 2 is)
    const PhiloxOut o = philox_native((ci_l + (odd ? cuB : cuA)) >> 1, 0, gp.seed);
    const uint32_t ra = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd ? o.w[0] : o.w[2]), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    const uint32_t rb2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd ? o.w[1] : o.w[3]), 0xB1, 0xF, 0xF, true);
    a = cell_from_words(odd ? ra : o.w[0], odd ? rb2 : o.w[1], k2A, iy, kz, rb, roA);
    b = cell_from_words(odd ? o.w[2] : ra, odd ? o.w[3] : rb2, k2B, iy, kz, rb, roB);
#else
    a = load(C0, cl, rb, roA);
    b = load(C0, cl, rb, roB);
#endif
  }
  static constexpr bool ROLLED_LOAD = false;
  RF_HD long long remap_tile(long long t) const { return t; }
  static constexpr bool HAS_FINISH = false;
  static constexpr int FIX_MODE = FIX;
  template <int F2> using with_fix = FastGenColIO64<F2, SLAB, POT, XS>;
  using fill_io = FastGenColIO64<1, SLAB, POT, 1>;
  const cplx<double>* fixbuf = nullptr;                         // FIX = 3: see FastGenColIOT
  RF_HD bool needs_fix(long long C) const { return FIX != 0 && kz0 + (int)((unsigned)C & (unsigned)(nzl - 1)) == 0; }
  RF_HD cplx<double> fix_load(long long C, int rb, int ro) const {
    const int iy = (int)((unsigned)C >> nzl_shift());
    return v16_load<double>((fixbuf + (XS * ro + (XS == 2 ? xp : 0))) + (uint32_t)(iy * gp.nx + XS * rb)).c[0];
  }
  RF_HD cplx<double> fix_value(long long C, int rb, int ro) const {
    if (FIX == 3) return fix_load(C, rb, ro);
    const uint64_t seed = gp.seed;                                                      // bind_seed() ran first
    cplx<float> p0, pn;
    const int iy = (int)((unsigned)C >> nzl_shift());
    const cplx<float> c = fast_fix_kz0(gp, rec, seed, XS * (rb + ro) + (XS == 2 ? xp : 0), iy, p0, pn);
    if (POT == 1) {
      cplx<double>* row = pot + ((long long)(rb + ro) * gp.ny + iy) * gp.ppitch;      // only the rank with kz0 = 0 gets here
      row[0] = mk<double>((double)p0.x, (double)p0.y);
      row[gp.zpitch - 1] = mk<double>((double)pn.x, (double)pn.y);
    }
    if (POT == 2) {
      const double ps = gp.pscale;
      return mk<double>((double)p0.x * ps - (double)pn.y * ps, (double)p0.y * ps + (double)pn.x * ps);
    }
    return mk<double>((double)c.x, (double)c.y);
  }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<double>& v) const {
    if (SLAB && (ro < x0 || ro >= x1)) return;          // uniform: see FastGenColIOT::store
    v16_store<double>(g.at<false>(base, C0, cl, rb, ro), v);
  }
};

}  // namespace rf
