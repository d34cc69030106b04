// rf_fft_row.h -- contiguous (z) passes: RowCfg, the row IOs (plain, gathering, lognormal / per-z epilogues), RowC2R, RowR2C, RowC2C (part of rf_fft.h: include that)
#pragma once
#include "rf_fft.h"

namespace rf {

// ---------------------------------------------------------------------------
// Row (z) pass: complex FFT of length M = nz/2 per row + Hermitian (un)tangle
// ---------------------------------------------------------------------------
template <typename T_, int M_, int R1_, int R2_, int R3_, int NRT_, int NT_>
struct RowCfg {
  using T = T_;
  static constexpr int M = M_, R1 = R1_, R2 = R2_, R3 = R3_, NRT = NRT_, NT = NT_;
  static_assert(R1_ * R2_ * R3_ == M_, "radices must multiply to M");
  static constexpr int NPASS = (R2 == 1 ? 1 : (R3 == 1 ? 2 : 3));
  static constexpr int RL = (NPASS == 1 ? R1 : (NPASS == 2 ? R2 : R3));
  static constexpr int RS = M + ((M - 1) >> 3) + 1 + 1;       // LDS row stride (complex): pad16() of the last element + 2
  static constexpr int TILE_BYTES = (NPASS == 1 ? 0 : NRT * RS * (int)sizeof(cplx<T>));
  static constexpr int TW_BYTES = 2 * M * (int)sizeof(cplx<T>);    // twiddle table exp(2 pi i q / 2M), staged behind the tile
  static constexpr int LDS_BYTES = TILE_BYTES + TW_BYTES;
  static constexpr int L1 = M / R1;                           // butterflies per row in pass 1
  static constexpr int TPR1 = cmax(1, L1 / 2);                // threads per row in pass 1 (each owns a mirror pair)
  static constexpr int IT1 = ceil_div(NRT * TPR1, NT);
  static constexpr int IT2 = (NPASS == 3 ? ceil_div(NRT * (M / R2), NT) : 1);
  static constexpr int ITL = ceil_div(NRT * (M / RL), NT);
};

// c2r row IO over the device array viewed as complex [nrows][M] on input and
// real [nrows][2M] on output (same memory).  Accumulates sum / sum of squares.
// streaming (non-temporal) access to one complex element: the z pass touches every byte exactly once, so there is
// nothing to keep in the caches (a read+write sweep with the hint ran 6 % faster than without, tools/xbench.hip)
template <typename T> RF_HD cplx<T> stream_load(const cplx<T>* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef T vt __attribute__((ext_vector_type(2)));
  const vt v = __builtin_nontemporal_load(reinterpret_cast<const vt*>(p));
  return mk<T>(v.x, v.y);
#else
  return *p;
#endif
}
template <typename T> RF_HD void stream_store(cplx<T>* p, cplx<T> z) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef T vt __attribute__((ext_vector_type(2)));
  vt v; v.x = z.x; v.y = z.y;
  __builtin_nontemporal_store(v, reinterpret_cast<vt*>(p));
#else
  *p = z;
#endif
}

// Per-thread (sum, sum of squares) of the values a thread stores in the z pass.  float32 fields: FOUR float32 accumulators (the real and
// the imaginary slot of a complex store each have their own pair: 16 values per accumulator at nz = 1024), widened once at the end --
// the float64 form cost 8 float64-rate instructions per stored complex, a fifth of the pass's vector work, and the pass is not purely
// HBM-bound (it gained 10 % from cheaper arithmetic alone).  Rounding: 16 fused adds of like-signed squares per accumulator (<= 1e-6
// relative, unbiased), then 10^7 such partial sums added in float64: the field's rms to ~1e-9.  float64 fields accumulate in float64.
template <typename T> struct MomAcc;
template <> struct MomAcc<float> {
  float a1 = 0, b1 = 0, a2 = 0, b2 = 0;
  RF_HD void add(cplx<float> z) { a1 += z.x; b1 += z.y; a2 = fmaf(z.x, z.x, a2); b2 = fmaf(z.y, z.y, b2); }
  RF_HD double sum() const { return (double)a1 + (double)b1; }
  RF_HD double sumsq() const { return (double)a2 + (double)b2; }
};
template <> struct MomAcc<double> {
  double s1 = 0, s2 = 0;
  RF_HD void add(cplx<double> z) { s1 += z.x + z.y; s2 += z.x * z.x + z.y * z.y; }
  RF_HD double sum() const { return s1; }
  RF_HD double sumsq() const { return s2; }
};

template <typename T> struct PlainRowIO {
  cplx<T>* base;
  T scale;                       // 1 / (nx ny nz)
  int M_of;                      // complex elements per row (nz / 2)
  // (tile, row of the tile, lane's element, uniform element offset): the split lets a gathering IO keep its address arithmetic
  // on the scalar unit; here it is just row = tile * NRT + rl, element = kb + ko
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const { return load(tile * NRT + rl, kb + ko); }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const { store(tile * NRT + rl, nb + no, z, mom); }
  RF_HD int gather_seg_shift() const { return -1; }
  RF_HD cplx<T> load(long long row, int k) const { return stream_load(base + row * (long long)M_of + k); }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    z.x *= scale; z.y *= scale;
    stream_store(base + row * (long long)M_of + n, z);
    mom.add(z);
  }
};

// z pass of a slab-decomposed (multi-GPU) plan: this rank owns nxl x-planes.  After the all-to-all
// the receive buffer holds P blocks [src rank g][nxl][ny][nzl]; row (x, y) is gathered from its P
// segments of nzl = nz/(2P) complex (1 KiB each at 2048^3 / 8 GPUs) -- no separate local transpose.
// Output goes to a different buffer (the send buffer, free by then): dense real [nxl][ny][nz].
template <typename T> struct GatherRowIO {
  const cplx<T>* src;
  cplx<T>* dst;
  T scale;
  int M_of;                      // nz / 2
  int nzl;                       // kz planes per source rank
  long long seg_stride;          // complex elements between two source blocks = nxl * ny * nzl
  // (tile, row of the tile, lane's element, uniform element offset): the source block and the tile's row base are workgroup
  // uniform (scalar unit); nzl is a power of two (shift / mask instead of a division per element); streaming accesses like the
  // plain z pass (every byte is touched once)
  RF_HD int nzl_shift() const { return 31 - __builtin_clz((unsigned)nzl); }
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const {
    const int sh = nzl_shift(), mask = nzl - 1;
    const cplx<T>* ub = src + (long long)(ko >> sh) * seg_stride + tile * (long long)(NRT * nzl);
    const int kl = kb + (ko & mask);                       // (< nzl whenever nzl >= the pass's L: the block index is uniform then)
    return stream_load(ub + ((long long)(kl >> sh) * seg_stride + (long long)(rl * nzl + (kl & mask))));
  }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const {
    cplx<T>* ub = dst + tile * (long long)(NRT * M_of) + no;
    z.x *= scale; z.y *= scale;
    stream_store(reinterpret_cast<cplx<T>*>((size_t)ub + (size_t)((uint32_t)(rl * M_of + nb) * (uint32_t)sizeof(cplx<T>))), z);
    mom.add(z);
  }
  RF_HD int gather_seg_shift() const { return -1; }
  RF_HD cplx<T> load(long long row, int k) const {
    const int g = k >> nzl_shift(), kk = k & (nzl - 1);
    return stream_load(src + ((long long)g * seg_stride + row * (long long)nzl + kk));
  }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    z.x *= scale; z.y *= scale;
    stream_store(dst + (row * (long long)M_of + n), z);
    mom.add(z);
  }
};

// z pass reading the blocked intermediate X [xb][kt][iy][rb][tc] (xblock_*_geom) and writing the dense rows of W.  The NRT rows
// of a workgroup are consecutive ix of one (xb, iy): local row index (of the slab the launch covers) = (xb * ny + iy) * rb + r,
// so tile T covers rows T * NRT .. + NRT of ONE (xb, iy) (rb is a multiple of NRT) and every kz tile of theirs is one contiguous
// chunk of NRT * tc cells.  SEG_SHIFT = log2(tc): pass 1 deals its threads so that a wave reads whole chunks (RowC2R::pass_first).
template <typename T> struct XGatherRowIO {
  const cplx<T>* src;            // X, at the first x block of the slab
  cplx<T>* dst;                  // W, at the first x plane of the slab
  T scale;
  int M_of;                      // nz / 2
  int seg_shift;                 // log2(tc)
  int rb_shift, ny_shift;        // log2 of the rows of x per block and of ny (both powers of two on this path)
  long long kt_stride;           // cells between two kz tiles of a block = ny * rb * tc
  long long xb_stride;           // cells between two x blocks = (M / tc) * kt_stride
  RF_HD int gather_seg_shift() const { return seg_shift; }
  // A tile's NRT rows share (xb, iy) and are consecutive r (NRT divides rb): everything but the row-in-tile, the lane's element
  // and the kz tile of the uniform offset is workgroup-uniform (scalar unit); the lane part fits 32 bits (one x block).
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const {
    const long long t0 = tile * NRT, q = t0 >> rb_shift, r0 = t0 & ((1LL << rb_shift) - 1);
    const long long xb = q >> ny_shift, iy = q & ((1LL << ny_shift) - 1);
    const int mask = (1 << seg_shift) - 1;
    const cplx<T>* ub = src + xb * xb_stride + ((((iy << rb_shift) + r0)) << seg_shift) + (long long)(ko >> seg_shift) * kt_stride;
    const int kl = kb + (ko & mask);                  // (ko is a multiple of the segment length in the product: kl == kb)
    const uint32_t lane = ((uint32_t)rl << seg_shift) + (uint32_t)(kl >> seg_shift) * (uint32_t)kt_stride + (uint32_t)(kl & mask);
    return stream_load(reinterpret_cast<const cplx<T>*>((size_t)ub + (size_t)(lane * (uint32_t)sizeof(cplx<T>))));
  }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const {
    const long long t0 = tile * NRT, q = t0 >> rb_shift, r0 = t0 & ((1LL << rb_shift) - 1);
    const long long xb = q >> ny_shift, iy = q & ((1LL << ny_shift) - 1);
    cplx<T>* ub = dst + (((((xb << rb_shift) + r0) << ny_shift) + iy)) * (long long)M_of + no;
    const uint32_t lane = (uint32_t)rl * ((uint32_t)M_of << ny_shift) + (uint32_t)nb;
    z.x *= scale; z.y *= scale;
    stream_store(reinterpret_cast<cplx<T>*>((size_t)ub + (size_t)(lane * (uint32_t)sizeof(cplx<T>))), z);
    mom.add(z);
  }
  RF_HD cplx<T> load(long long row, int k) const {
    const long long r = row & ((1LL << rb_shift) - 1), q = row >> rb_shift;      // q = xb * ny + iy
    const long long xb = q >> ny_shift, iy = q & ((1LL << ny_shift) - 1);
    return stream_load(src + xb * xb_stride + (long long)(k >> seg_shift) * kt_stride + ((((iy << rb_shift) + r)) << seg_shift) + (k & ((1 << seg_shift) - 1)));
  }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    const long long r = row & ((1LL << rb_shift) - 1), q = row >> rb_shift;
    const long long xb = q >> ny_shift, iy = q & ((1LL << ny_shift) - 1);
    z.x *= scale; z.y *= scale;
    stream_store(dst + (((((xb << rb_shift) + r) << ny_shift) + iy)) * (long long)M_of + n, z);
    mom.add(z);
  }
};

// z pass with the lognormal map in its epilogue: rho = exp(delta * Ap_z) * Bp_z with the float64 tables Ap = sqrt(log t) / sigma,
// Bp = density / sqrt(t), t = 1 + (sigma growth_z)^2, formed on the device from the y pass's Parseval sum (AccColIO,
// lognormal_tables_kernel).  The reference does the same map as four in-place numpy statements with a rounding to the array
// dtype after each (cosmotools.py:216-220, then generate.py:273); here the two divisions are folded into the tables (a float64
// division costs ~15 instructions per element and the pass has 2 x 10^9 of them): the result is within a few ulp of the
// argument of exp of the reference's chain (<= 1e-15 relative for float64 fields, 3e-7 for float32 ones; rf_lognormal is the
// rounding-exact, unfused form).  Element n of a row holds the reals z = 2n, 2n + 1; the tables are 16 KB, L1-resident.
RF_HD float exp_t(float x) { return expf(x); }
RF_HD double exp_t(double x) { return exp(x); }
// float64 plans: exp(t ln2 / 64) for an argument already in units of ln2 / 64 (the table Ap carries the factor 64 / ln2, lognormal_ap_unit):
// t = k + f, |f| <= 1/2, k = 64 e + j: 2^e * 2^(j/64) * exp(f ln2/64), the middle factor from a 64-entry table in LDS (rf_exp2_tab.h,
// correctly rounded), the last a degree-4 polynomial (|r| <= 0.0055: the first dropped term is 4e-14; round 4: degree 5).  11 float64-rate instructions
// and one ds_read_b64 per element where the library's exp takes ~22 (no table: a degree-11 polynomial, range checks); the z pass of a
// float64 plan issues 16 of them per thread.  |error| <= 4e-14 relative (the dropped term) + 1 ulp of the result + the rounding of t
// (ulp(t) ln2 / 128 <= 6e-16 at |x| = 5.5, the same size as the rounding of the product delta * Ap that both forms share).  Out-of-range arguments saturate through
// the conversion and ldexp (inf / 0), NaN propagates through r.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const double rf_exp2_tab[64] = {RF_EXP2_TAB_VALUES};
#else
static const double rf_exp2_tab[64] = {RF_EXP2_TAB_VALUES};
#endif
// the factor the float64 z pass expects in Ap: 64 / ln 2 (the unit of exp_scaled64) times the transform's 1 / (nx ny nz)
template <typename T> RF_HD double lognormal_ap_unit(double scale) { return sizeof(T) == 8 ? 0x1.71547652b82fep+6 /* 64 / ln 2 */ * scale : 1.0; }
RF_HD int exp_k_of(double kf) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)kf;                                         // (v_cvt_i32_f64 saturates)
#else
  return kf > 1e9 ? 1000000000 : (kf < -1e9 ? -1000000000 : (kf == kf ? (int)kf : 0));
#endif
}
// 2^(k >> 6) * tj * exp(c f), c = ln 2 / 64, tj = 2^((k & 63) / 64)
RF_HD double exp_finish(double f, double tj, int k) {
  // exp(c f) - 1 = f (c + f (c^2/2 + f (c^3/6 + f c^4/24))): |c f| <= ln2/128, so the first dropped term (c f)^5/120 is <= 3.9e-14 of
  // the result -- the fused map is checked against the reference's chain to 1e-12, its tolerance is 1e-11 (rounds 3 - 4 carried the
  // fifth-order term too: one more float64 fma per element, 2 x 10^9 of them per 1024^3 field)
  double q = 0x1.3b2ab6fba4e77p-31 /* c^4/24 */;
  q = __builtin_fma(f, q, 0x1.c6b08d704a0c0p-23 /* c^3/6 */);
  q = __builtin_fma(f, q, 0x1.ebfbdff82c58fp-15 /* c^2/2 */);
  q = __builtin_fma(f, q, 0x1.62e42fefa39efp-7 /* c */);
  return __builtin_ldexp(__builtin_fma(tj, f * q, tj), k >> 6);
}
template <int STRIDE = 1> RF_HD double exp_scaled64(double t, const double* tab) {
  const double kf = __builtin_rint(t);
  const double f = t - kf;                                // exact
  const int k = exp_k_of(kf);
  const double tj = tab ? tab[(k & 63) * STRIDE] : 1.0;    // (tab is never null in the product)
  return exp_finish(f, tj, k);
}
// Where the table lives: the LDS row image skips every ninth complex (pad16), so row 0 of a tile of rows of M >= 512 complex128 has 64
// unused 16-byte slots at 9 j + 8 -- entry j goes there (stage(), 64 threads, in front of the kernel's first barrier; nothing else
// ever touches those slots).  Measured at 1024^3 float64 on MI355X (z pass, plain 3.12 ms): the library's exp 3.65, the table read
// from global memory 3.62 (a 64-lane gather per element), from 512 more bytes of LDS 4.48 (the pass fills a third of the CU's LDS
// to within one allocation unit: two workgroups per CU instead of three), from the pad slots: see DESIGN.md section 3.9.
template <typename T, int SPARE = 0> struct LognormalRowIO {
  static_assert(SPARE == 0 || sizeof(T) == 8, "the exp table is the float64 plans'");
  cplx<T>* base;
  T scale;                       // 1 / (nx ny nz)
  int M_of;
  const double* Ap;              // [2 M] sqrt(log t_z) / sigma  (float64 plans: times lognormal_ap_unit)
  const double* Bp;              // [2 M] density_z / sqrt(t_z)
  const double* etab = nullptr;  // SPARE: rf_exp2_tab in the pad slots of the tile's row 0
  static constexpr bool WANTS_STAGE = SPARE != 0;
  static constexpr int ESTRIDE = SPARE ? 18 : 1;        // doubles between two entries
  template <class C> RF_HD void stage(int tid, void* lds) {
    static_assert(!SPARE || (C::NPASS >= 2 && C::RS >= 9 * 63 + 8 + 1), "64 pad slots in row 0");
    double* l = reinterpret_cast<double*>(lds) + 16;
    if (tid < 64) l[18 * tid] = rf_exp2_tab[tid];
    etab = l;
  }
  RF_HD int gather_seg_shift() const { return -1; }
  RF_HD float map(float d, int z) const {
    d = (float)((double)d * Ap[z]);
    d = exp_t(d);
    return (float)((double)d * Bp[z]);
  }
  RF_HD double map(double d, int z) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const double* tb = SPARE ? etab : rf_exp2_tab;
#else
    const double* tb = (SPARE && etab) ? etab : rf_exp2_tab;        // (the emulator has no staging step)
    if (!(SPARE && etab)) return exp_scaled64<1>(d * Ap[z], tb) * Bp[z];
#endif
    return exp_scaled64<ESTRIDE>(d * Ap[z], tb) * Bp[z];
  }
  RF_HD cplx<T> load(long long row, int k) const { return stream_load(base + row * (long long)M_of + k); }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    if (sizeof(T) == 8) {          // (float64 plans: 1 / (nx ny nz) is part of Ap too)
      z.x = map(z.x, 2 * n);
      z.y = map(z.y, 2 * n + 1);
    } else {
      z.x = map(z.x * scale, 2 * n);
      z.y = map(z.y * scale, 2 * n + 1);
    }
    stream_store(base + row * (long long)M_of + n, z);
    mom.add(z);
  }
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const { return load(tile * NRT + rl, kb + ko); }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const { store(tile * NRT + rl, nb + no, z, mom); }
  // the last pass of a multi-pass row: the 2 R entries of Ap and Bp first (RowC2R::pass_last), then all R outputs in one call --
  // for float64 in phases (arguments and table reads of all 2 R elements, then the polynomials, then the stores)
  static constexpr bool HAS_PRE = true;
  template <int R> struct Pre { double a[2 * R], b[2 * R]; };
  template <int R> RF_HD void prefetch(int j, int L, Pre<R>& p) const {
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int z = 2 * (j + m * L);
      p.a[2 * m] = Ap[z]; p.a[2 * m + 1] = Ap[z + 1];
      p.b[2 * m] = Bp[z]; p.b[2 * m + 1] = Bp[z + 1];
    }
  }
  template <int NRT, int R> RF_HD void store_row(long long tile, int rl, int j, int L, const cplx<T>* v, MomAcc<T>& mom, const Pre<R>& p) const {
    cplx<T>* const out = base + (tile * NRT + rl) * (long long)M_of + j;
    if constexpr (sizeof(T) == 8) {
#if defined(__HIP_DEVICE_COMPILE__)
      const double* tb = SPARE ? etab : rf_exp2_tab;
      constexpr int ES = ESTRIDE;
#else
      const double* tb = (SPARE && etab) ? etab : rf_exp2_tab;
      const int ES = (SPARE && etab) ? ESTRIDE : 1;
#endif
      double f[2 * R], tj[2 * R];
      int k[2 * R];
#pragma unroll
      for (int i = 0; i < 2 * R; ++i) {
        const double t = ((i & 1) ? v[i / 2].y : v[i / 2].x) * p.a[i];
        const double kf = __builtin_rint(t);
        f[i] = t - kf;
        k[i] = exp_k_of(kf);
        tj[i] = tb[(k[i] & 63) * ES];
      }
#pragma unroll
      for (int m = 0; m < R; ++m) {
        cplx<T> z;
        z.x = (T)(exp_finish(f[2 * m], tj[2 * m], k[2 * m]) * p.b[2 * m]);
        z.y = (T)(exp_finish(f[2 * m + 1], tj[2 * m + 1], k[2 * m + 1]) * p.b[2 * m + 1]);
        stream_store(out + m * L, z);
        mom.add(z);
      }
    } else {
#pragma unroll
      for (int m = 0; m < R; ++m) {
        cplx<T> z;
        z.x = (T)((double)exp_t((T)((double)(v[m].x * scale) * p.a[2 * m])) * p.b[2 * m]);
        z.y = (T)((double)exp_t((T)((double)(v[m].y * scale) * p.a[2 * m + 1])) * p.b[2 * m + 1]);
        stream_store(out + m * L, z);
        mom.add(z);
      }
    }
  }
};
// does a row IO stage something into the tile's spare LDS slots at the start of the kernel?
template <class IO, class = void> struct row_io_wants_stage { static constexpr bool value = false; };
template <class IO> struct row_io_wants_stage<IO, typename std::enable_if<IO::WANTS_STAGE>::type> { static constexpr bool value = true; };

// z pass whose store multiplies plane z by a per-z factor (float64 table, the rounding of rf_scale_z on the stored field): the
// light-cone weighting G(z) / (1 + z) of calculate_newtonian_potential (generate.py:344-347) without a sweep of its own
template <typename T> struct ScaleZRowIO {
  cplx<T>* base;
  T scale;                       // 1 / (nx ny nz)
  int M_of;
  const double* Sz;              // [2 M]
  RF_HD int gather_seg_shift() const { return -1; }
  RF_HD cplx<T> load(long long row, int k) const { return stream_load(base + row * (long long)M_of + k); }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    z.x = (T)((double)(z.x * scale) * Sz[2 * n]);
    z.y = (T)((double)(z.y * scale) * Sz[2 * n + 1]);
    stream_store(base + row * (long long)M_of + n, z);
    mom.add(z);
  }
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const { return load(tile * NRT + rl, kb + ko); }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const { store(tile * NRT + rl, nb + no, z, mom); }
  // (the table entries in front of the last pass, as LognormalRowIO)
  static constexpr bool HAS_PRE = true;
  template <int R> struct Pre { double s[2 * R]; };
  template <int R> RF_HD void prefetch(int j, int L, Pre<R>& p) const {
#pragma unroll
    for (int m = 0; m < R; ++m) { p.s[2 * m] = Sz[2 * (j + m * L)]; p.s[2 * m + 1] = Sz[2 * (j + m * L) + 1]; }
  }
  template <int NRT, int R> RF_HD void store_row(long long tile, int rl, int j, int L, const cplx<T>* v, MomAcc<T>& mom, const Pre<R>& p) const {
    cplx<T>* const out = base + (tile * NRT + rl) * (long long)M_of + j;
#pragma unroll
    for (int m = 0; m < R; ++m) {
      cplx<T> z;
      z.x = (T)((double)(v[m].x * scale) * p.s[2 * m]);
      z.y = (T)((double)(v[m].y * scale) * p.s[2 * m + 1]);
      stream_store(out + m * L, z);
      mom.add(z);
    }
  }
};

// does a row IO fetch table entries ahead of the last pass (IO::Pre<R>, prefetch<R>(), store_row<NRT, R>())?
struct RowNoPre {};
template <class IO, int R, class = void> struct row_io_pre { static constexpr bool value = false; using type = RowNoPre; };
template <class IO, int R> struct row_io_pre<IO, R, typename std::enable_if<IO::HAS_PRE>::type> {
  static constexpr bool value = true;
  using type = typename IO::template Pre<R>;
};

// tw = exp(+2 pi i q / (2M)), q in [0, 2M): t_k = tw[k], w_M^q = tw[2q]
template <class C, class IO>
struct RowC2R {
  using T = typename C::T;
  using cx = cplx<T>;
  static constexpr int M = C::M, NT = C::NT;
  static constexpr int DIR = +1;

  struct Regs { cx v[C::IT2][cmax(C::R2, 1)]; MomAcc<T> mom; };

  RF_HD static cx* lds_at(cx* lds, int rl, int i) { return lds + (long long)rl * C::RS + pad16(i); }
  RF_HD static cx* lds_tw(cx* lds) { return lds + C::TILE_BYTES / (int)sizeof(cx); }
  // The twiddles in LDS: NOT the plain table exp(2 pi i q / 2M) the kernel gets, but three tables cut from it, each in the order its
  // pass reads it, 2M entries in all (round 5).  Read from the plain table, the middle pass's w_(R1 R2)^(c m) sit 16 m entries apart
  // -- 128 m bytes: a 4- or 8-way bank conflict per read -- and the last pass's w_M^(m j) 2 m entries apart (2- to 8-way); on the
  // z pass of 1024^3 float32 40 % of the LDS-array cycles were conflict cycles (SQ_LDS_BANK_CONFLICT 1.08e7 of SQ_LDS_IDX_ACTIVE
  // 2.70e7 per launch, profiles/r05_a_pmc_sq_*), two thirds of them from these reads.
  //   U [k]            = t_k = tw[k], k < M                       : the untangle (consecutive lanes, consecutive k)
  //   LT[(m-1) LL + j] = tw[2 m j],   1 <= m < RL, j < LL = M / RL : the last pass (consecutive lanes, consecutive j)
  //   MT[(m-1) R1 + c] = tw[2 m c M / (R1 R2)], 1 <= m < R2, c < R1 : the middle pass (lane j reads entry c = j mod R1: broadcast)
  static constexpr int LL = M / C::RL;
  static constexpr int TW_LT = M, TW_MT = M + (C::NPASS >= 2 ? (C::RL - 1) * LL : 0);
  static constexpr int TW_END = TW_MT + (C::NPASS == 3 ? (C::R2 - 1) * C::R1 : 0);
  static_assert(TW_END <= 2 * M, "the three tables fit the space of the plain one");
  RF_HD static int tw_source(int e) {                  // entry e of the LDS image <- entry tw_source(e) of the plain table
    if (e < TW_LT) return e;
    if (e < TW_MT) { const int r = e - TW_LT; return 2 * (r / LL + 1) * (r % LL); }
    if (e < TW_END) { const int r = e - TW_MT; return 2 * (r / C::R1 + 1) * (r % C::R1) * (M / (C::R1 * cmax(C::R2, 1))); }
    return 0;
  }
  RF_HD static cx tw_last(const cx* ltw, int m, int j) { return ltw[TW_LT + (m - 1) * LL + j]; }
  RF_HD static cx tw_mid(const cx* ltw, int m, int j) {
    return ltw[TW_MT + (m - 1) * C::R1 + (j % C::R1)];
  }
  // The table (needed by pass 1 already: the untangle) goes global -> registers (tw_fetch), then the row data
  // (pass_first_load), then registers -> LDS (tw_stage) and a barrier: both trips to memory are in flight together, and the
  // older one -- the small table -- is the one that is waited for first (loads retire in order).
  static constexpr int TWPT = ceil_div(2 * M, NT);
  struct TwRegs { cx v[TWPT]; };
  RF_HD static void tw_fetch(int tid, const cx* tw, TwRegs& t) {
#pragma unroll
    for (int k = 0; k < TWPT; ++k) t.v[k] = tw[tw_source((tid + k * NT) & (2 * M - 1))];       // (M is a power of two: no branch, no undefined slot)
  }
  RF_HD static void tw_stage(int tid, cx* lds, const TwRegs& t) {
    cx* l = lds_tw(lds);
#pragma unroll
    for (int k = 0; k < TWPT; ++k)
      if (tid + k * NT < 2 * M) l[tid + k * NT] = t.v[k];
  }
  // prologue (emulator; the kernel calls the pieces): stage the twiddle table in LDS (a barrier follows)
  RF_HD static void prologue(int tid, const cx* tw, cx* lds) {
    TwRegs t;
    tw_fetch(tid, tw, t);
    tw_stage(tid, lds, t);
  }

  // pass 1 outputs: LDS (NPASS > 1) or global (NPASS == 1)
  template <int R>
  RF_HD static void emit(int rl, long long row, int idx, const cx* v, const IO& io, cx* lds, Regs& r) {
#pragma unroll
    for (int m = 0; m < R; ++m) {
      if (C::NPASS == 1) io.template store2<C::NRT>(row / C::NRT, rl, idx, m, v[m], r.mom);
      else if (R % 8 == 0) lds_at(lds, rl, idx)[m + (m >> 3)] = v[m];      // (idx is a multiple of R: the padding of idx + m splits, one base + immediates)
      else *lds_at(lds, rl, idx + m) = v[m];
    }
  }

  // which (row of the tile, butterfly pair) thread `w` of pass 1 owns
  RF_HD static void first_owner(int w, const IO& io, int& rl, int& q) {
    rl = w / C::TPR1;
    q = w % C::TPR1;
    // gathering IO: 2^sg consecutive k of a row are one segment of the source and the segments of the tile's NRT rows are
    // adjacent, so thread w takes k-in-segment = w % 2^sg, row = (w >> sg) % NRT, segment = w / (NRT 2^sg): a wave's loads
    // then cover whole chunks of NRT segments instead of one segment in each of many blocks
    const int sg = io.gather_seg_shift();
    if (sg >= 0 && C::TPR1 % (1 << sg) == 0) {
      rl = (w >> sg) % C::NRT;
      q = ((w >> sg) / C::NRT << sg) + (w & ((1 << sg) - 1));
      if (q >= C::TPR1) rl = C::NRT;         // (threads beyond NRT * TPR1: idle, as in the plain mapping)
    }
  }
  struct In { cx A[C::IT1][C::R1], B[C::IT1][C::R1]; };
  // pass 1, first half: the mirror pair's inputs global -> registers
  RF_HD static void pass_first_load(int tid, long long tile, long long nrows, const IO& io, In& in) {
    constexpr int R = C::R1, L = C::L1;
#pragma unroll
    for (int it = 0; it < C::IT1; ++it) {
      int rl, q;
      first_owner(it * NT + tid, io, rl, q);
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        const int ja = q, jb = (q == 0) ? L / 2 : L - q;
#pragma unroll
        for (int m = 0; m < R; ++m) in.A[it][m] = io.template load2<C::NRT>(tile, rl, ja, m * L);
        if (L >= 2) {
#pragma unroll
          for (int m = 0; m < R; ++m) in.B[it][m] = io.template load2<C::NRT>(tile, rl, jb, m * L);
        }
      }
    }
  }
  // pass 1, second half: untangle -> R1 butterflies of the mirror pair -> LDS
  RF_HD static void pass_first_compute(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds, Regs& r, const In& in) {
    constexpr int R = C::R1, L = C::L1;
    r.mom = MomAcc<T>();
#pragma unroll
    for (int it = 0; it < C::IT1; ++it) {
      int rl, q;
      first_owner(it * NT + tid, io, rl, q);
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        const bool self = (q == 0);
        const int ja = q;
        const int jb = self ? L / 2 : L - q;
        const bool has_b = (L >= 2);
        cx A[R], B[R], ZA[R], ZB[R];
#pragma unroll
        for (int m = 0; m < R; ++m) { A[m] = in.A[it][m]; B[m] = in.B[it][m]; }
#pragma unroll
        for (int m = 0; m < R; ++m) {
          const cx ta = tw[ja + m * L];
          if (self) {
            if (m == 0) ZA[0] = mk<T>(A[0].x + A[0].y, A[0].x - A[0].y);   // (DC + Nyq) + i (DC - Nyq)
            else ZA[m] = c2r_untangle(A[m], A[R - m], ta);
          } else {
            ZA[m] = c2r_untangle(A[m], B[R - 1 - m], ta);
          }
          if (has_b) {
            const cx tb = tw[jb + m * L];
            ZB[m] = self ? c2r_untangle(B[m], B[R - 1 - m], tb) : c2r_untangle(B[m], A[R - 1 - m], tb);
          }
        }
        DFT<R, DIR>::run(ZA);
        emit<R>(rl, row, ja * R, ZA, io, lds, r);
        if (has_b) {
          DFT<R, DIR>::run(ZB);
          emit<R>(rl, row, jb * R, ZB, io, lds, r);
        }
      }
    }
  }
  // pass 1: global -> untangle -> R1 butterflies of the mirror pair -> LDS
  RF_HD static void pass_first(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds, Regs& r) {
    In in;
    pass_first_load(tid, tile, nrows, io, in);
    pass_first_compute(tid, tile, nrows, io, tw, lds, r, in);
  }

  RF_HD static void pass_mid_read(int tid, const cx* tw, cx* lds, Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
        const cx* const rd = lds_at(lds, rl, j);                              // L % 8 == 0: pad16(j + m L) = pad16(j) + m (L + L / 8)
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = (L % 8 == 0) ? rd[m * (L + L / 8)] : *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, tw_mid(tw, m, j));
          r.v[it][m] = x;
        }
        DFT<R, DIR>::run(r.v[it]);
      }
    }
  }
  RF_HD static void pass_mid_write(int tid, cx* lds, const Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
        const int ob = stockham_out_base(j, Ns, R);
        cx* const wr = lds_at(lds, rl, ob);                                   // Ns % 8 == 0: pad16(ob + m Ns) = pad16(ob) + m (Ns + Ns / 8)
#pragma unroll
        for (int m = 0; m < R; ++m) {
          if (Ns % 8 == 0) wr[m * (Ns + Ns / 8)] = r.v[it][m];
          else *lds_at(lds, rl, ob + m * Ns) = r.v[it][m];
        }
      }
    }
  }

  RF_HD static void pass_last(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds, Regs& r) {
    constexpr int R = C::RL, L = M / R;
#pragma unroll
    for (int it = 0; it < C::ITL; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        cx v[R];
        // IOs whose store needs per-z table entries (LognormalRowIO, ScaleZRowIO) fetch the thread's 2 R entries HERE, in front of the
        // LDS reads and the butterfly, and store the R outputs in one call: written per element (load table -> map -> store) the
        // epilogue is R round trips in a row -- on gfx950 a load issued behind a store is waited for through the same counter
        // as the store (vmcnt, in order), so every element waited for the previous element's write to retire
        typename row_io_pre<IO, R>::type pre;
        if constexpr (row_io_pre<IO, R>::value) io.template prefetch<R>(j, L, pre);
        const cx* const rd = lds_at(lds, rl, j);
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = (L % 8 == 0) ? rd[m * (L + L / 8)] : *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, tw_last(tw, m, j));
          v[m] = x;
        }
        DFT<R, DIR>::run(v);
        if constexpr (row_io_pre<IO, R>::value) {
          io.template store_row<C::NRT, R>(tile, rl, j, L, v, r.mom, pre);
        } else {
#pragma unroll
          for (int m = 0; m < R; ++m) io.template store2<C::NRT>(tile, rl, j, m * L, v[m], r.mom);
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// Forward row pass: r2c along z (transform.py:199-206,270 -- np.fft.rfftn's last axis)
// ---------------------------------------------------------------------------
// The real row x[0..nz) is viewed as M = nz/2 complex z[m] = x[2m] + i x[2m+1]; Z = FFT_M(z) (forward);
// X[k] = (Z[k] + conj Z[M-k])/2 - (i/2) conj(t_k) (Z[k] - conj Z[M-k]).  The tangle needs the mirror
// pair (k, M-k) of the FFT *output*, so the LAST pass gives one thread the butterfly pair (j, L - j)
// (the mirror image of RowC2R's first pass).  Output in place: M complex per row, element 0 packs
// (X[0], X[M]) -- both are real.
template <typename T> struct PlainRowFwdIO {
  cplx<T>* base;
  int M_of;
  RF_HD cplx<T> load(long long row, int k) const { return base[row * (long long)M_of + k]; }
  RF_HD void store(long long row, int k, cplx<T> z) const { base[row * (long long)M_of + k] = z; }
};

template <class C, class IO>
struct RowR2C {
  using T = typename C::T;
  using cx = cplx<T>;
  static constexpr int M = C::M, NT = C::NT;
  static constexpr int DIR = -1;
  // the paired LAST pass needs L/2 threads per row (or 1)
  static constexpr int LL = M / C::RL;
  static constexpr int TPRL = cmax(1, LL / 2);
  static constexpr int ITF = ceil_div(C::NRT * (M / C::R1), NT);
  static constexpr int ITLP = ceil_div(C::NRT * TPRL, NT);

  struct Regs { cx v[C::IT2][cmax(C::R2, 1)]; };

  RF_HD static cx* lds_at(cx* lds, int rl, int i) { return lds + (long long)rl * C::RS + pad16(i); }
  RF_HD static cx* lds_tw(cx* lds) { return lds + C::TILE_BYTES / (int)sizeof(cx); }
  RF_HD static void prologue(int tid, const cx* tw, cx* lds) {
    cx* l = lds_tw(lds);
    for (int i = tid; i < 2 * M; i += NT) l[i] = tw[i];
  }

  // pass 1 (only when NPASS >= 2): global -> R1 butterfly -> LDS
  RF_HD static void pass_first(int tid, long long tile, long long nrows, const IO& io, cx* lds) {
    constexpr int R = C::R1, L = M / R;
#pragma unroll
    for (int it = 0; it < ITF; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        cx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = io.load(row, j + m * L);
        DFT<R, DIR>::run(v);
#pragma unroll
        for (int m = 0; m < R; ++m) *lds_at(lds, rl, j * R + m) = v[m];
      }
    }
  }

  RF_HD static void pass_mid_read(int tid, const cx* tw, cx* lds, Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, cconj(tw[2 * stockham_tw_index(j, m, Ns, R, M)]));
          r.v[it][m] = x;
        }
        DFT<R, DIR>::run(r.v[it]);
      }
    }
  }
  RF_HD static void pass_mid_write(int tid, cx* lds, const Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
        const int ob = stockham_out_base(j, Ns, R);
#pragma unroll
        for (int m = 0; m < R; ++m) *lds_at(lds, rl, ob + m * Ns) = r.v[it][m];
      }
    }
  }

  // last pass: (LDS | global when NPASS == 1) -> RL butterflies of the mirror pair -> tangle -> global
  RF_HD static void pass_last(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds) {
    constexpr int R = C::RL, L = LL;
#pragma unroll
    for (int it = 0; it < ITLP; ++it) {
      const int w = it * NT + tid;
      const int rl = w / TPRL, q = w % TPRL;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        const bool self = (q == 0);
        const int ja = q, jb = self ? L / 2 : L - q;
        const bool has_b = (L >= 2);
        cx A[R], B[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = (C::NPASS == 1) ? io.load(row, ja + m * L) : *lds_at(lds, rl, ja + m * L);
          if (C::NPASS > 1 && m > 0) x = cmul(x, cconj(tw[2 * m * ja]));
          A[m] = x;
        }
        DFT<R, DIR>::run(A);                      // A[m] = Z[ja + m L]
        if (has_b) {
#pragma unroll
          for (int m = 0; m < R; ++m) {
            cx x = (C::NPASS == 1) ? io.load(row, jb + m * L) : *lds_at(lds, rl, jb + m * L);
            if (C::NPASS > 1 && m > 0) x = cmul(x, cconj(tw[2 * m * jb]));
            B[m] = x;
          }
          DFT<R, DIR>::run(B);                    // B[m] = Z[jb + m L]
        }
        // mirror of k = ja + m L is M - k = jb + (R-1-m) L  (ja >= 1); for ja = 0: (R - m) L, and k = 0 <-> M
#pragma unroll
        for (int m = 0; m < R; ++m) {
          const int ka = ja + m * L;
          if (self) {
            if (m == 0) io.store(row, 0, mk<T>(A[0].x + A[0].y, A[0].x - A[0].y));   // (X[0], X[M]) packed
            else io.store(row, ka, r2c_tangle(A[m], A[R - m], tw[ka]));
          } else {
            io.store(row, ka, r2c_tangle(A[m], B[R - 1 - m], tw[ka]));
          }
          if (has_b) {
            const int kb = jb + m * L;
            io.store(row, kb, self ? r2c_tangle(B[m], B[R - 1 - m], tw[kb]) : r2c_tangle(B[m], A[R - 1 - m], tw[kb]));
          }
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// Plain complex row pass (unpacked c2c plans, transform.py:207-213,266-270): FFT of length M = nz along
// the contiguous axis, either direction.  tw = exp(+2 pi i q / M), q in [0, M) (conjugated for DIR = -1).
// ---------------------------------------------------------------------------
template <typename T> struct ScaledRowIO {
  cplx<T>* base;
  int M_of;                      // complex elements per row (nz)
  T scale;                       // 1 (forward) or 1 / (nx ny nz) (inverse, numpy normalisation)
  RF_HD cplx<T> load(long long row, int k) const { return base[row * (long long)M_of + k]; }
  RF_HD void store(long long row, int k, cplx<T> z) const {
    z.x *= scale; z.y *= scale;
    base[row * (long long)M_of + k] = z;
  }
};

template <class C, int DIR_, class IO>
struct RowC2C {
  using T = typename C::T;
  using cx = cplx<T>;
  static constexpr int M = C::M, NT = C::NT, DIR = DIR_;
  static constexpr int ITF = ceil_div(C::NRT * (M / C::R1), NT);

  struct Regs { cx v[C::IT2][cmax(C::R2, 1)]; };

  RF_HD static cx* lds_at(cx* lds, int rl, int i) { return lds + (long long)rl * C::RS + pad16(i); }
  RF_HD static cx* lds_tw(cx* lds) { return lds + C::TILE_BYTES / (int)sizeof(cx); }
  RF_HD static void prologue(int tid, const cx* tw, cx* lds) {
    cx* l = lds_tw(lds);
    for (int i = tid; i < M; i += NT) l[i] = tw[i];
  }

  // pass 1: global -> R1 butterfly -> LDS (or straight back to global when M == R1)
  RF_HD static void pass_first(int tid, long long tile, long long nrows, const IO& io, cx* lds) {
    constexpr int R = C::R1, L = M / R;
#pragma unroll
    for (int it = 0; it < ITF; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        cx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = io.load(row, j + m * L);
        DFT<R, DIR>::run(v);
#pragma unroll
        for (int m = 0; m < R; ++m) {
          if (C::NPASS == 1) io.store(row, j * R + m, v[m]);
          else *lds_at(lds, rl, j * R + m) = v[m];
        }
      }
    }
  }

  RF_HD static void pass_mid_read(int tid, const cx* tw, cx* lds, Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, tw_dir<DIR>(tw[stockham_tw_index(j, m, Ns, R, M)]));
          r.v[it][m] = x;
        }
        DFT<R, DIR>::run(r.v[it]);
      }
    }
  }
  RF_HD static void pass_mid_write(int tid, cx* lds, const Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
        const int ob = stockham_out_base(j, Ns, R);
#pragma unroll
        for (int m = 0; m < R; ++m) *lds_at(lds, rl, ob + m * Ns) = r.v[it][m];
      }
    }
  }

  // last pass (NPASS >= 2): LDS -> RL butterfly -> global
  RF_HD static void pass_last(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds) {
    constexpr int R = C::RL, L = M / R;
#pragma unroll
    for (int it = 0; it < C::ITL; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        cx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, tw_dir<DIR>(tw[m * j]));
          v[m] = x;
        }
        DFT<R, DIR>::run(v);
#pragma unroll
        for (int m = 0; m < R; ++m) io.store(row, j + m * L, v[m]);
      }
    }
  }
};

}  // namespace rf
