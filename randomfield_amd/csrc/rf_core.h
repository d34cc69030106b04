// rf_core.h -- arithmetic shared by every kernel of the MI355X random-field path.
//
// Everything here is plain C++ templates marked RF_HD so that the same code is
// (a) inlined into the HIP kernels for gfx950 and (b) compiled by g++ into the
// CPU kernel emulator (csrc/emu) that checks index math, twiddles and phase
// structure in the build container, where there is no GPU.
//
// Reference rows restated here:
//   K  powertools.py:27-61   |k| per cell            -> log10k_cell()
//   T  powertools.py:125-164 sigma(k) table lookup   -> sigma_lookup()
//   R  random.py:12-29       sigma * N(0,1)          -> gen_cell()
//   S  transform.py:141-158  Hermitian symmetrise    -> sym_role(), gen_cell()
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RF_HD __host__ __device__ __forceinline__
#define RF_DEVICE_CODE 1
#else
#define RF_HD inline
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace rf {

// ---------------------------------------------------------------- complex --
template <typename T>
struct cplx {
  T x, y;
};

template <typename T> RF_HD cplx<T> mk(T x, T y) { cplx<T> r; r.x = x; r.y = y; return r; }
template <typename T> RF_HD cplx<T> operator+(cplx<T> a, cplx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> RF_HD cplx<T> operator-(cplx<T> a, cplx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> RF_HD cplx<T> cmul(cplx<T> a, cplx<T> b) {
  return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <typename T> RF_HD cplx<T> cconj(cplx<T> a) { return mk<T>(a.x, -a.y); }
// multiply by +i (DIR=+1) or -i (DIR=-1)
template <int DIR, typename T> RF_HD cplx<T> mul_i(cplx<T> a) {
  return DIR > 0 ? mk<T>(-a.y, a.x) : mk<T>(a.y, -a.x);
}
// conjugate the twiddle for the forward transform: tables hold exp(+2 pi i q/N)
template <int DIR, typename T> RF_HD cplx<T> tw_dir(cplx<T> w) { return DIR > 0 ? w : mk<T>(w.x, -w.y); }

// ------------------------------------------------------- radix butterflies --
// DFT<R, DIR>::run(v): in-place R-point DFT, natural order in and out,
// kernel exp(DIR * 2 pi i n k / R).  DIR=+1 is the (unnormalised) inverse.
template <int R, int DIR> struct DFT;

template <int DIR> struct DFT<1, DIR> {
  template <typename T> RF_HD static void run(cplx<T>*) {}
};

template <int DIR> struct DFT<2, DIR> {
  template <typename T> RF_HD static void run(cplx<T>* v) {
    cplx<T> a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  }
};

template <int DIR> struct DFT<4, DIR> {
  template <typename T> RF_HD static void run(cplx<T>* v) {
    cplx<T> t0 = v[0] + v[2], t1 = v[0] - v[2];
    cplx<T> t2 = v[1] + v[3], t3 = mul_i<DIR>(v[1] - v[3]);
    v[0] = t0 + t2;
    v[1] = t1 + t3;
    v[2] = t0 - t2;
    v[3] = t1 - t3;
  }
};

// exp(DIR * 2 pi i q / 16) for q = 0..9 (products n2*k1 of the 4x4 split)
template <int DIR, typename T> RF_HD cplx<T> w16(int q) {
  const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173, r = (T)0.70710678118654752440;
  T cx, sy;
  switch (q) {
    case 0: cx = 1; sy = 0; break;
    case 1: cx = c1; sy = s1; break;
    case 2: cx = r; sy = r; break;
    case 3: cx = s1; sy = c1; break;
    case 4: cx = 0; sy = 1; break;
    case 6: cx = -r; sy = r; break;
    default: cx = -c1; sy = -s1; break;  // q == 9
  }
  return mk<T>(cx, DIR > 0 ? sy : -sy);
}

template <int DIR> struct DFT<8, DIR> {
  // n = 2 n1 + n2, k = k1 + 4 k2
  template <typename T> RF_HD static void run(cplx<T>* v) {
    cplx<T> a[4] = {v[0], v[2], v[4], v[6]};
    cplx<T> b[4] = {v[1], v[3], v[5], v[7]};
    DFT<4, DIR>::run(a);
    DFT<4, DIR>::run(b);
    const T r = (T)0.70710678118654752440;
    // w8^k1 = exp(DIR 2 pi i k1 / 8)
    b[1] = cmul(b[1], mk<T>(r, DIR > 0 ? r : -r));
    b[2] = mul_i<DIR>(b[2]);
    b[3] = cmul(b[3], mk<T>(-r, DIR > 0 ? r : -r));
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      v[k1] = a[k1] + b[k1];
      v[k1 + 4] = a[k1] - b[k1];
    }
  }
};

template <int DIR> struct DFT<16, DIR> {
  // n = 4 n1 + n2, k = k1 + 4 k2
  template <typename T> RF_HD static void run(cplx<T>* v) {
    cplx<T> a[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
#pragma unroll
      for (int n1 = 0; n1 < 4; ++n1) a[n2][n1] = v[4 * n1 + n2];
      DFT<4, DIR>::run(a[n2]);  // a[n2][k1]
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      cplx<T> c[4];
      c[0] = a[0][k1];
#pragma unroll
      for (int n2 = 1; n2 < 4; ++n2) c[n2] = (k1 == 0) ? a[n2][k1] : cmul(a[n2][k1], w16<DIR, T>(n2 * k1));
      DFT<4, DIR>::run(c);  // c[k2]
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) v[k1 + 4 * k2] = c[k2];
    }
  }
};

// ----------------------------------------------------------- Stockham maps --
// Pass with radix R on sub-transforms of length Ns (Ns = product of earlier
// radices): butterfly j in [0, N/R) reads in[j + m*N/R], multiplies by
// exp(DIR 2 pi i m (j % Ns) / (Ns R)), does an R-point DFT, writes
// out[(j / Ns) * Ns * R + (j % Ns) + m * Ns].  Natural order after the last pass.
RF_HD int stockham_out_base(int j, int Ns, int R) { return (j / Ns) * Ns * R + (j % Ns); }
// index into a length-N table of exp(2 pi i q / N)
RF_HD int stockham_tw_index(int j, int m, int Ns, int R, int N) { return m * (j % Ns) * (N / (Ns * R)); }

// -------------------------------------------------------------------- RNG --
struct PhiloxOut { uint32_t w[4]; };

RF_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

// Philox4x32-R (Salmon, Moraes, Dror, Shaw 2011).  counter = (ctr_lo, ctr_hi) as
// two 64-bit words, key = 64-bit seed.
// A workgroup-uniform 64-bit value pinned to scalar registers: keeps the compiler from re-associating
// (lane part) + (uniform part) sums back into per-lane 64-bit multiplies.
RF_HD uint64_t pin_uniform(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
  return ((uint64_t)hi << 32) | lo;
#else
  return x;
#endif
}

RF_HD long long pin_uniform_ll(long long x) { return (long long)pin_uniform((uint64_t)x); }

RF_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);      // one v_bitop3_b32 instead of two v_xor_b32
#else
  return a ^ b ^ c;
#endif
}

// Rounds of the NATIVE stream: Philox4x32-7 is the smallest round count the authors found Crush-resistant (SC'11, table 2: it
// passes BigCrush); the library default of 10 is a safety margin.  The generation pass is VALU-bound and a Philox round costs
// two v_mad_u64_u32 + two v_bitop3 per lane-load: 7 instead of 10 rounds is -6 % of its instructions (measured -1.3 % of a
// 1024^3 realisation).  The test-side restatement of the stream uses the same count; both are pinned by Random123's
// published known-answer vectors for 7 and 10 rounds (tests/test_oracle_golden.py).
constexpr int PHILOX_ROUNDS = 7;          // the native stream's definition (CHANGES.md): Philox4x32-7
template <int ROUNDS = PHILOX_ROUNDS>
RF_HD PhiloxOut philox4x32(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
  uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    // one 32x32->64 product each (v_mad_u64_u32), not separate mul_lo / mul_hi
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
    const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    uint32_t n0 = xor3(h1, c1, k0), n2 = xor3(h0, c3, k1);
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  PhiloxOut o; o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
  return o;
}
RF_HD PhiloxOut philox_native(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) { return philox4x32<PHILOX_ROUNDS>(ctr_lo, ctr_hi, key); }

// Box-Muller from two 32-bit words: u = (w + 0.5) / 2^32 in (0, 1).
template <typename T> struct BoxMuller;
template <> struct BoxMuller<double> {
  RF_HD static void run(uint32_t wa, uint32_t wb, double& g0, double& g1) {
    double u1 = ((double)wa + 0.5) * (1.0 / 4294967296.0);
    double u2 = ((double)wb + 0.5) * (1.0 / 4294967296.0);
    double r = sqrt(-2.0 * log(u1));
#if defined(__HIP_DEVICE_COMPILE__)
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);       // no Payne-Hanek style reduction: the argument is an exact multiple of pi
    g0 = r * cs;
    g1 = r * sn;
#else
    double a = (2.0 * M_PI) * u2;
    g0 = r * cos(a);
    g1 = r * sin(a);
#endif
  }
};
template <> struct BoxMuller<float> {
  // (g0, g1) = scale * N(0,1) pair.
  // u1 = (w + 1/2) / 2^32 formed as ONE fused multiply-add of float(w): the float keeps 24 significant bits at
  // any magnitude, so the radius resolves the tail down to u1 = 2^-33 (6.7 sigma); u1 is never 0 (it may round
  // to exactly 1, giving radius 0).  The angle uses the top 23 bits of its word: u2 = (w >> 9) / 2^23 revolutions.
  RF_HD static void run_scaled(uint32_t wa, uint32_t wb, float scale, float& g0, float& g1) {
    const float u1 = fmaf((float)wa, 1.0f / 4294967296.0f, 1.0f / 8589934592.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    // raw v_log_f32 / v_sqrt_f32: u1 >= 2^-33 and -2 ln u1 in [0, 46) need no denormal fix-ups.
    // -2 ln u = (-2 ln 2) log2 u
    const float r = scale * __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    // the float with exponent 0 and mantissa w >> 9 is 1 + u2 (one v_alignbit_b32); v_sin_f32 / v_cos_f32 take
    // their argument in revolutions and reduce it themselves
    const float rev = __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, wb, 9));
    g0 = r * __builtin_amdgcn_cosf(rev);
    g1 = r * __builtin_amdgcn_sinf(rev);
#else
    const float r = scale * sqrtf(-2.0f * logf(u1));
    const float a = (float)(2.0 * M_PI) * ((float)(wb >> 9) * (1.0f / 8388608.0f));
    g0 = r * cosf(a);
    g1 = r * sinf(a);
#endif
  }
  RF_HD static void run(uint32_t wa, uint32_t wb, float& g0, float& g1) { run_scaled(wa, wb, 1.0f, g0, g1); }
};

// ------------------------------------------------------------ generation --
enum { NOISE_PHILOX = 0, NOISE_EXTERNAL = 1 };

struct GenParams {
  int nx, ny, nz;             // real-space grid
  const double* kx2;          // [nx]      k_x(i)^2, float64, host-computed as powertools.py:27-37
  const double* ky2;          // [ny]
  const double* kz2;          // [nz/2+1]
  const double* xt;           // [nt] log10 k_i      (powertools.py:153)
  const double* st;           // [nt] sigma_i        (powertools.py:154)
  const double* sl;           // [nt-1] (st[j+1]-st[j])/(xt[j+1]-xt[j])
  const int* bin;             // [nbins] acceleration grid: lower bound of the interval index per bin
  int nt, nbins;
  double x0, inv_dx;          // bin b covers x0 + [b, b+1) / inv_dx
  int noise_mode;
  uint64_t seed;
  const uint64_t* seed_dev;   // if non-null the seed is read from device memory (graph replay)
  const double* noise;        // external mode: 2*nx*ny*(nz/2+1) float64 deviates, reference order
  // The API-layout side arrays (noise, k space, potential) of a kz-slab rank hold its planes only: rows of `zpitch`
  // slots, slot j < zpitch - 1 = plane zoff + j, the last slot = the Nyquist plane kz = nz/2 (which rank 0 packs into
  // its slot kz = 0).  One rank: zpitch = nz/2 + 1, zoff = 0 -- the reference's layout itself.
  int zpitch, zoff;
};

// slot of plane kz in a row of a side array (GenParams / FastGenParams)
template <class G> RF_HD int side_slot(const G& g, int kz) { return kz == g.nz / 2 ? g.zpitch - 1 : kz - g.zoff; }

// no-FMA arithmetic: these must round exactly like numpy's separate ufunc calls / numpy's C code built
// without FMA.  (hipcc contracts a*b+c by default and its __dmul_rn/__dadd_rn are plain operators, so the
// contraction is switched off by pragma; the host build parks the product in a volatile.)
RF_HD double mul_then_add(double a, double b, double c) {
#if defined(__clang__)
#pragma clang fp contract(off)
  const double p = a * b;
  return p + c;
#else
  volatile double p = a * b;
  return p + c;
#endif
}
RF_HD double sum_of_squares(double a, double b) {
#if defined(__clang__)
#pragma clang fp contract(off)
  const double p = a * a, q = b * b;
  return p + q;
#else
  volatile double p = a * a, q = b * b;
  return p + q;
#endif
}

// Row T: piecewise-linear sigma(log10 k), 0 outside the table (powertools.py:155-157,163).
RF_HD double sigma_lookup(const GenParams& g, double x) {
  const int n = g.nt;
  if (!(x >= g.xt[0] && x <= g.xt[n - 1])) return 0.0;  // also -inf (DC cell) and NaN
  if (x == g.xt[n - 1]) return g.st[n - 1];
  int b = (int)((x - g.x0) * g.inv_dx);
  b = b < 0 ? 0 : (b >= g.nbins ? g.nbins - 1 : b);
  int j = g.bin[b];
  while (j > 0 && g.xt[j] > x) --j;
  while (j < n - 2 && g.xt[j + 1] <= x) ++j;
  return mul_then_add(g.sl[j], x - g.xt[j], g.st[j]);
}

// Row K + T: sigma of cell (ix, iy, iz), rounded to the array's real type.
template <typename T> RF_HD T sigma_cell(const GenParams& g, int ix, int iy, int iz);
template <> RF_HD float sigma_cell<float>(const GenParams& g, int ix, int iy, int iz) {
  float t = (float)(g.kx2[ix] + g.ky2[iy]);          // powertools.py:50
  t = (float)((double)t + g.kz2[iz]);                 // powertools.py:52
  t = log10f(t) * 0.5f;                               // powertools.py:56,60  (-inf at DC)
  return (float)sigma_lookup(g, (double)t);           // powertools.py:163
}
template <> RF_HD double sigma_cell<double>(const GenParams& g, int ix, int iy, int iz) {
  double t = g.kx2[ix] + g.ky2[iy];
  t = t + g.kz2[iz];
  t = log10(t) * 0.5;
  return sigma_lookup(g, t);
}

// Native noise index of API cell (ix, iy, iz): cells with iz < nz/2 are numbered
// in the device-internal order [ix][iy][nz/2]; the Nyquist plane follows.
RF_HD uint64_t native_noise_index(const GenParams& g, int ix, int iy, int iz) {
  const uint64_t nzc = (uint64_t)(g.nz / 2);
  const uint64_t col = (uint64_t)ix * (uint64_t)g.ny + (uint64_t)iy;
  return iz < g.nz / 2 ? col * nzc + (uint64_t)iz : (uint64_t)g.nx * (uint64_t)g.ny * nzc + col;
}

// Row R deviates (g_re, g_im) of the cell that OWNS the draw.
template <typename T>
RF_HD void noise_of_cell(const GenParams& g, uint64_t seed, int ix, int iy, int iz, double& gre, double& gim) {
  if (g.noise_mode == NOISE_EXTERNAL) {
    const uint64_t c = ((uint64_t)ix * (uint64_t)g.ny + (uint64_t)iy) * (uint64_t)g.zpitch + (uint64_t)side_slot(g, iz);
    gre = g.noise[2 * c];
    gim = g.noise[2 * c + 1];
  } else {
    const uint64_t ci = native_noise_index(g, ix, iy, iz);
    PhiloxOut o = philox_native(ci >> 1, 0, seed);
    T a, b;
    if (ci & 1) BoxMuller<T>::run(o.w[2], o.w[3], a, b);
    else        BoxMuller<T>::run(o.w[0], o.w[1], a, b);
    gre = (double)a;
    gim = (double)b;
  }
}

// Row S roles inside a kz in {0, nz/2} plane (transform.py:141-158).
enum { RF_SRC = 0, RF_DEST = 1, RF_SELF = 2 };
RF_HD int sym_role(int nx, int ny, int ix, int iy) {
  const bool xe = (ix == 0) || (ix == nx / 2);
  const bool ye = (iy == 0) || (iy == ny / 2);
  if (xe && ye) return RF_SELF;
  if (iy > ny / 2) return RF_DEST;
  if (ye && ix > nx / 2) return RF_DEST;
  return RF_SRC;
}

// Rows K,T,R,S for one API cell (ix, iy, iz), iz in [0, nz/2].
template <typename T>
RF_HD cplx<T> gen_cell(const GenParams& g, uint64_t seed, int ix, int iy, int iz) {
  int sx = ix, sy = iy, role = RF_SRC;
  if (iz == 0 || iz == g.nz / 2) {
    role = sym_role(g.nx, g.ny, ix, iy);
    if (role == RF_DEST) {  // take conj of the draw at (-ix, -iy)
      sx = (g.nx - ix) % g.nx;
      sy = (g.ny - iy) % g.ny;
    }
  }
  const T s = sigma_cell<T>(g, sx, sy, iz);
  double gre, gim;
  noise_of_cell<T>(g, seed, sx, sy, iz, gre, gim);
  T re = (T)((double)s * gre);       // random.py:28: product in float64, rounded once
  T im = (T)((double)s * gim);
  if (role == RF_DEST) im = -im;
  if (role == RF_SELF) im = (T)0;    // transform.py:154-156
  if (ix == 0 && iy == 0 && iz == 0) re = (T)0;  // transform.py:158
  return mk<T>(re, im);
}

// Device-internal packed cell: kz in [0, nz/2); slot kz = 0 carries
// A(kz=0) + i * A(kz=nz/2)  (both planes are 2-D Hermitian, so after the x and
// y inverse passes they are real and occupy re / im of one complex plane).
template <typename T>
RF_HD cplx<T> gen_packed(const GenParams& g, uint64_t seed, int ix, int iy, int kz) {
  cplx<T> a = gen_cell<T>(g, seed, ix, iy, kz);
  if (kz == 0) {
    cplx<T> n = gen_cell<T>(g, seed, ix, iy, g.nz / 2);
    a = mk<T>(a.x - n.y, a.y + n.x);
  }
  return a;
}

// ------------------------------------------------- fast native generation --
// float32 plans with the native RNG do rows K,T,R,S entirely in float32: the
// values are this repo's own definition (checked against the oracle's
// restatement of the same stream to ~1e-6), so the reference's float64
// rounding chain -- which only matters for same-noise parity -- is not needed.
struct FastRec {          // 16 bytes per bin of the uniform acceleration grid in x = log10 k
  float v0;               // sigma at the bin's left edge
  float sa;               // slope of the piece at the left edge, per bin width
  float fs;               // position of the knot inside the bin in bin units (>= 1 if none)
  float ds;               // slope change at the knot, per bin width
};                        // sigma(f) = v0 + sa * f + ds * max(f - fs, 0),  f in [0, 1) the position inside the bin

// (where the float32 deviate pairs of one row of the stream live: see make_rowloc / row_pair below)
struct RowLoc {
  uint32_t off;                 // in-segment slot of the row's cell kz = 0
  uint32_t seg_n;               // (seg << 11) | nfirst,  1 <= nfirst <= nz/2 + 1 <= 1025
};
enum { ROWLOC_NBITS = 11 };

struct FastGenParams {
  int nx, ny, nz;
  float dkx, dky, dkz;    // k_axis(i) = dk_axis * signed fftfreq index (2 pi / (n spacing)): the kernels form |k|^2
                          // arithmetically, so the generation loop reads no global memory at all (on gfx950 a load
                          // would also wait for every store issued before it -- one counter for both)
  const FastRec* rec;     // global copy of the records; the kernels stage them in LDS when nbins <= FAST_LDS_BINS
  int nbins;
  float u_scale, u_off;   // bin coordinate u = log2(k^2) * u_scale + u_off
  uint64_t seed;
  const uint64_t* seed_dev;
  const double* noise;    // SRC = 1 kernels: resident float64 deviates in the reference's order (random.py:24-28),
                          // 2 per cell of the API layout [nx][ny][nz/2+1]
  int zpitch, zoff;       // row pitch and first plane of the side arrays (noise, potential): see GenParams
  int ppitch;             // row pitch (cells) of the POTENTIAL array: zpitch rounded up to even for float32 plans, so that the
                          // generation pass stores a cell pair (kz even, kz + 1) with one aligned 16-byte store
  // SRC = 2 kernels: the same deviates as float32 pairs (g_re, g_im), read where the one-pass replay left them -- every
  // MT19937 segment's accepted pairs densely from slot seg * seg_cap of `noise32`.  No copy into cell order: `rowtab` (built from
  // the scan of the per-segment counts by mt_rowtab_kernel, one 8-byte entry per row (ix, iy) of the stream, index iy * nx + ix so
  // that the rows of one x-pass tile are neighbours) says where a row's nz/2 + 1 consecutive stream cells start.
  const cplx<float>* noise32;
  const RowLoc* rowtab;
  unsigned long long seg_cap;          // attempts (= slots) per segment (< 2^32)
  // POT = 2 kernels: the pass emits pscale * delta(k) / k^2 instead of delta(k) -- the saved potential of generate.py:200-217
  // regenerated from the seed (or the resident deviates) when calculate_newtonian_potential asks for it, never stored
  double pscale;          // (float32 plans round it to float32, as the scaled copy of the stored potential does)
  int emit_potential;     // host side: selects those kernels
};
enum { FAST_LDS_BINS = 512 };

RF_HD float fast_log2(float t) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_logf(t);   // v_log_f32
#else
  return log2f(t);
#endif
}

// rec points at the records (LDS or global)
RF_HD float fast_sigma(const FastGenParams& g, const FastRec* rec, float t /* |k|^2 */) {
  float u = fast_log2(t) * g.u_scale + g.u_off;          // (log10|k| - x0) / dx
  const float hi = (float)g.nbins - 0.001f;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __builtin_amdgcn_fmed3f(u, 0.0f, hi);              // clamp; t == 0 gives -inf -> bin 0 (a guard bin)
  const float f = __builtin_amdgcn_fractf(u);
#else
  u = fminf(fmaxf(u, 0.0f), hi);
  const float f = u - floorf(u);
#endif
  const int b = (int)u;
  const FastRec r = rec[b];
  const float d = fmaxf(f - r.fs, 0.0f);
  return (r.v0 + r.sa * f) + r.ds * d;
}

// signed fftfreq index of ix without a select: nx is a power of two, so (ix & nx/2) is 0 or nx/2
RF_HD int fast_signed_index(int ix, int nx) { return ix - 2 * (ix & (nx >> 1)); }
// kx^2 + ky^2 of column (ix, iy), and |k|^2 of its cell kz (0 <= kz <= nz/2: no wrap on the half axis)
RF_HD float fast_kxy2(const FastGenParams& g, int ix, int iy) {
  const float kx = (float)fast_signed_index(ix, g.nx) * g.dkx, ky = (float)fast_signed_index(iy, g.ny) * g.dky;
  return fmaf(kx, kx, ky * ky);
}
RF_HD float fast_k2(const FastGenParams& g, float kxy, int kz) {
  const float k = (float)kz * g.dkz;
  return fmaf(k, k, kxy);
}

// Philox counter of the cell pair (kz even, kz + 1) of column (ix, iy): half the native noise index
RF_HD uint64_t fast_pair_counter(const FastGenParams& g, int ix, int iy, int kz) {
  return (((uint64_t)ix * (uint64_t)g.ny + (uint64_t)iy) * (uint64_t)(g.nz / 2) + (uint64_t)kz) >> 1;
}

// The two packed cells kz, kz + 1 of one column from ONE Philox call: k^2 = kxy + kz2a / kz2b.
RF_HD void fast_gen_pair_at(const FastGenParams& g, const FastRec* rec, uint64_t seed, uint64_t ctr, float k2a,
                            float k2b, cplx<float>& c0, cplx<float>& c1) {
  const PhiloxOut o = philox_native(ctr, 0, seed);
  float g0, g1;
  const float s0 = fast_sigma(g, rec, k2a), s1 = fast_sigma(g, rec, k2b);
  BoxMuller<float>::run_scaled(o.w[0], o.w[1], s0, g0, g1);
  c0 = mk<float>(g0, g1);
  BoxMuller<float>::run_scaled(o.w[2], o.w[3], s1, g0, g1);
  c1 = mk<float>(g0, g1);
}
RF_HD void fast_gen_pair(const FastGenParams& g, const FastRec* rec, uint64_t seed, int ix, int iy, int kz,
                         cplx<float>& c0, cplx<float>& c1) {
  const float kxy = fast_kxy2(g, ix, iy);
  fast_gen_pair_at(g, rec, seed, fast_pair_counter(g, ix, iy, kz), fast_k2(g, kxy, kz), fast_k2(g, kxy, kz + 1), c0, c1);
}

// One packed cell with native noise index ci (float64 plans: one complex128 per lane, so the two cells of a Philox
// pair sit in neighbouring lanes; each lane runs the pair's call and keeps its own half).
RF_HD cplx<float> fast_gen_one(const FastGenParams& g, const FastRec* rec, uint64_t seed, uint64_t ci, float k2) {
  const PhiloxOut o = philox_native(ci >> 1, 0, seed);
  const bool odd = (ci & 1u) != 0;
  float g0, g1;
  BoxMuller<float>::run_scaled(odd ? o.w[2] : o.w[0], odd ? o.w[3] : o.w[1], fast_sigma(g, rec, k2), g0, g1);
  return mk<float>(g0, g1);
}

// slot kz = 0 of column (ix, iy): (plane kz=0) + i (plane kz=nz/2), each Hermitian-symmetrised
// by the rules of gen_cell() (transform.py:141-158)
RF_HD float fast_rcp(float t) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(t);   // v_rcp_f32 (1 ulp)
#else
  return 1.0f / t;
#endif
}

// p0, pn: delta(k) / k^2 of the symmetrised cells kz = 0 and kz = nz/2 (0 at DC) for the save_potential store
RF_HD cplx<float> fast_fix_kz0(const FastGenParams& g, const FastRec* rec, uint64_t seed, int ix, int iy,
                               cplx<float>& p0, cplx<float>& pn) {
  const int nzc = g.nz / 2;
  const int role = sym_role(g.nx, g.ny, ix, iy);
  int sx = ix, sy = iy;
  if (role == RF_DEST) { sx = (g.nx - ix) % g.nx; sy = (g.ny - iy) % g.ny; }
  const float kxy_s = fast_kxy2(g, sx, sy);
  const uint64_t scol = (uint64_t)sx * (uint64_t)g.ny + (uint64_t)sy;
  float g0, g1;
  const float s0 = fast_sigma(g, rec, fast_k2(g, kxy_s, 0));
  const PhiloxOut os = philox_native((scol * (uint64_t)nzc) >> 1, 0, seed);
  BoxMuller<float>::run_scaled(os.w[0], os.w[1], s0, g0, g1);
  cplx<float> a = mk<float>(g0, g1);
  const float sn = fast_sigma(g, rec, fast_k2(g, kxy_s, nzc));
  const uint64_t cn = (uint64_t)g.nx * (uint64_t)g.ny * (uint64_t)nzc + scol;
  const PhiloxOut on = philox_native(cn >> 1, 0, seed);
  if (cn & 1) BoxMuller<float>::run_scaled(on.w[2], on.w[3], sn, g0, g1);
  else        BoxMuller<float>::run_scaled(on.w[0], on.w[1], sn, g0, g1);
  cplx<float> n = mk<float>(g0, g1);
  if (role == RF_DEST) { a.y = -a.y; n.y = -n.y; }
  if (role == RF_SELF) { a.y = 0.0f; n.y = 0.0f; }
  if (ix == 0 && iy == 0) a = mk<float>(0.0f, 0.0f);     // DC mode (its sigma lookup is meaningless)
  // the potential of a destination cell is the conjugate of its source's: same |k|^2 on both sides of the mirror
  const float r0 = (ix == 0 && iy == 0) ? 0.0f : fast_rcp(fast_k2(g, kxy_s, 0)), rn = fast_rcp(fast_k2(g, kxy_s, nzc));
  p0 = mk<float>(a.x * r0, a.y * r0);
  pn = mk<float>(n.x * rn, n.y * rn);
  return mk<float>(a.x - n.y, a.y + n.x);
}

// The same slot from resident deviates (the reference's stream, e.g. replayed MT19937): cell = sigma * (g_re + i g_im)
// with the float64 product rounded once (random.py:28), symmetrised as above.  SRC = 1: float64 deviates; SRC = 2: the
// float32 copies (float32 plans: the product is then formed in float32, 6e-8 relative from the once-rounded one).
// one float32 deviate pair through a GLOBAL-address-space pointer: its address comes out of selects, the compiler cannot
// infer where it points and would emit flat_load -- counted on lgkmcnt too, so that the LDS reads of the sigma lookup
// would wait for the deviates' trip to HBM
RF_HD cplx<float> load_pair_global(const cplx<float>* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = *(const __attribute__((address_space(1))) f2*)p;
  return mk<float>(v.x, v.y);
#else
  return *p;
#endif
}
// Where the pairs of one row of the stream live (SRC = 2).  Stream cell c = (ix ny + iy) (nz/2 + 1) + kz; the cells of segment s
// are [first[s], first[s + 1]) (exclusive scan of the per-segment counts) and sit densely from slot s * cap.  A row is nz/2 + 1
// consecutive cells, far fewer than a segment holds (the builder checks it): its first `nfirst` cells lie in segment `seg` from
// in-segment slot `off`, the others -- a segment boundary falls inside one row in a few hundred -- at the start of segment seg + 1.
// entry of the row whose first cell is c.  first: [nseg + 1] with first[nseg] = total accepted pairs.  Rows that reach beyond the
// total (a failed replay: the host reports it) point at the start of segment 0 -- in bounds, never used.  *bad is set when a row
// would span more than two segments (segments shorter than a row: the host refuses such a replay geometry).
RF_HD RowLoc make_rowloc(const unsigned long long* first, int nseg, unsigned long long c, unsigned nzh, int* bad) {
  RowLoc e;
  e.off = 0; e.seg_n = nzh;
  if (c + nzh > first[nseg]) return e;
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {                                   // largest s with first[s] <= c
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= c) lo = mid; else hi = mid - 1;
  }
  const unsigned long long left = first[lo + 1] - c;  // cells of segment lo from c on (>= 1)
  const unsigned nfirst = left < nzh ? (unsigned)left : nzh;
  if (nfirst < nzh && first[lo + 2 <= nseg ? lo + 2 : nseg] < c + nzh) *bad = 1;
  e.off = (uint32_t)(c - first[lo]);
  e.seg_n = ((uint32_t)lo << ROWLOC_NBITS) | nfirst;
  return e;
}
RF_HD RowLoc load_rowloc(const RowLoc* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const u2 v = *(const __attribute__((address_space(1))) u2*)p;      // one global_load_dwordx2 (not a flat load: see load_pair_global)
  RowLoc e; e.off = v.x; e.seg_n = v.y;
  return e;
#else
  return *p;
#endif
}
// address of the pair of cell kz of the row
RF_HD const cplx<float>* row_pair(const FastGenParams& g, RowLoc e, int kz) {
  const uint32_t nfirst = e.seg_n & ((1u << ROWLOC_NBITS) - 1u), seg = e.seg_n >> ROWLOC_NBITS, cap = (uint32_t)g.seg_cap;
  const uint32_t slot = (uint32_t)kz < nfirst ? e.off + (uint32_t)kz : cap + ((uint32_t)kz - nfirst);
  return g.noise32 + ((unsigned long long)seg * cap + slot);
}

template <int SRC>
RF_HD cplx<float> fast_noise_cell(const FastGenParams& g, const FastRec* rec, int ix, int iy, int kz, float k2) {
  if (SRC == 2) {
    const float s = fast_sigma(g, rec, k2);
    const cplx<float> d = load_pair_global(row_pair(g, load_rowloc(g.rowtab + ((long long)iy * g.nx + ix)), kz));
    return mk<float>(s * d.x, s * d.y);
  }
  const long long c = ((long long)ix * g.ny + iy) * g.zpitch + side_slot(g, kz);
  const double* d = g.noise + 2 * c;
  const double s = (double)fast_sigma(g, rec, k2);
  return mk<float>((float)(s * d[0]), (float)(s * d[1]));
}
// p0, pn: delta(k) / k^2 of the two symmetrised cells, as fast_fix_kz0()
template <int SRC>
RF_HD cplx<float> fast_fix_kz0_noise(const FastGenParams& g, const FastRec* rec, int ix, int iy, cplx<float>& p0, cplx<float>& pn) {
  const int nzc = g.nz / 2;
  const int role = sym_role(g.nx, g.ny, ix, iy);
  int sx = ix, sy = iy;
  if (role == RF_DEST) { sx = (g.nx - ix) % g.nx; sy = (g.ny - iy) % g.ny; }
  const float kxy_s = fast_kxy2(g, sx, sy);
  cplx<float> a = fast_noise_cell<SRC>(g, rec, sx, sy, 0, fast_k2(g, kxy_s, 0));
  cplx<float> n = fast_noise_cell<SRC>(g, rec, sx, sy, nzc, fast_k2(g, kxy_s, nzc));
  if (role == RF_DEST) { a.y = -a.y; n.y = -n.y; }
  if (role == RF_SELF) { a.y = 0.0f; n.y = 0.0f; }
  if (ix == 0 && iy == 0) a = mk<float>(0.0f, 0.0f);
  const float r0 = (ix == 0 && iy == 0) ? 0.0f : fast_rcp(fast_k2(g, kxy_s, 0)), rn = fast_rcp(fast_k2(g, kxy_s, nzc));
  p0 = mk<float>(a.x * r0, a.y * r0);
  pn = mk<float>(n.x * rn, n.y * rn);
  return mk<float>(a.x - n.y, a.y + n.x);
}

// ------------------------------------------------------- c2r / r2c untangle --
// Inverse: Zc[k] = (X[k] + conj X[M-k]) + i t_k (X[k] - conj X[M-k]),  t_k = exp(+2 pi i k / N), N = 2M.
// The length-M unnormalised inverse FFT of Zc is z[m] = x[2m] + i x[2m+1] (x unnormalised, i.e. N * irfft).
template <typename T> RF_HD cplx<T> c2r_untangle(cplx<T> xk, cplx<T> xmk, cplx<T> tk) {
  cplx<T> a = mk<T>(xk.x + xmk.x, xk.y - xmk.y);
  cplx<T> d = mk<T>(xk.x - xmk.x, xk.y + xmk.y);
  cplx<T> b = cmul(tk, d);
  return mk<T>(a.x - b.y, a.y + b.x);
}
// Forward: given Z = FFT_M(z) (kernel exp(-...)), X[k] = (Z[k] + conj Z[M-k])/2 - (i/2) conj(t_k) (Z[k] - conj Z[M-k]).
template <typename T> RF_HD cplx<T> r2c_tangle(cplx<T> zk, cplx<T> zmk, cplx<T> tk) {
  cplx<T> a = mk<T>(zk.x + zmk.x, zk.y - zmk.y);
  cplx<T> d = mk<T>(zk.x - zmk.x, zk.y + zmk.y);
  cplx<T> b = cmul(cconj(tk), d);
  return mk<T>((T)0.5 * (a.x + b.y), (T)0.5 * (a.y - b.x));
}

// LDS padding: one extra slot every 16 so that stride-16 (and stride-8) write
// patterns of the first Stockham pass spread over the banks.
// LDS row image of the contiguous-axis passes: one complex of padding per 8 (a bank-conflict simulation of the
// radix-8 passes gives 1.33x the conflict-free cycles for this shift against 1.83x for one per 16)
RF_HD int pad16(int i) { return i + (i >> 3); }

}  // namespace rf
