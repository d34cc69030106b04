// rf_generic.h -- transforms for grids whose axes are NOT powers of two.
//
// The reference accepts any even nx, ny, nz (transform.py:172-177) and its own tests run on (4, 6, 8) and
// (40, 60, 80) (tests/test_transform.py:11, tests/test_random.py:12-22).  The tiled power-of-two kernels of
// rf_fft.h do not cover those; this file does, with one mixed-radix Stockham line transform whose radices are
// run-time values (any factorisation of the axis length; a radix-R butterfly is evaluated as R dot products
// of length R, so the cost per line is n * (sum of the factors)).  It is the small-grid / odd-shape path:
// correctness and the reference's semantics first, coalesced accesses where the layout gives them, no
// tuning beyond that.  Arrays are in the API layout throughout ([nx][ny][nz/2+1] complex half spectrum,
// dense [nx][ny][nz] reals), so the kz = 0 / nz/2 planes need no special packing: the contiguous pass drops
// the imaginary parts of the DC and Nyquist bins of every row exactly as numpy's irfft does (transform.py:314).
//
// Every function takes (tid, nth) and a `sync` callable: the kernels pass (threadIdx.x, blockDim.x,
// __syncthreads) and the CPU emulator (0, 1, no-op), which executes the same statements in the same order.
#pragma once
#include "rf_core.h"

namespace rf {

// An axis is transformed with its whole line (two buffers of n elements) in LDS: 160 KB per workgroup hold lines of up to 8192 complex64
// or 4096 complex128 (generic_max_axis); beyond 64 KB the kernels' dynamic-LDS attribute is raised (rf_k_generic.hip).
enum { GENERIC_MAX_N = 8192, GENERIC_MAX_FACTORS = 12, GENERIC_LDS_MAX = 160 * 1024 - 256 };
inline int generic_max_axis(int f64) { return f64 ? 4096 : 8192; }

struct GenericAxis {
  int n;                                // line length
  int nf;                               // number of radices
  int f[GENERIC_MAX_FACTORS];           // their product is n
  // derived by generic_factor (the kernels read these as uniform values from their argument block):
  int smooth;                           // all radices among 2, 3, 4, 5: the in-place form (see generic_stage_inplace)
  int w[GENERIC_MAX_FACTORS];           // Ns of stage s = f[0] ... f[s-1]
  unsigned fm[GENERIC_MAX_FACTORS];     // multiply-high reciprocals of f[s] and of w[s] (FastDiv)
  unsigned wm[GENERIC_MAX_FACTORS];
};

// reciprocal for division by multiply-high (FastDiv below): floor(2^32 / d) + 1, d >= 2
inline unsigned generic_magic(unsigned d) { return d > 1 ? 0xFFFFFFFFu / d + 1u : 0u; }

// radices of n: 4s first (fewest stages), then 2, then the odd primes in increasing order
inline bool generic_factor(int n, GenericAxis& ax) {
  ax.n = n;
  ax.nf = 0;
  if (n < 1 || n > GENERIC_MAX_N) return false;
  int m = n;
  auto push = [&](int r) { if (ax.nf < GENERIC_MAX_FACTORS) ax.f[ax.nf] = r; ++ax.nf; m /= r; };
  while (m % 4 == 0) push(4);
  while (m % 2 == 0) push(2);
  // a 4 and the lone 2 behind it as ONE radix-8 stage (a radix-2 stage costs as much index arithmetic per butterfly as a radix-4 one,
  // for half the elements)
  if (ax.nf >= 2 && ax.nf <= GENERIC_MAX_FACTORS && ax.f[ax.nf - 1] == 2 && ax.f[ax.nf - 2] == 4) { ax.f[ax.nf - 2] = 8; --ax.nf; }
  for (int p = 3; p * p <= m; p += 2)
    while (m % p == 0) push(p);
  if (m > 1) push(m);
  if (ax.nf > GENERIC_MAX_FACTORS) return false;
  ax.smooth = 1;
  int Ns = 1;
  for (int s = 0; s < GENERIC_MAX_FACTORS; ++s) {
    if (s >= ax.nf) { ax.f[s] = 1; ax.w[s] = n; ax.fm[s] = 0; ax.wm[s] = generic_magic((unsigned)n); continue; }
    if (ax.f[s] > 5 && ax.f[s] != 8) ax.smooth = 0;
    ax.w[s] = Ns;
    ax.fm[s] = generic_magic((unsigned)ax.f[s]);
    ax.wm[s] = generic_magic((unsigned)Ns);
    Ns *= ax.f[s];
  }
  return true;
}

// Division by a run-time constant that is the same for all threads: q = (a * m) >> 32 with m = floor(2^32 / d) + 1 is exact whenever
// a * d < 2^32 (the error term a * (m d - 2^32) / (d 2^32) stays below 1 / d).  Here a < n * TC <= 2^17 and d <= 2^13.  A 32-bit
// integer division costs ~40 instructions on this hardware, and the loops below need three to five per element.
struct FastDiv {
  uint32_t d, m;
  RF_HD explicit FastDiv(uint32_t dd) : d(dd), m(dd > 1 ? 0xFFFFFFFFu / dd + 1u : 0u) {}
  RF_HD FastDiv(uint32_t dd, uint32_t mm) : d(dd), m(mm) {}                        // (reciprocal formed on the host: generic_magic)
  RF_HD uint32_t div(uint32_t a) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return d > 1 ? __umulhi(a, m) : a;
#else
    return d > 1 ? (uint32_t)(((uint64_t)a * m) >> 32) : a;
#endif
  }
  RF_HD void divmod(uint32_t a, uint32_t& q, uint32_t& r) const { q = div(a); r = a - q * d; }
};

// How a block's threads walk the n * TC elements of its LDS image (element e of line c at [e * TC + c]): thread tid takes
// idx = tid, tid + nth, ...  When nth is a multiple of TC (every kernel launch) a thread stays on ONE line c = tid % TC and its e
// advances by nth / TC: no division per element; otherwise (the emulator's single thread) c and e come from idx.
struct GenericWalk {
  int TC, total, nth, c0, e0, estep;
  bool fixed;
  FastDiv dtc;
  RF_HD GenericWalk(int n, int TC_, int tid, int nth_) : TC(TC_), total(n * TC_), nth(nth_), c0(0), e0(0), estep(0), fixed(nth_ % TC_ == 0), dtc((uint32_t)TC_) {
    if (fixed) { c0 = tid % TC; e0 = tid / TC; estep = nth / TC; }
  }
  RF_HD void at(int idx, int i, int& c, int& e) const {
    if (fixed) { c = c0; e = e0 + i * estep; }
    else { uint32_t q, r; dtc.divmod((uint32_t)idx, q, r); c = (int)r; e = (int)q; }
  }
};

// One Stockham stage of radix R on TC interleaved lines (element e of line c at [e * TC + c]): output o of a line is
//   out[o] = sum_r in[j + r n/R] * exp(sign 2 pi i r (k / (Ns R) + u / R)),   o = jhi Ns R + u Ns + k,  j = jhi Ns + k
// (Ns = product of the radices already applied).  `root` holds exp(+2 pi i t / (n * rstep)), t in [0, n * rstep).
//
// Any radix: every OUTPUT is a dot product of length R (R table reads and R LDS reads per element).
template <typename T>
RF_HD void generic_stage_any(const cplx<T>* in, cplx<T>* out, int n, int TC, int P, int R, int Ns, const cplx<T>* root, int rstep,
                             int sign, int tid, int nth) {
  const int m = n / R, unit = n / (Ns * R);
  const GenericWalk walk(n, TC, tid, nth);
  const FastDiv dNs((uint32_t)Ns), dR((uint32_t)R);
  const T sg = sign < 0 ? (T)-1 : (T)1;
  int i = 0;
  for (int idx = tid; idx < walk.total; idx += nth, ++i) {
    int c, o;
    walk.at(idx, i, c, o);
    uint32_t t, k, jhi, u;
    dNs.divmod((uint32_t)o, t, k);                 // k = o % Ns, t = o / Ns
    dR.divmod(t, jhi, u);                          // u = t % R,  jhi = t / R
    const int j = (int)jhi * Ns + (int)k;
    int q = (int)k * unit + (int)u * m;            // < n / R + n
    if (q >= n) q -= n;
    T sr = (T)0, si = (T)0;
    int ri = 0;
    const cplx<T>* pin = in + j * P + c;
    const int mstep = m * P;
    for (int r = 0; r < R; ++r) {
      const cplx<T> v = pin[r * mstep];
      cplx<T> w = root[ri * rstep];
      w.y *= sg;
      sr += v.x * w.x - v.y * w.y;
      si += v.x * w.y + v.y * w.x;
      ri += q;
      if (ri >= n) ri -= n;
    }
    out[o * P + c] = mk<T>(sr, si);
  }
}

// Radices 2, 3, 4, 5, 8 (all but the large prime factors of an axis): one thread per BUTTERFLY -- R inputs, their R - 1 twiddles
// w^(r k) (one table read each; none in the first stage, where k = 0), the R-point transform in registers, R outputs.  Per element one
// LDS read, one LDS write and at most one table read, where the form above has R of each.  sg = +1 / -1: the sign of the exponent.
template <typename T> RF_HD cplx<T> gmul(const cplx<T>& a, const cplx<T>& w, T sg) {
  const T wy = w.y * sg;
  return mk<T>(a.x * w.x - a.y * wy, a.x * wy + a.y * w.x);
}
template <typename T> RF_HD cplx<T> gadd(const cplx<T>& a, const cplx<T>& b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> RF_HD cplx<T> gsub(const cplx<T>& a, const cplx<T>& b) { return mk<T>(a.x - b.x, a.y - b.y); }
// a + sg * i * b   and   a - sg * i * b
template <typename T> RF_HD cplx<T> gadd_i(const cplx<T>& a, const cplx<T>& b, T sg) { return mk<T>(a.x - sg * b.y, a.y + sg * b.x); }
template <typename T> RF_HD cplx<T> gsub_i(const cplx<T>& a, const cplx<T>& b, T sg) { return mk<T>(a.x + sg * b.y, a.y - sg * b.x); }

template <typename T, int R> RF_HD void generic_dft(cplx<T>* y, T sg) {
  if (R == 2) {
    const cplx<T> a = y[0], b = y[1];
    y[0] = gadd(a, b); y[1] = gsub(a, b);
  } else if (R == 4) {
    const cplx<T> a = gadd(y[0], y[2]), b = gsub(y[0], y[2]), c = gadd(y[1], y[3]), d = gsub(y[1], y[3]);
    y[0] = gadd(a, c); y[2] = gsub(a, c);
    y[1] = gadd_i(b, d, sg); y[3] = gsub_i(b, d, sg);
  } else if (R == 3) {
    const T h = (T)0.86602540378443864676;        // sin(2 pi / 3)
    const cplx<T> t1 = gadd(y[1], y[2]), d = gsub(y[1], y[2]);
    const cplx<T> t2 = mk<T>(y[0].x - (T)0.5 * t1.x, y[0].y - (T)0.5 * t1.y), t3 = mk<T>(h * d.x, h * d.y);
    y[0] = gadd(y[0], t1);
    y[1] = gadd_i(t2, t3, sg); y[2] = gsub_i(t2, t3, sg);
  } else if (R == 8) {
    // even / odd halves through the 4-point transform, then X[k] = a[k] + w8^k b[k], X[k + 4] = a[k] - w8^k b[k], w8 = exp(sg 2 pi i / 8)
    const T h = (T)0.70710678118654752440;
    cplx<T> a[4] = {y[0], y[2], y[4], y[6]}, b[4] = {y[1], y[3], y[5], y[7]};
    generic_dft<T, 4>(a, sg);
    generic_dft<T, 4>(b, sg);
    const cplx<T> b1 = mk<T>(h * (b[1].x - sg * b[1].y), h * (b[1].y + sg * b[1].x));          // w8   b[1]
    const cplx<T> b2 = mk<T>(-sg * b[2].y, sg * b[2].x);                                          // w8^2 b[2] = sg i b[2]
    const cplx<T> b3 = mk<T>(h * (-b[3].x - sg * b[3].y), h * (-b[3].y + sg * b[3].x));        // w8^3 b[3]
    y[0] = gadd(a[0], b[0]); y[4] = gsub(a[0], b[0]);
    y[1] = gadd(a[1], b1);   y[5] = gsub(a[1], b1);
    y[2] = gadd(a[2], b2);   y[6] = gsub(a[2], b2);
    y[3] = gadd(a[3], b3);   y[7] = gsub(a[3], b3);
  } else {                                         // R == 5
    const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;       // cos(2 pi / 5), cos(4 pi / 5)
    const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;        // sin(2 pi / 5), sin(4 pi / 5)
    const cplx<T> a1 = gadd(y[1], y[4]), a2 = gadd(y[2], y[3]), b1 = gsub(y[1], y[4]), b2 = gsub(y[2], y[3]);
    const cplx<T> p1 = mk<T>(y[0].x + c1 * a1.x + c2 * a2.x, y[0].y + c1 * a1.y + c2 * a2.y);
    const cplx<T> p2 = mk<T>(y[0].x + c2 * a1.x + c1 * a2.x, y[0].y + c2 * a1.y + c1 * a2.y);
    const cplx<T> q1 = mk<T>(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const cplx<T> q2 = mk<T>(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    y[0] = mk<T>(y[0].x + a1.x + a2.x, y[0].y + a1.y + a2.y);
    y[1] = gadd_i(p1, q1, sg); y[4] = gsub_i(p1, q1, sg);
    y[2] = gadd_i(p2, q2, sg); y[3] = gsub_i(p2, q2, sg);
  }
}

template <typename T, int R>
RF_HD void generic_stage_r(const cplx<T>* in, cplx<T>* out, int n, int TC, int P, int Ns, unsigned Nsm, const cplx<T>* root, int rstep, int sign, int tid, int nth) {
  const FastDiv dNs((uint32_t)Ns, Nsm);
  const int m = n / R, unit = (int)dNs.div((uint32_t)m);
  const GenericWalk walk(m, TC, tid, nth);         // the m * TC butterflies: butterfly j of line c
  const T sg = sign < 0 ? (T)-1 : (T)1;
  const int mstep = m * P, ostep = Ns * P;
  int i = 0;
  for (int idx = tid; idx < walk.total; idx += nth, ++i) {
    int c, j;
    walk.at(idx, i, c, j);
    uint32_t jhi, k;
    dNs.divmod((uint32_t)j, jhi, k);
    cplx<T> y[R];
    const cplx<T>* pin = in + j * P + c;
#pragma unroll
    for (int r = 0; r < R; ++r) y[r] = pin[r * mstep];
    if (Ns > 1) {
      const int q = (int)k * unit * rstep;         // r * k * unit < n for r < R
#pragma unroll
      for (int r = 1; r < R; ++r) y[r] = gmul(y[r], root[r * q], sg);
    }
    generic_dft<T, R>(y, sg);
    cplx<T>* po = out + ((int)jhi * Ns * R + (int)k) * P + c;
#pragma unroll
    for (int u = 0; u < R; ++u) po[u * ostep] = y[u];
  }
}

// ---- axes whose radices are all among 2, 3, 4, 5, 8 ("smooth": every power-of-two times 3^a 5^b length): IN PLACE, one buffer.
// Decimation in time on a line stored in digit-reversed order: stage s (radix R, Ns = product of the radices before it) finds the R
// sub-transforms of length Ns it combines in the R consecutive runs of Ns positions of one block of Ns R, and leaves the block's
// transform in those same positions -- butterflies touch disjoint positions, so a stage needs no second buffer and the line needs
// half the LDS (twice the lines per workgroup, or two workgroups per CU).  The digit reversal costs nothing: it is the position an
// element is WRITTEN to when the line is loaded (generic_pos).  Same butterflies on the same values as the two-buffer form.
RF_HD bool generic_smooth(const GenericAxis& ax) { return ax.smooth != 0; }
// buffers of n * TC elements a line transform of this axis needs in LDS
RF_HD int generic_bufs(const GenericAxis& ax) { return generic_smooth(ax) ? 1 : 2; }

// where element e of a line goes when the line is loaded: e = d_(nf-1) + f_(nf-1) (d_(nf-2) + f_(nf-2) (...)), the digit of the LAST
// radix least significant, lands at sum_s d_s Ns_s; the identity for an axis that is not smooth (two-buffer form, natural order)
RF_HD int generic_pos(const GenericAxis& ax, int e) {
  if (!ax.smooth) return e;
  uint32_t j = (uint32_t)e;
  int p = 0;
  for (int s = ax.nf - 1; s >= 0; --s) {
    uint32_t q, r;
    FastDiv((uint32_t)ax.f[s], ax.fm[s]).divmod(j, q, r);
    p += (int)r * ax.w[s];
    j = q;
  }
  return p;
}

template <typename T, int R>
RF_HD void generic_stage_inplace(cplx<T>* a, int n, int TC, int P, int Ns, unsigned Nsm, const cplx<T>* root, int rstep, int sign, int tid, int nth) {
  const FastDiv dNs((uint32_t)Ns, Nsm);
  const int m = n / R, unit = (int)dNs.div((uint32_t)m);
  const GenericWalk walk(m, TC, tid, nth);         // the m * TC butterflies
  const T sg = sign < 0 ? (T)-1 : (T)1;
  const int step = Ns * P;
  int i = 0;
  for (int idx = tid; idx < walk.total; idx += nth, ++i) {
    int c, bf;
    walk.at(idx, i, c, bf);
    uint32_t jhi, k;
    dNs.divmod((uint32_t)bf, jhi, k);
    cplx<T>* p = a + ((int)jhi * Ns * R + (int)k) * P + c;
    cplx<T> y[R];
#pragma unroll
    for (int r = 0; r < R; ++r) y[r] = p[r * step];
    if (Ns > 1) {
      const int q = (int)k * unit * rstep;         // r * k * unit < n for r < R
#pragma unroll
      for (int r = 1; r < R; ++r) y[r] = gmul(y[r], root[r * q], sg);
    }
    generic_dft<T, R>(y, sg);
#pragma unroll
    for (int u = 0; u < R; ++u) p[u * step] = y[u];
  }
}

template <typename T>
RF_HD void generic_stage(const cplx<T>* in, cplx<T>* out, int n, int TC, int P, int R, int Ns, unsigned Nsm, const cplx<T>* root, int rstep,
                         int sign, int tid, int nth) {
  switch (R) {
    case 2: generic_stage_r<T, 2>(in, out, n, TC, P, Ns, Nsm, root, rstep, sign, tid, nth); break;
    case 3: generic_stage_r<T, 3>(in, out, n, TC, P, Ns, Nsm, root, rstep, sign, tid, nth); break;
    case 4: generic_stage_r<T, 4>(in, out, n, TC, P, Ns, Nsm, root, rstep, sign, tid, nth); break;
    case 5: generic_stage_r<T, 5>(in, out, n, TC, P, Ns, Nsm, root, rstep, sign, tid, nth); break;
    case 8: generic_stage_r<T, 8>(in, out, n, TC, P, Ns, Nsm, root, rstep, sign, tid, nth); break;
    default: generic_stage_any<T>(in, out, n, TC, P, R, Ns, root, rstep, sign, tid, nth);
  }
}

// all stages; returns the buffer that holds the result (a or b).  The caller has written element e of line c to
// a[generic_pos(ax, e) * P + c] -- P >= TC is the pitch of the LDS image (TC where the threads walk along c, TC + 1 where they walk along
// e: an odd pitch keeps those walks off the same banks); a smooth axis is transformed in place (b is not touched and may be null).
template <typename T, class Sync>
RF_HD cplx<T>* generic_line_fft(cplx<T>* a, cplx<T>* b, const GenericAxis& ax, int TC, int P, const cplx<T>* root, int rstep,
                                int sign, int tid, int nth, Sync sync) {
  int Ns = 1;
  if (generic_smooth(ax)) {
    for (int s = 0; s < ax.nf; ++s) {
      switch (ax.f[s]) {
        case 2: generic_stage_inplace<T, 2>(a, ax.n, TC, P, Ns, ax.wm[s], root, rstep, sign, tid, nth); break;
        case 3: generic_stage_inplace<T, 3>(a, ax.n, TC, P, Ns, ax.wm[s], root, rstep, sign, tid, nth); break;
        case 4: generic_stage_inplace<T, 4>(a, ax.n, TC, P, Ns, ax.wm[s], root, rstep, sign, tid, nth); break;
        case 8: generic_stage_inplace<T, 8>(a, ax.n, TC, P, Ns, ax.wm[s], root, rstep, sign, tid, nth); break;
        default: generic_stage_inplace<T, 5>(a, ax.n, TC, P, Ns, ax.wm[s], root, rstep, sign, tid, nth);
      }
      sync();
      Ns *= ax.f[s];
    }
    return a;
  }
  for (int s = 0; s < ax.nf; ++s) {
    generic_stage<T>(a, b, ax.n, TC, P, ax.f[s], Ns, ax.wm[s], root, rstep, sign, tid, nth);
    sync();
    Ns *= ax.f[s];
    cplx<T>* t = a; a = b; b = t;
  }
  return a;
}

// The stages' twiddles from LDS instead of the root table in global memory: tw[t] = root[t * rstep], t < n, staged behind the line
// image(s) by the block functions when the launcher found room for it (`tw_lds`).  A butterfly waits for its R - 1 twiddles before
// it can start, and with two butterflies per thread and stage there is nothing else to overlap a cache hit's latency with.
template <typename T>
RF_HD const cplx<T>* generic_stage_table(int tw_lds, cplx<T>* lds_tw, const cplx<T>* root, int n, int& rstep, int tid, int nth) {
  if (!tw_lds) return root;
  for (int t = tid; t < n; t += nth) lds_tw[t] = root[t * rstep];
  rstep = 1;
  return lds_tw;                                   // (the caller's barrier before the first stage covers these writes)
}
// ... and, behind the stage table, generic_pos as a table of n 16-bit entries (smooth axes; `tw_lds` switches both on): the digit
// reversal is ~5 instructions per radix and element when computed, one LDS read when looked up.  The caller puts a barrier between
// this and the first use.  nullptr: compute.
template <typename T>
RF_HD const uint16_t* generic_pos_table(int tw_lds, cplx<T>* behind_stage_table, const GenericAxis& ax, int tid, int nth) {
  if (!tw_lds || !ax.smooth) return nullptr;
  uint16_t* pt = reinterpret_cast<uint16_t*>(behind_stage_table);
  for (int t = tid; t < ax.n; t += nth) pt[t] = (uint16_t)generic_pos(ax, t);
  return pt;
}
// bytes of LDS behind the line image(s) when tw_lds is on: the stage table and the position table
inline size_t generic_extra_bytes(const GenericAxis& ax, int elem_bytes) {
  return (size_t)ax.n * elem_bytes + (ax.smooth ? (((size_t)ax.n * 2 + 15) & ~(size_t)15) : 0);
}

// Strided (or contiguous) complex pass: block `blk` transforms lines [blk TC, blk TC + TC) of length ax.n;
// line l starts at (l / inner) * outer + l % inner and its elements are `stride` apart.  src == dst is allowed
// (a block reads all of its lines before it writes any).  lds: generic_bufs(ax) * ax.n * TC elements.
template <typename T, class Sync>
RF_HD void generic_axis_block(const cplx<T>* src, cplx<T>* dst, const GenericAxis& ax, long long stride, long long inner,
                              long long outer, long long nlines, int TC, const cplx<T>* root, int sign, T scale,
                              cplx<T>* lds, long long blk, int tid, int nth, Sync sync, int tw_lds = 0) {
  const int n = ax.n, total = n * TC;
  const long long l0 = blk * TC;
  cplx<T>*a = lds, *b = lds + total;
  const GenericWalk walk(n, TC, tid, nth);
  int rs = 1;
  const cplx<T>* rt = generic_stage_table<T>(tw_lds, lds + generic_bufs(ax) * total, root, n, rs, tid, nth);
  const uint16_t* ptab = generic_pos_table<T>(tw_lds, lds + generic_bufs(ax) * total + n, ax, tid, nth);
  if (ptab) sync();
  // (a thread that stays on one line forms that line's base once: the 64-bit division is ~100 instructions)
  const long long lf = l0 + walk.c0;
  const long long basef = walk.fixed && lf < nlines ? (lf / inner) * outer + lf % inner : 0;
  int i = 0;
  if (walk.fixed) {
    // four loads in flight per thread before the first is used (the loop below, left to the compiler, waits for each load in turn:
    // the position of the LDS write is a run-time loop over the radices)
    const bool live = lf < nlines;
    for (int e = walk.e0; e < n; e += 4 * walk.estep) {
      cplx<T> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ee = e + u * walk.estep;
        v[u] = mk<T>((T)0, (T)0);
        if (live && ee < n) v[u] = src[basef + ee * stride];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ee = e + u * walk.estep;
        if (ee < n) a[(ptab ? (int)ptab[ee] : generic_pos(ax, ee)) * TC + walk.c0] = v[u];
      }
    }
  } else {
    for (int idx = tid; idx < total; idx += nth, ++i) {
      int c, e;
      walk.at(idx, i, c, e);
      const long long l = l0 + c;
      cplx<T> v = mk<T>((T)0, (T)0);
      if (l < nlines) v = src[(l / inner) * outer + l % inner + e * stride];
      a[(ptab ? (int)ptab[e] : generic_pos(ax, e)) * TC + c] = v;
    }
  }
  sync();
  const cplx<T>* r = generic_line_fft<T>(a, b, ax, TC, TC, rt, rs, sign, tid, nth, sync);
  if (walk.fixed) {
    if (lf < nlines) {
      cplx<T>* pd = dst + basef + walk.e0 * stride;
      const long long step = walk.estep * stride;
      const cplx<T>* pr = r + walk.e0 * TC + walk.c0;
      const int rstepl = walk.estep * TC;
      for (int e = walk.e0; e < n; e += walk.estep, pd += step, pr += rstepl) {
        const cplx<T> v = *pr;
        *pd = mk<T>(v.x * scale, v.y * scale);
      }
    }
  } else {
    i = 0;
    for (int idx = tid; idx < total; idx += nth, ++i) {
      int c, e;
      walk.at(idx, i, c, e);
      const long long l = l0 + c;
      if (l < nlines) {
        const cplx<T> v = r[idx];
        dst[(l / inner) * outer + l % inner + e * stride] = mk<T>(v.x * scale, v.y * scale);
      }
    }
  }
}

// Pitch of the LDS image of the contiguous passes (element e of row c at [e * pitch + c]).  Their threads walk ALONG a row on load and
// store (that is what keeps the global accesses whole lines), i.e. with a stride of `pitch` elements in LDS: at pitch TR = 8 that is
// 64 bytes and eight lanes share a bank (three quarters of the pass's LDS cycles were conflict cycles: SQ_LDS_BANK_CONFLICT 9.9e8 of
// SQ_LDS_IDX_ACTIVE 1.3e9 at 1000^3); the odd pitch TR + 1 spreads both that walk and the stages' walk along c over all banks.
RF_HD int generic_row_pitch(int TR) { return TR > 1 ? TR + 1 : TR; }

// Contiguous pass of the packed inverse transform: rows of M + 1 = nz/2 + 1 half-spectrum bins -> nz reals.
// With w = exp(2 pi i / nz):  z[m] = x[2m] + i x[2m+1] = IDFT_M( (X[k] + conj X[M-k]) + i w^k (X[k] - conj X[M-k]) ),
// the imaginary parts of X[0] and X[M] being ignored.  `ax` factors M; root = exp(2 pi i t / nz), t in [0, nz).
// Block `blk` owns rows [blk TR, blk TR + TR); the calling thread's share of (sum, sum of squares) of the outputs
// (after `scale`) is added to s1, s2.  lds: generic_bufs(ax) * M * generic_row_pitch(TR) elements.
template <typename T, class Sync>
RF_HD void generic_row_c2r_block(const cplx<T>* G, T* W, const GenericAxis& ax, long long nrows, int TR, const cplx<T>* root,
                                 T scale, cplx<T>* lds, long long blk, int tid, int nth, Sync sync, double& s1, double& s2, int tw_lds = 0) {
  const int M = ax.n, total = M * TR, P = generic_row_pitch(TR);
  const long long r0 = blk * TR;
  cplx<T>*a = lds, *b = lds + M * P;
  int rs = 2;
  const cplx<T>* rt = generic_stage_table<T>(tw_lds, lds + generic_bufs(ax) * M * P, root, M, rs, tid, nth);
  const uint16_t* ptab = generic_pos_table<T>(tw_lds, lds + generic_bufs(ax) * M * P + M, ax, tid, nth);
  if (ptab) sync();
  const FastDiv dM((uint32_t)M);
  // four elements per trip, their loads issued before the first is used (see generic_axis_block)
  for (int idx0 = tid; idx0 < total; idx0 += 4 * nth) {
    cplx<T> pv[4], qv[4], wv[4];
    int kk[4], cc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + u * nth;
      uint32_t cq = 0, kr = 0;
      if (idx < total) dM.divmod((uint32_t)idx, cq, kr);
      kk[u] = (int)kr; cc[u] = (int)cq;             // consecutive threads walk along a row
      pv[u] = qv[u] = wv[u] = mk<T>((T)0, (T)0);
      if (idx < total && r0 + cc[u] < nrows) {
        const cplx<T>* X = G + (r0 + cc[u]) * (long long)(M + 1);
        pv[u] = X[kk[u]]; qv[u] = X[M - kk[u]];     // conj X[M-k] = (q.x, -q.y); k = 0 reads X[0] and X[M]
        wv[u] = root[kk[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (idx0 + u * nth >= total) continue;
      const cplx<T> p = pv[u], q = qv[u], w = wv[u];
      cplx<T> z;
      if (kk[u] == 0) {
        z = mk<T>(p.x + q.x, p.x - q.x);
      } else {
        const T er = p.x + q.x, ei = p.y - q.y, orr = p.x - q.x, oi = p.y + q.y;
        // e + i w o
        z = mk<T>(er - (w.x * oi + w.y * orr), ei + (w.x * orr - w.y * oi));
      }
      a[(ptab ? (int)ptab[kk[u]] : generic_pos(ax, kk[u])) * P + cc[u]] = z;
    }
  }
  sync();
  const cplx<T>* r = generic_line_fft<T>(a, b, ax, TR, P, rt, rs, +1, tid, nth, sync);
  for (int idx = tid; idx < total; idx += nth) {
    uint32_t cq, mr;
    dM.divmod((uint32_t)idx, cq, mr);
    const int m = (int)mr, c = (int)cq;
    if (r0 + c < nrows) {
      const cplx<T> v = r[m * P + c];
      const T x0 = v.x * scale, x1 = v.y * scale;
      T* out = W + (r0 + c) * (long long)(2 * M) + 2 * m;
      out[0] = x0;
      out[1] = x1;
      s1 += (double)x0 + (double)x1;
      s2 += (double)x0 * (double)x0 + (double)x1 * (double)x1;
    }
  }
}

// Contiguous pass of the packed forward transform: rows of nz reals -> M + 1 half-spectrum bins (np.fft.rfft):
//   Z = DFT_M(x[2m] + i x[2m+1]);  X[k] = (Z[k] + conj Z[M-k]) / 2 - (i / 2) conj(w)^k (Z[k] - conj Z[M-k]),  Z[M] = Z[0].
template <typename T, class Sync>
RF_HD void generic_row_r2c_block(const T* W, cplx<T>* G, const GenericAxis& ax, long long nrows, int TR, const cplx<T>* root,
                                 cplx<T>* lds, long long blk, int tid, int nth, Sync sync, int tw_lds = 0) {
  const int M = ax.n, total = M * TR, P = generic_row_pitch(TR);
  const long long r0 = blk * TR;
  cplx<T>*a = lds, *b = lds + M * P;
  int rs = 2;
  const cplx<T>* rt = generic_stage_table<T>(tw_lds, lds + generic_bufs(ax) * M * P, root, M, rs, tid, nth);
  const uint16_t* ptab = generic_pos_table<T>(tw_lds, lds + generic_bufs(ax) * M * P + M, ax, tid, nth);
  if (ptab) sync();
  const FastDiv dM((uint32_t)M), dM1((uint32_t)(M + 1));
  for (int idx = tid; idx < total; idx += nth) {
    uint32_t cq, mr;
    dM.divmod((uint32_t)idx, cq, mr);
    const int m = (int)mr, c = (int)cq;
    cplx<T> z = mk<T>((T)0, (T)0);
    if (r0 + c < nrows) {
      const T* in = W + (r0 + c) * (long long)(2 * M) + 2 * m;
      z = mk<T>(in[0], in[1]);
    }
    a[(ptab ? (int)ptab[m] : generic_pos(ax, m)) * P + c] = z;
  }
  sync();
  const cplx<T>* r = generic_line_fft<T>(a, b, ax, TR, P, rt, rs, -1, tid, nth, sync);
  const int totalo = (M + 1) * TR;
  for (int idx = tid; idx < totalo; idx += nth) {
    uint32_t cq, kr;
    dM1.divmod((uint32_t)idx, cq, kr);
    const int k = (int)kr, c = (int)cq;
    if (r0 + c < nrows) {
      const cplx<T> p = r[(k == M ? 0 : k) * P + c], q = r[(k == 0 ? 0 : M - k) * P + c];
      const T er = (T)0.5 * (p.x + q.x), ei = (T)0.5 * (p.y - q.y), orr = (T)0.5 * (p.x - q.x), oi = (T)0.5 * (p.y + q.y);
      cplx<T> w = root[k];
      w.y = -w.y;
      // e - i w o
      G[(r0 + c) * (long long)(M + 1) + k] = mk<T>(er + (w.x * oi + w.y * orr), ei - (w.x * orr - w.y * oi));
    }
  }
}

// ---------------------------------------------------------------------------
// Axes LONGER than a line that fits the LDS (n > generic_max_axis): the four-step form through global memory.  With n = n1 n2,
// j = j1 n2 + j2 and k = k1 + n1 k2:
//     X[k1 + n1 k2] = sum_j2 w_n2^(j2 k2) * [ w_n^(j2 k1) * sum_j1 x[j1 n2 + j2] w_n1^(j1 k1) ]
//   step 1: for every j2 the length-n1 transform over j1 (elements n2 apart), in place: Y[k1][j2] at position k1 n2 + j2;
//   step 3: for every k1 the length-n2 transform over j2 (adjacent elements), its inputs multiplied by w_n^(j2 k1) on load, its outputs
//           stored n1 apart from position k1: natural order -- into ANOTHER array (a line's stores land among other lines' inputs).
// Both steps are "lines with sub-lines" (GenericLines) run by generic_lines_block, the generalisation of generic_axis_block: parent
// line l0 starts at (l0 / inner) outer + l0 % inner; its sub-line q starts q * sub further on and has its own element stride.
// Every even n <= cap^2 that splits into two factors <= cap is served (transform.py:172-177 accepts any even shape).
// ---------------------------------------------------------------------------
struct GenericLong {
  int n = 0, n1 = 0, n2 = 0;            // n1 = 0: the axis is not split (its line fits the LDS)
  GenericAxis a1, a2;
  RF_HD bool split() const { return n1 > 0; }
};
// n = n1 n2 with both factors <= cap (and factorable): the most balanced split; false if there is none
inline bool generic_split(long long n, int cap, GenericLong& lg) {
  lg = GenericLong();
  if (n < 4 || n > (long long)cap * cap || n > 0x7fffffffLL) return false;
  lg.n = (int)n;
  int best = 0;
  for (long long d = 1; d * d <= n; ++d)
    if (n % d == 0 && n / d <= cap && d <= cap) best = (int)d;           // largest divisor <= sqrt(n) whose cofactor fits
  if (best < 2) return false;
  lg.n1 = (int)(n / best); lg.n2 = best;                                  // n1 >= n2
  return generic_factor(lg.n1, lg.a1) && generic_factor(lg.n2, lg.a2);
}

struct GenericLines {
  GenericAxis ax;                       // the line's own transform
  int rstep = 1;                        // root[] holds exp(2 pi i t / (ax.n * rstep))
  long long stride_s = 1, inner_s = 1, outer_s = 0, sub_s = 0;      // source: parent line, element stride, sub-line offset
  long long stride_d = 1, inner_d = 1, outer_d = 0, sub_d = 0;      // destination
  long long nparent = 0;                // parent lines
  int nsub = 1;                         // sub-lines per parent line; line index l = q * nparent + l0 (neighbouring l0 are neighbours in memory)
  int tw_n = 0, tw_step = 1;            // > 0: element e of sub-line q is multiplied by root[((e q) mod tw_n) * tw_step] on load (conjugated for sign < 0)
  int sign = +1;
  double scale = 1.0;
  RF_HD long long nlines() const { return nparent * nsub; }
};
// step 1 / step 3 of the four-step transform of the lines (S, inner, outer, nparent) of length lg.n; root_mul: the root table holds
// exp(2 pi i t / (lg.n * root_mul)).  Step 3 stores through (Sd, inner_d, outer_d).
inline GenericLines generic_long_step1(const GenericLong& lg, long long S, long long inner, long long outer, long long nparent, int root_mul, int sign) {
  GenericLines L;
  L.ax = lg.a1; L.rstep = lg.n2 * root_mul;
  L.stride_s = L.stride_d = (long long)lg.n2 * S; L.inner_s = L.inner_d = inner; L.outer_s = L.outer_d = outer; L.sub_s = L.sub_d = S;
  L.nparent = nparent; L.nsub = lg.n2; L.sign = sign;
  return L;
}
inline GenericLines generic_long_step3(const GenericLong& lg, long long S, long long inner, long long outer, long long Sd, long long inner_d,
                                       long long outer_d, long long nparent, int root_mul, int sign, double scale) {
  GenericLines L;
  L.ax = lg.a2; L.rstep = lg.n1 * root_mul;
  L.stride_s = S; L.inner_s = inner; L.outer_s = outer; L.sub_s = (long long)lg.n2 * S;
  L.stride_d = (long long)lg.n1 * Sd; L.inner_d = inner_d; L.outer_d = outer_d; L.sub_d = Sd;
  L.nparent = nparent; L.nsub = lg.n1; L.tw_n = lg.n; L.tw_step = root_mul; L.sign = sign; L.scale = scale;
  return L;
}

// block `blk` transforms lines [blk TC, blk TC + TC) of L; lds: generic_bufs(L.ax) * L.ax.n * TC elements.  src == dst is allowed when the two
// addressings are the same (step 1); step 3 needs another array.
template <typename T, class Sync>
RF_HD void generic_lines_block(const cplx<T>* src, cplx<T>* dst, const GenericLines& L, int TC, const cplx<T>* root, cplx<T>* lds,
                               long long blk, int tid, int nth, Sync sync, int tw_lds = 0) {
  const int n = L.ax.n, total = n * TC;
  const long long l0b = blk * TC, nl = L.nlines();
  cplx<T>*a = lds, *b = lds + total;
  const GenericWalk walk(n, TC, tid, nth);
  int rs = L.rstep;
  const cplx<T>* rt = generic_stage_table<T>(tw_lds, lds + generic_bufs(L.ax) * total, root, n, rs, tid, nth);
  const uint16_t* ptab = generic_pos_table<T>(tw_lds, lds + generic_bufs(L.ax) * total + n, L.ax, tid, nth);
  if (ptab) sync();
  // (a thread that stays on one line forms that line's sub-line index and bases once)
  long long qf = 0, bsf = 0, bdf = 0;
  if (walk.fixed && l0b + walk.c0 < nl) {
    const long long l = l0b + walk.c0;
    qf = l / L.nparent;
    const long long l0 = l - qf * L.nparent;
    bsf = (l0 / L.inner_s) * L.outer_s + l0 % L.inner_s + qf * L.sub_s;
    bdf = (l0 / L.inner_d) * L.outer_d + l0 % L.inner_d + qf * L.sub_d;
  }
  int i = 0;
  if (walk.fixed) {
    // four loads (and their twiddles) in flight per thread; the twiddle index (e q) mod tw_n advances by (estep q) mod tw_n per
    // element -- one 64-bit modulo per thread instead of one per element
    const bool live = l0b + walk.c0 < nl;
    const bool twid = L.tw_n > 0;
    long long t = twid ? ((long long)walk.e0 * qf) % L.tw_n : 0;
    const long long dt = twid ? ((long long)walk.estep * qf) % L.tw_n : 0;
    for (int e = walk.e0; e < n; e += 4 * walk.estep) {
      cplx<T> v[4], w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ee = e + u * walk.estep;
        v[u] = mk<T>((T)0, (T)0);
        w[u] = mk<T>((T)1, (T)0);
        if (live && ee < n) {
          v[u] = src[bsf + ee * L.stride_s];
          if (twid) w[u] = root[t * L.tw_step];
        }
        t += dt;
        if (t >= L.tw_n) t -= L.tw_n;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ee = e + u * walk.estep;
        if (ee >= n) continue;
        cplx<T> x = v[u];
        if (twid) {
          cplx<T> ww = w[u];
          if (L.sign < 0) ww.y = -ww.y;
          x = mk<T>(x.x * ww.x - x.y * ww.y, x.x * ww.y + x.y * ww.x);
        }
        a[(ptab ? (int)ptab[ee] : generic_pos(L.ax, ee)) * TC + walk.c0] = x;
      }
    }
  } else {
    for (int idx = tid; idx < total; idx += nth, ++i) {
      int c, e;
      walk.at(idx, i, c, e);
      const long long l = l0b + c;
      cplx<T> v = mk<T>((T)0, (T)0);
      if (l < nl) {
        const long long q = l / L.nparent, l0 = l - q * L.nparent;
        const long long bs = (l0 / L.inner_s) * L.outer_s + l0 % L.inner_s + q * L.sub_s;
        v = src[bs + e * L.stride_s];
        if (L.tw_n > 0) {
          const long long t = ((long long)e * q) % L.tw_n;
          cplx<T> w = root[t * L.tw_step];
          if (L.sign < 0) w.y = -w.y;
          v = mk<T>(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
        }
      }
      a[(ptab ? (int)ptab[e] : generic_pos(L.ax, e)) * TC + c] = v;
    }
  }
  sync();
  const cplx<T>* r = generic_line_fft<T>(a, b, L.ax, TC, TC, rt, rs, L.sign, tid, nth, sync);
  const T scale = (T)L.scale;
  i = 0;
  for (int idx = tid; idx < total; idx += nth, ++i) {
    int c, e;
    walk.at(idx, i, c, e);
    const long long l = l0b + c;
    if (l < nl) {
      long long bd = bdf;
      if (!walk.fixed) {
        const long long q = l / L.nparent, l0 = l - q * L.nparent;
        bd = (l0 / L.inner_d) * L.outer_d + l0 % L.inner_d + q * L.sub_d;
      }
      const cplx<T> v = r[idx];
      dst[bd + e * L.stride_d] = mk<T>(v.x * scale, v.y * scale);
    }
  }
}

// The Hermitian (un)tangle of the packed transforms as passes of their own, for rows too long for generic_row_c2r_block /
// generic_row_r2c_block (element idx of the grid-stride loop; root = exp(2 pi i t / nz), M = nz / 2):
//   untangle: Z[row][k] = (X[k] + conj X[M-k]) + i w^k (X[k] - conj X[M-k]), k < M, from rows of M + 1 bins (imaginary parts of X[0], X[M] ignored)
template <typename T>
RF_HD void generic_untangle_at(const cplx<T>* G, cplx<T>* Z, int M, const cplx<T>* root, long long idx) {
  const long long row = idx / M;
  const int k = (int)(idx - row * M);
  const cplx<T>* X = G + row * (long long)(M + 1);
  cplx<T> z;
  if (k == 0) {
    z = mk<T>(X[0].x + X[M].x, X[0].x - X[M].x);
  } else {
    const cplx<T> p = X[k], q = X[M - k];
    const T er = p.x + q.x, ei = p.y - q.y, orr = p.x - q.x, oi = p.y + q.y;
    const cplx<T> w = root[k];
    z = mk<T>(er - (w.x * oi + w.y * orr), ei + (w.x * orr - w.y * oi));
  }
  Z[idx] = z;
}
//   tangle: X[k] = (Z[k] + conj Z[M-k]) / 2 - (i / 2) conj(w)^k (Z[k] - conj Z[M-k]), k <= M, Z[M] = Z[0], into rows of M + 1 bins
template <typename T>
RF_HD void generic_tangle_at(const cplx<T>* Z, cplx<T>* G, int M, const cplx<T>* root, long long idx) {
  const long long row = idx / (M + 1);
  const int k = (int)(idx - row * (M + 1));
  const cplx<T>* z = Z + row * (long long)M;
  const cplx<T> p = z[k % M], q = z[(M - k) % M];
  const T er = (T)0.5 * (p.x + q.x), ei = (T)0.5 * (p.y - q.y), orr = (T)0.5 * (p.x - q.x), oi = (T)0.5 * (p.y + q.y);
  cplx<T> w = root[k];
  w.y = -w.y;
  G[idx] = mk<T>(er + (w.x * oi + w.y * orr), ei - (w.x * orr - w.y * oi));
}

// ---------------------------------------------------------------------------
// The sequences of the three generic transforms, written ONCE for the library (Ops = kernel launches, rf_capi.hip) and for the CPU
// emulator (Ops = loops over blocks, emu/rf_emu.cpp), so that the buffer choreography of the long axes is tested on the CPU too.
// Ops provides (all return 0 or an error code; `which` = 0 / 1 / 2 selects the x / y / z root table):
//   axis(src, dst, ax, stride, inner, outer, nlines, which, sign, scale)   one plain pass (generic_axis_block)
//   lines(src, dst, L, which)                                              one pass of lines with sub-lines (generic_lines_block)
//   row_c2r(G, W, scale) / row_r2c(W, G)                                   the fused contiguous passes of rows that fit the LDS
//   untangle(G, Z) / tangle(Z, G) / moments(W)                             the pieces of the contiguous passes for long rows
//   copy(dst, src, bytes)
// ---------------------------------------------------------------------------
// Lines of a strided pass that go into one workgroup together (neighbours in memory: tc lines = segments of tc elements): the widest
// tile up to 16 whose image(s) and stage table fit a CU's LDS (rf_k_generic.hip strided_shape launches exactly this).  Below 4 lines
// the segments are 16 bytes of complex64 and a pass moves a fraction of what the memory system can: generic_prefers_split.
inline int generic_strided_tile(const GenericAxis& ax, int elem_bytes) {
  auto lds = [&](int tc) { return (long long)generic_bufs(ax) * ax.n * tc * elem_bytes + (long long)generic_extra_bytes(ax, elem_bytes); };
  int tc = 16;
  while (tc > 1 && lds(tc) > (long long)GENERIC_LDS_MAX) tc >>= 1;
  return tc;
}
// A strided axis that fits one line of the LDS but only one or two lines per workgroup is faster in the four-step form, whose factors
// go 16 lines to a workgroup (measured, 1000-cell transverse planes: 4096 points 0.49 -> 0.40 ms, 8192 points 0.36 -> 0.27 ms per
// realisation; 2000 points -- 4 lines -- 1.25 against 1.54: stays one pass).  Contiguous axes never: their lines ARE the segments.
inline bool generic_prefers_split(const GenericAxis& ax, int elem_bytes) { return generic_strided_tile(ax, elem_bytes) < 4; }

struct GenericDims {
  int nx = 0, ny = 0, nz = 0;
  GenericAxis ax, ay, az;               // az factors nz / 2 (packed plans) or nz (c2c plans); unused where the long form applies
  GenericLong lx, ly, lz;
  size_t csize = 8;                     // bytes per complex element
};
// four-step pass of the lines (S, inner, outer, nparent) of a long axis: step 1 src -> tmp (tmp may be src), step 3 tmp -> dst (dst != tmp)
template <class Ops>
int generic_long_pass(Ops& ops, const void* src, void* tmp, void* dst, const GenericLong& lg, long long S, long long inner, long long outer,
                      long long Sd, long long inner_d, long long outer_d, long long nparent, int which, int root_mul, int sign, double scale) {
  if (int rc = ops.lines(src, tmp, generic_long_step1(lg, S, inner, outer, nparent, root_mul, sign), which)) return rc;
  return ops.lines(tmp, dst, generic_long_step3(lg, S, inner, outer, Sd, inner_d, outer_d, nparent, root_mul, sign, scale), which);
}
// half spectrum K [nx][ny][nz/2+1] -> dense reals W [nx][ny][nz] (np.fft.irfftn with `scale`); G, G2: scratch arrays of K's size (G2 is
// touched only when an axis is long); the (sum, sumsq) partials are left by row_c2r / moments
template <class Ops>
int generic_c2r_seq(Ops& ops, const GenericDims& d, const void* K, void* G, void* G2, void* W, double scale) {
  const long long nzh = d.nz / 2 + 1, M = d.nz / 2;
  const long long Lx = (long long)d.ny * nzh, Ly = (long long)d.nx * nzh, rows = (long long)d.nx * d.ny;
  if (d.lx.split()) { if (int rc = generic_long_pass(ops, K, G2, G, d.lx, Lx, Lx, 0, Lx, Lx, 0, Lx, 0, 1, +1, 1.0)) return rc; }
  else if (int rc = ops.axis(K, G, d.ax, Lx, Lx, 0, Lx, 0, +1, 1.0)) return rc;
  void* cur = G;
  if (d.ly.split()) { if (int rc = generic_long_pass(ops, G, G, G2, d.ly, nzh, nzh, (long long)d.ny * nzh, nzh, nzh, (long long)d.ny * nzh, Ly, 1, 1, +1, 1.0)) return rc; cur = G2; }
  else if (int rc = ops.axis(G, G, d.ay, nzh, nzh, (long long)d.ny * nzh, Ly, 1, +1, 1.0)) return rc;
  if (!d.lz.split()) return ops.row_c2r(cur, W, scale);
  void* other = cur == G ? G2 : G;                       // rows of M complex: the untangled spectrum, then (step 1, in place) its first transform
  if (int rc = ops.untangle(cur, other)) return rc;
  if (int rc = generic_long_pass(ops, other, other, W, d.lz, 1, 1, M, 1, 1, M, rows, 2, 2, +1, scale)) return rc;     // W as rows of M complex = nz reals
  return ops.moments(W);
}
// dense reals W -> half spectrum K (np.fft.rfftn); W is left untouched
template <class Ops>
int generic_r2c_seq(Ops& ops, const GenericDims& d, const void* W, void* K, void* G, void* G2) {
  const long long nzh = d.nz / 2 + 1, M = d.nz / 2;
  const long long Lx = (long long)d.ny * nzh, Ly = (long long)d.nx * nzh, rows = (long long)d.nx * d.ny;
  if (d.lz.split()) {
    if (int rc = generic_long_pass(ops, W, G, G2, d.lz, 1, 1, M, 1, 1, M, rows, 2, 2, -1, 1.0)) return rc;
    if (int rc = ops.tangle(G2, K)) return rc;
  } else if (int rc = ops.row_r2c(W, K)) return rc;
  void* cur = K;
  if (d.ly.split()) { if (int rc = generic_long_pass(ops, K, K, G, d.ly, nzh, nzh, (long long)d.ny * nzh, nzh, nzh, (long long)d.ny * nzh, Ly, 1, 1, -1, 1.0)) return rc; cur = G; }
  else if (int rc = ops.axis(K, K, d.ay, nzh, nzh, (long long)d.ny * nzh, Ly, 1, -1, 1.0)) return rc;
  if (d.lx.split()) {
    void* dst = cur == K ? G : K;
    if (int rc = generic_long_pass(ops, cur, cur, dst, d.lx, Lx, Lx, 0, Lx, Lx, 0, Lx, 0, 1, -1, 1.0)) return rc;
    cur = dst;
  } else if (int rc = ops.axis(cur, cur, d.ax, Lx, Lx, 0, Lx, 0, -1, 1.0)) return rc;
  if (cur != K) return ops.copy(K, cur, (size_t)d.nx * d.ny * nzh * d.csize);
  return 0;
}
// unpacked complex array D [nx][ny][nz] in place (np.fft.fftn / ifftn: sign -1 unscaled, +1 with `scale`); G: scratch of D's size,
// touched only when an axis is long
template <class Ops>
int generic_c2c_seq(Ops& ops, const GenericDims& d, void* D, void* G, int sign, double scale) {
  const long long nz = d.nz, Lx = (long long)d.ny * nz, Ly = (long long)d.nx * nz, rows = (long long)d.nx * d.ny;
  const size_t bytes = (size_t)d.nx * d.ny * nz * d.csize;
  if (d.lx.split()) { if (int rc = generic_long_pass(ops, D, D, G, d.lx, Lx, Lx, 0, Lx, Lx, 0, Lx, 0, 1, sign, 1.0)) return rc; if (int rc = ops.copy(D, G, bytes)) return rc; }
  else if (int rc = ops.axis(D, D, d.ax, Lx, Lx, 0, Lx, 0, sign, 1.0)) return rc;
  if (d.ly.split()) { if (int rc = generic_long_pass(ops, D, D, G, d.ly, nz, nz, (long long)d.ny * nz, nz, nz, (long long)d.ny * nz, Ly, 1, 1, sign, 1.0)) return rc; if (int rc = ops.copy(D, G, bytes)) return rc; }
  else if (int rc = ops.axis(D, D, d.ay, nz, nz, (long long)d.ny * nz, Ly, 1, sign, 1.0)) return rc;
  if (d.lz.split()) { if (int rc = generic_long_pass(ops, D, D, G, d.lz, 1, 1, nz, 1, 1, nz, rows, 2, 1, sign, scale)) return rc; return ops.copy(D, G, bytes); }
  return ops.axis(D, D, d.az, 1, 1, nz, rows, 2, sign, scale);
}

// lines / rows per block so that the two LDS buffers stay within 64 KB (no function attribute needed); a single line longer than
// that (n > 4096 complex64 / 2048 complex128) takes what it needs, up to GENERIC_LDS_MAX
inline int generic_lines_per_block(int n, int elem_bytes, int want, long long budget = 65536, int bufs = 2, bool row_pitch = false) {
  int tc = want;
  while (tc > 1 && ((long long)bufs * n * (row_pitch ? generic_row_pitch(tc) : tc) + n) * elem_bytes + 2LL * n + 16 > budget) tc >>= 1;     // (+ the stage and position tables)
  return tc;
}

}  // namespace rf
