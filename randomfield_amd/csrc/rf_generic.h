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
};

// radices of n: 4s first (fewest stages), then 2, then the odd primes in increasing order
inline bool generic_factor(int n, GenericAxis& ax) {
  ax.n = n;
  ax.nf = 0;
  if (n < 1 || n > GENERIC_MAX_N) return false;
  int m = n;
  auto push = [&](int r) { if (ax.nf < GENERIC_MAX_FACTORS) ax.f[ax.nf] = r; ++ax.nf; m /= r; };
  while (m % 4 == 0) push(4);
  while (m % 2 == 0) push(2);
  for (int p = 3; p * p <= m; p += 2)
    while (m % p == 0) push(p);
  if (m > 1) push(m);
  return ax.nf <= GENERIC_MAX_FACTORS;
}

// One Stockham stage of radix R on TC interleaved lines (element e of line c at [e * TC + c]): output o of a line is
//   out[o] = sum_r in[j + r n/R] * exp(sign 2 pi i r (k / (Ns R) + u / R)),   o = jhi Ns R + u Ns + k,  j = jhi Ns + k
// (Ns = product of the radices already applied).  `root` holds exp(+2 pi i t / (n * rstep)), t in [0, n * rstep).
template <typename T>
RF_HD void generic_stage(const cplx<T>* in, cplx<T>* out, int n, int TC, int R, int Ns, const cplx<T>* root, int rstep,
                         int sign, int tid, int nth) {
  const int m = n / R, unit = n / (Ns * R), total = n * TC;
  for (int idx = tid; idx < total; idx += nth) {
    const int c = idx % TC, o = idx / TC;
    const int k = o % Ns, u = (o / Ns) % R, jhi = o / (Ns * R);
    const int j = jhi * Ns + k;
    int q = k * unit + u * m;                      // < n / R + n
    if (q >= n) q -= n;
    T sr = (T)0, si = (T)0;
    int ri = 0;
    for (int r = 0; r < R; ++r) {
      const cplx<T> v = in[(j + r * m) * TC + c];
      cplx<T> w = root[ri * rstep];
      if (sign < 0) w.y = -w.y;
      sr += v.x * w.x - v.y * w.y;
      si += v.x * w.y + v.y * w.x;
      ri += q;
      if (ri >= n) ri -= n;
    }
    out[o * TC + c] = mk<T>(sr, si);
  }
}

// all stages; returns the buffer that holds the result (a or b)
template <typename T, class Sync>
RF_HD cplx<T>* generic_line_fft(cplx<T>* a, cplx<T>* b, const GenericAxis& ax, int TC, const cplx<T>* root, int rstep,
                                int sign, int tid, int nth, Sync sync) {
  int Ns = 1;
  for (int s = 0; s < ax.nf; ++s) {
    generic_stage<T>(a, b, ax.n, TC, ax.f[s], Ns, root, rstep, sign, tid, nth);
    sync();
    Ns *= ax.f[s];
    cplx<T>* t = a; a = b; b = t;
  }
  return a;
}

// Strided (or contiguous) complex pass: block `blk` transforms lines [blk TC, blk TC + TC) of length ax.n;
// line l starts at (l / inner) * outer + l % inner and its elements are `stride` apart.  src == dst is allowed
// (a block reads all of its lines before it writes any).  lds: 2 * ax.n * TC elements.
template <typename T, class Sync>
RF_HD void generic_axis_block(const cplx<T>* src, cplx<T>* dst, const GenericAxis& ax, long long stride, long long inner,
                              long long outer, long long nlines, int TC, const cplx<T>* root, int sign, T scale,
                              cplx<T>* lds, long long blk, int tid, int nth, Sync sync) {
  const int n = ax.n, total = n * TC;
  const long long l0 = blk * TC;
  cplx<T>*a = lds, *b = lds + total;
  for (int idx = tid; idx < total; idx += nth) {
    const int c = idx % TC, e = idx / TC;
    const long long l = l0 + c;
    cplx<T> v = mk<T>((T)0, (T)0);
    if (l < nlines) v = src[(l / inner) * outer + l % inner + e * stride];
    a[idx] = v;
  }
  sync();
  const cplx<T>* r = generic_line_fft<T>(a, b, ax, TC, root, 1, sign, tid, nth, sync);
  for (int idx = tid; idx < total; idx += nth) {
    const int c = idx % TC, e = idx / TC;
    const long long l = l0 + c;
    if (l < nlines) {
      const cplx<T> v = r[idx];
      dst[(l / inner) * outer + l % inner + e * stride] = mk<T>(v.x * scale, v.y * scale);
    }
  }
}

// Contiguous pass of the packed inverse transform: rows of M + 1 = nz/2 + 1 half-spectrum bins -> nz reals.
// With w = exp(2 pi i / nz):  z[m] = x[2m] + i x[2m+1] = IDFT_M( (X[k] + conj X[M-k]) + i w^k (X[k] - conj X[M-k]) ),
// the imaginary parts of X[0] and X[M] being ignored.  `ax` factors M; root = exp(2 pi i t / nz), t in [0, nz).
// Block `blk` owns rows [blk TR, blk TR + TR); the calling thread's share of (sum, sum of squares) of the outputs
// (after `scale`) is added to s1, s2.  lds: 2 * M * TR elements.
template <typename T, class Sync>
RF_HD void generic_row_c2r_block(const cplx<T>* G, T* W, const GenericAxis& ax, long long nrows, int TR, const cplx<T>* root,
                                 T scale, cplx<T>* lds, long long blk, int tid, int nth, Sync sync, double& s1, double& s2) {
  const int M = ax.n, total = M * TR;
  const long long r0 = blk * TR;
  cplx<T>*a = lds, *b = lds + total;
  for (int idx = tid; idx < total; idx += nth) {
    const int k = idx % M, c = idx / M;             // consecutive threads walk along a row
    cplx<T> z = mk<T>((T)0, (T)0);
    if (r0 + c < nrows) {
      const cplx<T>* X = G + (r0 + c) * (long long)(M + 1);
      if (k == 0) {
        z = mk<T>(X[0].x + X[M].x, X[0].x - X[M].x);
      } else {
        const cplx<T> p = X[k], q = X[M - k];       // conj X[M-k] = (q.x, -q.y)
        const T er = p.x + q.x, ei = p.y - q.y, orr = p.x - q.x, oi = p.y + q.y;
        const cplx<T> w = root[k];
        // e + i w o
        z = mk<T>(er - (w.x * oi + w.y * orr), ei + (w.x * orr - w.y * oi));
      }
    }
    a[k * TR + c] = z;
  }
  sync();
  const cplx<T>* r = generic_line_fft<T>(a, b, ax, TR, root, 2, +1, tid, nth, sync);
  for (int idx = tid; idx < total; idx += nth) {
    const int m = idx % M, c = idx / M;
    if (r0 + c < nrows) {
      const cplx<T> v = r[m * TR + c];
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
                                 cplx<T>* lds, long long blk, int tid, int nth, Sync sync) {
  const int M = ax.n, total = M * TR;
  const long long r0 = blk * TR;
  cplx<T>*a = lds, *b = lds + total;
  for (int idx = tid; idx < total; idx += nth) {
    const int m = idx % M, c = idx / M;
    cplx<T> z = mk<T>((T)0, (T)0);
    if (r0 + c < nrows) {
      const T* in = W + (r0 + c) * (long long)(2 * M) + 2 * m;
      z = mk<T>(in[0], in[1]);
    }
    a[m * TR + c] = z;
  }
  sync();
  const cplx<T>* r = generic_line_fft<T>(a, b, ax, TR, root, 2, -1, tid, nth, sync);
  const int totalo = (M + 1) * TR;
  for (int idx = tid; idx < totalo; idx += nth) {
    const int k = idx % (M + 1), c = idx / (M + 1);
    if (r0 + c < nrows) {
      const cplx<T> p = r[(k % M) * TR + c], q = r[((M - k) % M) * TR + c];
      const T er = (T)0.5 * (p.x + q.x), ei = (T)0.5 * (p.y - q.y), orr = (T)0.5 * (p.x - q.x), oi = (T)0.5 * (p.y + q.y);
      cplx<T> w = root[k];
      w.y = -w.y;
      // e - i w o
      G[(r0 + c) * (long long)(M + 1) + k] = mk<T>(er + (w.x * oi + w.y * orr), ei - (w.x * orr - w.y * oi));
    }
  }
}

// lines / rows per block so that the two LDS buffers stay within 64 KB (no function attribute needed); a single line longer than
// that (n > 4096 complex64 / 2048 complex128) takes what it needs, up to GENERIC_LDS_MAX
inline int generic_lines_per_block(int n, int elem_bytes, int want, long long budget = 65536) {
  int tc = want;
  while (tc > 1 && 2LL * n * tc * elem_bytes > budget) tc >>= 1;
  return tc;
}

}  // namespace rf
