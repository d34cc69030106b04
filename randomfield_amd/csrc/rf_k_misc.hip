// elementwise / reduction kernels: standalone generation (rows K,T,R,S in the API
// layout), moments finalisation (row D), lognormal map and per-z scaling (row L),
// save_potential helpers.
#include "rf_kernels.h"
#include "rf_launch.h"

namespace rf {
namespace {

// Stand-in for the RCCL all-to-all of a kz-slab rank (rf_slab_set_exchange_standin, randomfield_hip_diag.h): nblk blocks of `bytes`
// bytes each are read from src[b] and written to dst[b] by a FIXED number of workgroups -- the footprint of RCCL's send / receive
// channels on the compute units (one 256-thread workgroup per channel) and in local HBM (every block read once, every block written
// once), without the links.  Eight 16-byte loads in flight per thread, then eight stores; grid-stride over the blocks in turn.
struct StandinBlocks { const char* src[16]; char* dst[16]; };
// read_pct / write_pct (0 .. 100): the share of every block that is read / written -- 100 / 100 is the copy; other values take the two
// directions of the exchange's local traffic apart (reads without their writes, writes of a constant without reads, half of both ...)
__global__ __launch_bounds__(256) void exchange_standin_kernel(StandinBlocks blk, int nblk, unsigned long long bytes, int read_pct, int write_pct,
                                                               unsigned* __restrict__ sink) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) u4 gu4;
  const unsigned long long nvec = bytes / 16, stride = (unsigned long long)gridDim.x * 256;
  const unsigned long long nr = nvec / 100 * (unsigned)read_pct, nw = nvec / 100 * (unsigned)write_pct;
  const bool copy = read_pct == 100 && write_pct == 100;
  u4 acc = {0u, 0u, 0u, 0u};
  for (int b = 0; b < nblk; ++b) {
    const gu4* s = (const gu4*)blk.src[b];
    gu4* d = (gu4*)blk.dst[b];
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (copy) {
      for (; i + 7 * stride < nvec; i += 8 * stride) {
        u4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(s + i + k * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k], d + i + k * stride);
      }
      for (; i < nvec; i += stride) d[i] = s[i];
    } else {
      for (unsigned long long j = i; j + 7 * stride < nr; j += 8 * stride) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= __builtin_nontemporal_load(s + j + k * stride);
      }
      const u4 c = {(unsigned)b, 1u, 2u, 3u};
      for (unsigned long long j = i; j + 7 * stride < nw; j += 8 * stride) {
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(c, d + j + k * stride);
      }
    }
  }
  if (!copy && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && sink) *sink = acc.x;      // (keeps the reads alive; practically never taken)
}

// rf_comm_enable_direct's proof that the peers' receive buffers are really mapped: thread t stores `value` into word `slot` of bases[t]
// -- a store from a kernel through the mapped pointer, which is what the storing y pass will do
__global__ __launch_bounds__(64) void peer_mark_kernel(void* const* __restrict__ bases, int n, int slot, unsigned long long value) {
  const int t = threadIdx.x;
  if (t < n && bases[t]) reinterpret_cast<unsigned long long*>(bases[t])[slot] = value;
}

// Rows (ix, iy) of the half spectrum: blockDim.x threads walk along kz, blockDim.y rows per workgroup (short rows share a wave); a
// row's indices are formed once per row with 32-bit divisions -- the flat form paid four 64-bit divisions per cell, more than the
// cell's own arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void gen_kspace_kernel(cplx<T>* __restrict__ K, GenParams gp, unsigned nrows) {
  const int nzh = gp.zpitch;            // this rank's planes + the Nyquist plane (nz/2 + 1 on one rank)
  const uint64_t seed = gp.seed_dev ? *gp.seed_dev : gp.seed;
  for (unsigned long long r0 = (unsigned long long)blockIdx.x * blockDim.y; r0 < nrows; r0 += (unsigned long long)gridDim.x * blockDim.y) {
    const unsigned long long rr = r0 + threadIdx.y;
    if (rr >= nrows) continue;
    const unsigned row = (unsigned)rr, ix = row / (unsigned)gp.ny, iy = row - ix * (unsigned)gp.ny;
    cplx<T>* Kr = K + (long long)row * nzh;
    for (int sl = threadIdx.x; sl < nzh; sl += blockDim.x) {
      const int iz = sl == nzh - 1 ? gp.nz / 2 : gp.zoff + sl;
      Kr[sl] = gen_cell<T>(gp, seed, (int)ix, (int)iy, iz);
    }
  }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ partials, long long n,
                                                             double* __restrict__ out, long long chunk) {
  // block b sums pairs [b*chunk, min(n, (b+1)*chunk)) -> out[2b], out[2b+1]; fixed order: deterministic
  __shared__ double red[2 * 4];
  const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  double a = 0, b = 0;
  for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) { a += partials[2 * i]; b += partials[2 * i + 1]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[2 * wave] = a; red[2 * wave + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = 0; b = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
    out[2 * (long long)blockIdx.x] = a;
    out[2 * (long long)blockIdx.x + 1] = b;
  }
}

// sigma of the field from the y pass's Parseval partials (AccColIO), and the tables of the fused lognormal map that depend on it
// (cosmotools.py:216: t = 1 + (sigma growth_z)^2): Ap = sqrt(log t) / sigma, Bp = density / sqrt(t).  One workgroup; fixed
// summation order.  `sigma_as_float`: the reference's sigma is np.std of the array, a float32 for float32 data -- same here.
// (1024 threads, four independent accumulators each: the ~10^5 partials of a 1024^3 grid are summed with the loads in flight -- the
// 256-thread serial form took 0.11 ms, a latency chain of 512 dependent loads per thread.)  `ap_unit`: float64 plans hand the z pass
// Ap in units of ln2 / 64 (rf_fft.h exp_scaled64).
__global__ __launch_bounds__(1024) void lognormal_tables_kernel(const double* __restrict__ partials, long long n, double norm,
                                                                const double* __restrict__ growth, const double* __restrict__ density,
                                                                int nz, int sigma_as_float, double ap_unit,
                                                                double* __restrict__ sig, double* __restrict__ Ap, double* __restrict__ Bp) {
  __shared__ double red[16];
  __shared__ double sh_sigma;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  const long long nt = blockDim.x;
  long long i = threadIdx.x;
  for (; i + 3 * nt < n; i += 4 * nt) {
    a0 += partials[i]; a1 += partials[i + nt]; a2 += partials[i + 2 * nt]; a3 += partials[i + 3 * nt];
  }
  for (; i < n; i += nt) a0 += partials[i];
  double a = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    double sigma = sqrt(t * norm);
    if (sigma_as_float) sigma = (double)(float)sigma;
    sh_sigma = sigma;
    sig[0] = sigma;
  }
  __syncthreads();
  const double sigma = sh_sigma;
  for (int z = threadIdx.x; z < nz; z += blockDim.x) {
    const double g = sigma * growth[z], t = g * g + 1.0;      // np.square(sigma * growth) + 1
    Ap[z] = sqrt(log(t)) / sigma * ap_unit;
    Bp[z] = (density ? density[z] : 1.0) / sqrt(t);
  }
}

// streaming (non-temporal) vector accesses of the elementwise maps: every element is read once and written once
template <typename T, int VEC> struct MapVec {
  typedef T vt __attribute__((ext_vector_type(VEC)));
  T v[VEC];
  __device__ static MapVec load(const T* base, long long i) {
    const vt x = __builtin_nontemporal_load(reinterpret_cast<const vt*>(base) + i);
    MapVec r;
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.v[e] = x[e];
    return r;
  }
  __device__ void store(T* base, long long i) const {
    vt x;
#pragma unroll
    for (int e = 0; e < VEC; ++e) x[e] = v[e];
    __builtin_nontemporal_store(x, reinterpret_cast<vt*>(base) + i);
  }
};

// cosmotools.py:216-220: delta /= sigma; delta *= sqrt(log t); delta = exp(delta); delta /= sqrt(t)
// each step rounded to the array dtype; the two table factors are float64 (growth is a float64 (nz,) array)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void lognormal_vec_kernel(T* __restrict__ W, long long nvec, int nz,
                                                            const double* __restrict__ a_z,
                                                            const double* __restrict__ b_z, T sigma) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    MapVec<T, VEC> x = MapVec<T, VEC>::load(W, i);
    const int iz0 = (int)((i * VEC) % nz);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      T d = x.v[e] / sigma;
      d = (T)((double)d * a_z[iz0 + e]);
      d = sizeof(T) == 4 ? (T)expf((float)d) : (T)exp((double)d);
      d = (T)((double)d / b_z[iz0 + e]);
      x.v[e] = d;
    }
    x.store(W, i);
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void affine_vec_kernel(T* __restrict__ W, long long nvec, int nz,
                                                         const double* __restrict__ mul_z, double add, int has_add) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    MapVec<T, VEC> x = MapVec<T, VEC>::load(W, i);
    const int iz0 = (int)((i * VEC) % nz);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      T d = (T)((double)x.v[e] * mul_z[iz0 + e]);   // delta *= f64 (nz,) array: f64 product rounded to T
      if (has_add) d = d + (T)add;                   // delta += 1 in the array dtype
      x.v[e] = d;
    }
    x.store(W, i);
  }
}

// generate.py:205-215: potential = delta(k) / k^2, 0 at DC; k^2 rounded to the array dtype in two steps
template <typename T>
__global__ __launch_bounds__(256) void save_potential_kernel(const cplx<T>* __restrict__ K, cplx<T>* __restrict__ P,
                                                             int nx, int ny, int nz, const double* __restrict__ kx2,
                                                             const double* __restrict__ ky2, const double* __restrict__ kz2,
                                                             int zpitch, int zoff, int ppitch) {
  const int nzh = zpitch;
  const long long total = (long long)nx * ny * nzh;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long long)gridDim.x * blockDim.x) {
    const int sl = (int)(c % nzh), iz = sl == nzh - 1 ? nz / 2 : zoff + sl;
    const long long col = c / nzh;
    const int iy = (int)(col % ny), ix = (int)(col / ny);
    T t = (T)(kx2[ix] + ky2[iy]);
    t = (T)((double)t + kz2[iz]);
    T inv = (T)1 / t;
    cplx<T> d = K[c];
    // complex (inv + 0i) * d: the 0*x terms of numpy's complex product vanish exactly
    cplx<T> r = mk<T>(inv * d.x, inv * d.y);
    if (ix == 0 && iy == 0 && iz == 0) r = mk<T>((T)0, (T)0);
    P[col * ppitch + sl] = r;                 // the potential array's rows may be padded (FastGenParams::ppitch)
  }
}

// K (rows of zpitch cells) = scale * P (rows of ppitch >= zpitch cells)
template <typename T>
__global__ __launch_bounds__(256) void scale_copy_kernel(const cplx<T>* __restrict__ P, cplx<T>* __restrict__ K,
                                                         long long n, int zpitch, int ppitch, T scale) {
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (long long)gridDim.x * blockDim.x) {
    const long long col = c / zpitch;
    cplx<T> p = P[col * ppitch + (c - col * zpitch)];
    K[c] = mk<T>(p.x * scale, p.y * scale);
  }
}

// after the forward x and y passes slot kz = 0 holds C = A0 + i A_nyq with A0, A_nyq Hermitian in (kx, ky):
// A0(k) = (C(k) + conj C(-k)) / 2,  A_nyq(k) = (C(k) - conj C(-k)) / (2i)
// W: [nx][ny][nzl] (this rank's kz planes kz0 .. kz0 + nzl; one rank: nzl = nz/2, kz0 = 0);  K: the side array [nx][ny][nzl + 1]
// (own planes, then the Nyquist plane -- filled by the rank that owns kz = 0, zero on the others, which never read it)
template <typename T>
__global__ __launch_bounds__(256) void unpack_kspace_kernel(const cplx<T>* __restrict__ W, cplx<T>* __restrict__ K,
                                                            int nx, int ny, int nzl, int kz0) {
  const int nzh = nzl + 1;
  const long long total = (long long)nx * ny * nzh;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long long)gridDim.x * blockDim.x) {
    const int iz = (int)(c % nzh);
    const long long col = c / nzh;
    const int iy = (int)(col % ny), ix = (int)(col / ny);
    if (iz < nzl && kz0 + iz > 0) { K[c] = W[col * nzl + iz]; continue; }
    if (kz0 != 0) { K[c] = mk<T>((T)0, (T)0); continue; }           // the Nyquist slot of a rank that does not hold it
    const cplx<T> a = W[col * nzl];
    const long long mcol = (long long)((nx - ix) % nx) * ny + (ny - iy) % ny;
    const cplx<T> b = W[mcol * nzl];
    if (iz == 0) K[c] = mk<T>((T)0.5 * (a.x + b.x), (T)0.5 * (a.y - b.y));
    else         K[c] = mk<T>((T)0.5 * (a.y + b.y), (T)0.5 * (b.x - a.x));
  }
}

inline unsigned grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;   // 16 blocks per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ---- lensing potential (generate.py:352-416): psi[e] = Simpson integral over j in [i_min, e] of
// -2 (cot[j] - cot[e]) phi[j] dD along z.  The reference recomputes the integral for every endpoint e (O(nz^2)
// per column); with a uniform step h the composite rule of scipy.integrate.simps(even='avg') is a closed form in
// the prefix sums of the samples at even and odd positions, for the two sequences a_j = cot[j] phi[j] and
// b_j = phi[j]:  psi[e] = -2 (S_a(e) - cot[e] S_b(e)).  One wave per (x, y) row: every lane owns nz/64
// consecutive samples, the four prefix sums cross the lanes with a wave scan, accumulation in float64.
__device__ __forceinline__ double lens_simps(double E, double O, double y0, double y1, double yp, double ym, int m, double h) {
  // E, O: inclusive sums over even / odd positions <= m; y0, y1, yp = y[m-1], ym = y[m]
  if (m == 0) return 0.0;
  if ((m & 1) == 0) return (h / 3.0) * (4.0 * O + 2.0 * E - y0 - ym);
  const double A = (m >= 3 ? (h / 3.0) * (4.0 * (O - ym) + 2.0 * E - y0 - yp) : 0.0) + 0.5 * h * (yp + ym);
  const double B = 0.5 * h * (y0 + y1) + (m >= 3 ? (h / 3.0) * (4.0 * (E - y0) + 2.0 * O - y1 - ym) : 0.0);
  return 0.5 * (A + B);
}

__device__ __forceinline__ double wave_excl_scan(double v, int lane) {
  double inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  return inc - v;
}

// A wave walks its row in segments of 64 * VW samples (VW = samples per 16-byte vector): every lane loads ONE vector
// per segment (a fully coalesced 1-KiB wave access), the four running sums and the previous sample are carried from
// segment to segment (lane 63's inclusive values).  Rows shorter than a segment use guarded scalar accesses.
template <typename T>
__global__ __launch_bounds__(256) void lensing_kernel(const T* __restrict__ phi, T* __restrict__ psi, long long nrows, int nz,
                                                      const double* __restrict__ cot, double h, int i_min) {
  constexpr int VW = 16 / (int)sizeof(T);
  struct alignas(16) Vec { T e[VW]; };
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;                         // the whole wave leaves together
  const T* in = phi + row * (long long)nz;
  T* out = psi + row * (long long)nz;
  const bool vec = nz % (64 * VW) == 0;             // whole segments, 16-byte aligned rows; anything else: guarded scalar accesses
  const double b0 = (double)in[i_min], a0 = cot[i_min] * b0;
  const double b1 = i_min + 1 < nz ? (double)in[i_min + 1] : 0.0, a1 = i_min + 1 < nz ? cot[i_min + 1] * b1 : 0.0;
  double cEa = 0, cOa = 0, cEb = 0, cOb = 0;        // sums over all earlier segments
#pragma unroll 1
  for (int seg = 0; seg < nz; seg += 64 * VW) {
    const int j0 = seg + lane * VW;
    Vec v;
    if (vec) v = *reinterpret_cast<const Vec*>(in + j0);
    else {
#pragma unroll
      for (int e = 0; e < VW; ++e) v.e[e] = j0 + e < nz ? in[j0 + e] : (T)0;
    }
    double c[VW];
    double Ea = 0, Oa = 0, Eb = 0, Ob = 0;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      const int j = j0 + e;
      c[e] = j < nz ? cot[j] : 0.0;
      if (j < nz && j >= i_min) {
        const double x = (double)v.e[e], a = c[e] * x;
        if ((j - i_min) & 1) { Oa += a; Ob += x; } else { Ea += a; Eb += x; }
      }
    }
    const double tEa = Ea, tOa = Oa, tEb = Eb, tOb = Ob;                 // this lane's share of the segment
    Ea = cEa + wave_excl_scan(Ea, lane); Oa = cOa + wave_excl_scan(Oa, lane);
    Eb = cEb + wave_excl_scan(Eb, lane); Ob = cOb + wave_excl_scan(Ob, lane);
    // inclusive totals at the end of the segment = lane 63's exclusive prefix + its own share
    cEa = __shfl(Ea + tEa, 63); cOa = __shfl(Oa + tOa, 63); cEb = __shfl(Eb + tEb, 63); cOb = __shfl(Ob + tOb, 63);
    double bp = 0.0, ap = 0.0;                      // the sample before this lane's first one
    if (j0 - 1 >= i_min && j0 - 1 < nz) { bp = (double)in[j0 - 1]; ap = cot[j0 - 1] * bp; }
    Vec o;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      const int j = j0 + e;
      o.e[e] = (T)0;
      if (j < nz && j >= i_min) {
        const int m = j - i_min;
        const double x = (double)v.e[e], a = c[e] * x;
        if (m & 1) { Oa += a; Ob += x; } else { Ea += a; Eb += x; }
        const double Sa = lens_simps(Ea, Oa, a0, a1, ap, a, m, h);
        const double Sb = lens_simps(Eb, Ob, b0, b1, bp, x, m, h);
        o.e[e] = (T)(-2.0 * (Sa - c[e] * Sb));
        ap = a; bp = x;
      }
    }
    if (vec) *reinterpret_cast<Vec*>(out + j0) = o;
    else {
#pragma unroll
      for (int e = 0; e < VW; ++e) if (j0 + e < nz) out[j0 + e] = o.e[e];
    }
  }
}

// The same scan with the whole row in flight: rows of exactly NSEG segments (NSEG * 64 * VW samples, 16-byte aligned).  All NSEG
// 16-byte loads of a lane go out before the first scan -- with one vector per wave in flight (the loop above) 4096 resident waves
// hold 4 MiB, a quarter of what 8 TB/s times the memory latency needs, and that kernel ran at 1.7 TB/s -- the segments' scans are
// independent of each other until their carries are added, and the sample in front of a lane's first one comes from the neighbour
// lane (or the previous segment's last lane) by a shuffle instead of a second, dependent trip to memory in the middle of every
// segment.  Same arithmetic in the same order: bit-identical results; 1024^3 float32 4.74 -> 3.25 ms (float64 arithmetic is what
// is left: ~36 float64 operations per sample).  Tried and dropped: every lane owning nz/64 CONSECUTIVE samples (one scan per row
// instead of one per segment): its 16-byte loads 64 bytes apart cost more than the scans save (4.8 ms).
template <typename T, int NSEG>
__global__ __launch_bounds__(256) void lensing_rows_kernel(const T* __restrict__ phi, T* __restrict__ psi, long long nrows, int nz,
                                                           const double* __restrict__ cot, double h, int i_min) {
  constexpr int VW = 16 / (int)sizeof(T);
  struct alignas(16) Vec { T e[VW]; };
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;                         // the whole wave leaves together
  const T* in = phi + row * (long long)nz;
  T* out = psi + row * (long long)nz;
  Vec v[NSEG];
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg) v[sg] = *reinterpret_cast<const Vec*>(in + sg * 64 * VW + lane * VW);
  const double b0 = (double)in[i_min], a0 = cot[i_min] * b0;
  const double b1 = i_min + 1 < nz ? (double)in[i_min + 1] : 0.0, a1 = i_min + 1 < nz ? cot[i_min + 1] * b1 : 0.0;
  double cEa = 0, cOa = 0, cEb = 0, cOb = 0;        // sums over all earlier segments
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg) {
    const int j0 = sg * 64 * VW + lane * VW;
    double c[VW];
    double Ea = 0, Oa = 0, Eb = 0, Ob = 0;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      const int j = j0 + e;
      c[e] = cot[j];
      if (j >= i_min) {
        const double x = (double)v[sg].e[e], a = c[e] * x;
        if ((j - i_min) & 1) { Oa += a; Ob += x; } else { Ea += a; Eb += x; }
      }
    }
    const double tEa = Ea, tOa = Oa, tEb = Eb, tOb = Ob;
    Ea = cEa + wave_excl_scan(Ea, lane); Oa = cOa + wave_excl_scan(Oa, lane);
    Eb = cEb + wave_excl_scan(Eb, lane); Ob = cOb + wave_excl_scan(Ob, lane);
    cEa = __shfl(Ea + tEa, 63); cOa = __shfl(Oa + tOa, 63); cEb = __shfl(Eb + tEb, 63); cOb = __shfl(Ob + tOb, 63);
    // the sample before this lane's first one: the neighbour lane's last, or the previous segment's very last
    T prev = __shfl_up(v[sg].e[VW - 1], 1);
    if (sg > 0) { const T tail = __shfl(v[sg - 1].e[VW - 1], 63); if (lane == 0) prev = tail; }
    double bp = 0.0, ap = 0.0;
    if (j0 - 1 >= i_min) { bp = (double)prev; ap = cot[j0 - 1] * bp; }
    Vec o;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      const int j = j0 + e;
      o.e[e] = (T)0;
      if (j >= i_min) {
        const int m = j - i_min;
        const double x = (double)v[sg].e[e], a = c[e] * x;
        if (m & 1) { Oa += a; Ob += x; } else { Ea += a; Eb += x; }
        const double Sa = lens_simps(Ea, Oa, a0, a1, ap, a, m, h);
        const double Sb = lens_simps(Eb, Ob, b0, b1, bp, x, m, h);
        o.e[e] = (T)(-2.0 * (Sa - c[e] * Sb));
        ap = a; bp = x;
      }
    }
    *reinterpret_cast<Vec*>(out + j0) = o;
  }
}

}  // namespace

hipError_t launch_exchange_standin(const void* const* src, void* const* dst, int nblk, size_t bytes, int workgroups, int read_pct, int write_pct,
                                   unsigned* sink, hipStream_t s) {
  if (nblk < 0 || nblk > 16 || workgroups < 1 || bytes % 16 || read_pct < 0 || read_pct > 100 || write_pct < 0 || write_pct > 100) return hipErrorInvalidValue;
  if (nblk == 0) return hipSuccess;
  StandinBlocks blk;
  for (int b = 0; b < 16; ++b) { blk.src[b] = b < nblk ? (const char*)src[b] : nullptr; blk.dst[b] = b < nblk ? (char*)dst[b] : nullptr; }
  hipLaunchKernelGGL(exchange_standin_kernel, dim3((unsigned)workgroups), dim3(256), 0, s, blk, nblk, (unsigned long long)bytes, read_pct, write_pct, sink);
  return hipGetLastError();
}

hipError_t launch_peer_mark(void* const* bases_dev, int n, int slot, unsigned long long value, hipStream_t s) {
  if (n < 1 || n > 64 || slot < 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(peer_mark_kernel, dim3(1), dim3(64), 0, s, bases_dev, n, slot, value);
  return hipGetLastError();
}

hipError_t launch_gen_kspace(int f64, void* K, const GenParams& gp, hipStream_t s) {
  const long long nrows = (long long)gp.nx * gp.ny;
  if (nrows <= 0 || nrows > 0x7fffffffLL) return hipErrorInvalidValue;
  // 256 threads: tx along a row (the power of two that covers it, up to 256), ty rows per workgroup; many rows: a grid-stride loop
  int tx = 1;
  while (tx < 256 && tx < gp.zpitch) tx <<= 1;
  const int ty = 256 / tx;
  const long long nblk = (nrows + ty - 1) / ty;
  const unsigned grid = (unsigned)(nblk < (1LL << 22) ? nblk : (1LL << 22));
  if (f64) hipLaunchKernelGGL(gen_kspace_kernel<double>, dim3(grid), dim3(tx, ty), 0, s, (cplx<double>*)K, gp, (unsigned)nrows);
  else hipLaunchKernelGGL(gen_kspace_kernel<float>, dim3(grid), dim3(tx, ty), 0, s, (cplx<float>*)K, gp, (unsigned)nrows);
  return hipGetLastError();
}

hipError_t launch_unpack_kspace(int f64, const void* W, void* K, int nx, int ny, int nzl, int kz0, hipStream_t s) {
  const long long total = (long long)nx * ny * (nzl + 1);
  if (f64) hipLaunchKernelGGL(unpack_kspace_kernel<double>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const cplx<double>*)W, (cplx<double>*)K, nx, ny, nzl, kz0);
  else hipLaunchKernelGGL(unpack_kspace_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const cplx<float>*)W, (cplx<float>*)K, nx, ny, nzl, kz0);
  return hipGetLastError();
}

// two levels: 256 blocks reduce chunks of the partials into `scratch` (2*256 doubles), one block finishes
hipError_t launch_reduce_partials(const double* partials, long long n, double* stats, double* scratch, hipStream_t s) {
  const int nb = n > 4096 ? 256 : 1;
  if (nb == 1) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, s, partials, n, stats, n);
    return hipGetLastError();
  }
  const long long chunk = (n + nb - 1) / nb;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(nb), dim3(256), 0, s, partials, n, scratch, chunk);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, s, (const double*)scratch, (long long)nb, stats, (long long)nb);
  return hipGetLastError();
}

hipError_t launch_lognormal_tables(const double* partials, long long n, double norm, const double* growth, const double* density, int nz,
                                   int sigma_as_float, double ap_unit, double* sig, double* A, double* B, hipStream_t s) {
  hipLaunchKernelGGL(lognormal_tables_kernel, dim3(1), dim3(1024), 0, s, partials, n, norm, growth, density, nz, sigma_as_float, ap_unit, sig, A, B);
  return hipGetLastError();
}

hipError_t launch_lognormal(int f64, void* W, long long nrows, int nz, const double* a_z, const double* b_z,
                            double sigma, hipStream_t s) {
  const long long total = nrows * nz;
  if (f64) {
    const long long nvec = total / 2;
    hipLaunchKernelGGL((lognormal_vec_kernel<double, 2>), dim3(grid_for(nvec, 256)), dim3(256), 0, s, (double*)W, nvec, nz, a_z, b_z, sigma);
  } else if (nz % 4 == 0) {
    const long long nvec = total / 4;
    hipLaunchKernelGGL((lognormal_vec_kernel<float, 4>), dim3(grid_for(nvec, 256)), dim3(256), 0, s, (float*)W, nvec, nz, a_z, b_z, (float)sigma);
  } else {                       // nz = 2 (mod 4) (non-power-of-two grids): a vector must not straddle two rows
    const long long nvec = total / 2;
    hipLaunchKernelGGL((lognormal_vec_kernel<float, 2>), dim3(grid_for(nvec, 256)), dim3(256), 0, s, (float*)W, nvec, nz, a_z, b_z, (float)sigma);
  }
  return hipGetLastError();
}

hipError_t launch_affine_z(int f64, void* W, long long nrows, int nz, const double* mul_z, double add, hipStream_t s) {
  const long long total = nrows * nz;
  const int has_add = add != 0.0;
  if (f64) {
    const long long nvec = total / 2;
    hipLaunchKernelGGL((affine_vec_kernel<double, 2>), dim3(grid_for(nvec, 256)), dim3(256), 0, s, (double*)W, nvec, nz, mul_z, add, has_add);
  } else if (nz % 4 == 0) {
    const long long nvec = total / 4;
    hipLaunchKernelGGL((affine_vec_kernel<float, 4>), dim3(grid_for(nvec, 256)), dim3(256), 0, s, (float*)W, nvec, nz, mul_z, add, has_add);
  } else {
    const long long nvec = total / 2;
    hipLaunchKernelGGL((affine_vec_kernel<float, 2>), dim3(grid_for(nvec, 256)), dim3(256), 0, s, (float*)W, nvec, nz, mul_z, add, has_add);
  }
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_lensing_t(const T* phi, T* psi, long long nrows, int nz, const double* cot_z, double h, int i_min, hipStream_t s) {
  constexpr int SEG = 64 * (16 / (int)sizeof(T));
  const dim3 grid((unsigned)((nrows + 3) / 4));
  const bool aligned = ((uintptr_t)phi % 16 == 0) && ((uintptr_t)psi % 16 == 0);
#define RF_LENS(NS) hipLaunchKernelGGL((lensing_rows_kernel<T, NS>), grid, dim3(256), 0, s, phi, psi, nrows, nz, cot_z, h, i_min)
  // (float64 rows: 5.9 ms either way at 1024^3 -- they keep the loop)
  const bool rows = aligned && sizeof(T) == 4;
  if (rows && nz == SEG) RF_LENS(1);
  else if (rows && nz == 2 * SEG) RF_LENS(2);
  else if (rows && nz == 4 * SEG) RF_LENS(4);
  else if (rows && nz == 8 * SEG) RF_LENS(8);
  else hipLaunchKernelGGL(lensing_kernel<T>, grid, dim3(256), 0, s, phi, psi, nrows, nz, cot_z, h, i_min);
#undef RF_LENS
  return hipGetLastError();
}

hipError_t launch_lensing(int f64, const void* phi, void* psi, long long nrows, int nz, const double* cot_z, double h, int i_min,
                          hipStream_t s) {
  return f64 ? launch_lensing_t<double>((const double*)phi, (double*)psi, nrows, nz, cot_z, h, i_min, s)
             : launch_lensing_t<float>((const float*)phi, (float*)psi, nrows, nz, cot_z, h, i_min, s);
}

hipError_t launch_save_potential(int f64, const void* K, void* P, int nx, int ny, int nz, const double* kx2,
                                 const double* ky2, const double* kz2, int zpitch, int zoff, int ppitch, hipStream_t s) {
  const long long total = (long long)nx * ny * zpitch;
  if (f64) hipLaunchKernelGGL(save_potential_kernel<double>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const cplx<double>*)K, (cplx<double>*)P, nx, ny, nz, kx2, ky2, kz2, zpitch, zoff, ppitch);
  else hipLaunchKernelGGL(save_potential_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const cplx<float>*)K, (cplx<float>*)P, nx, ny, nz, kx2, ky2, kz2, zpitch, zoff, ppitch);
  return hipGetLastError();
}

hipError_t launch_scale_copy(int f64, const void* P, void* K, long long n, int zpitch, int ppitch, double scale, hipStream_t s) {
  if (f64) hipLaunchKernelGGL(scale_copy_kernel<double>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const cplx<double>*)P, (cplx<double>*)K, n, zpitch, ppitch, scale);
  else hipLaunchKernelGGL(scale_copy_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const cplx<float>*)P, (cplx<float>*)K, n, zpitch, ppitch, (float)scale);
  return hipGetLastError();
}
}  // namespace rf
