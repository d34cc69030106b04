// kernels of the non-power-of-two path (rf_generic.h): one workgroup per group of lines, lines staged in LDS
#include "rf_generic.h"
#include "rf_launch.h"

namespace rf {
namespace {

struct BlockSync { __device__ void operator()() const { __syncthreads(); } };

template <typename T>
__global__ __launch_bounds__(1024) void generic_axis_kernel(const cplx<T>* src, cplx<T>* dst, GenericAxis ax, long long stride,
                                                          long long inner, long long outer, long long nlines, int TC,
                                                          const cplx<T>* __restrict__ root, int sign, T scale, int tw_lds) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  generic_axis_block<T>(src, dst, ax, stride, inner, outer, nlines, TC, root, sign, scale, reinterpret_cast<cplx<T>*>(lds_raw),
                        (long long)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, BlockSync(), tw_lds);
}

template <typename T>
__global__ __launch_bounds__(256) void generic_row_c2r_kernel(const cplx<T>* __restrict__ G, T* __restrict__ W, GenericAxis ax,
                                                             long long nrows, int TR, const cplx<T>* __restrict__ root, T scale,
                                                             double* __restrict__ partials, int tw_lds) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ double red[2 * 4];
  double s1 = 0.0, s2 = 0.0;
  generic_row_c2r_block<T>(G, W, ax, nrows, TR, root, scale, reinterpret_cast<cplx<T>*>(lds_raw), (long long)blockIdx.x,
                           (int)threadIdx.x, (int)blockDim.x, BlockSync(), s1, s2, tw_lds);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {          // fixed order: deterministic
    double a = 0.0, b = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
    partials[2 * (long long)blockIdx.x] = a;
    partials[2 * (long long)blockIdx.x + 1] = b;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void generic_row_r2c_kernel(const T* __restrict__ W, cplx<T>* __restrict__ G, GenericAxis ax,
                                                             long long nrows, int TR, const cplx<T>* __restrict__ root, int tw_lds) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  generic_row_r2c_block<T>(W, G, ax, nrows, TR, root, reinterpret_cast<cplx<T>*>(lds_raw), (long long)blockIdx.x,
                           (int)threadIdx.x, (int)blockDim.x, BlockSync(), tw_lds);
}

// lines with sub-lines (the two steps of the four-step transform of an axis too long for the LDS: rf_generic.h GenericLines)
template <typename T>
__global__ __launch_bounds__(1024) void generic_lines_kernel(const cplx<T>* src, cplx<T>* dst, GenericLines L, int TC, const cplx<T>* __restrict__ root, int tw_lds) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  generic_lines_block<T>(src, dst, L, TC, root, reinterpret_cast<cplx<T>*>(lds_raw), (long long)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, BlockSync(), tw_lds);
}
template <typename T>
__global__ __launch_bounds__(256) void generic_untangle_kernel(const cplx<T>* __restrict__ G, cplx<T>* __restrict__ Z, int M, long long total,
                                                              const cplx<T>* __restrict__ root) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) generic_untangle_at<T>(G, Z, M, root, i);
}
template <typename T>
__global__ __launch_bounds__(256) void generic_tangle_kernel(const cplx<T>* __restrict__ Z, cplx<T>* __restrict__ G, int M, long long total,
                                                            const cplx<T>* __restrict__ root) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) generic_tangle_at<T>(Z, G, M, root, i);
}
// (sum, sum of squares) of n reals: block b leaves its share in partials[2b], partials[2b + 1] (fixed order inside a block: deterministic)
template <typename T>
__global__ __launch_bounds__(256) void generic_moments_kernel(const T* __restrict__ W, long long n, double* __restrict__ partials) {
  __shared__ double red[2 * 4];
  double s1 = 0.0, s2 = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double v = (double)W[i];
    s1 += v; s2 += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 4; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
    partials[2 * (long long)blockIdx.x] = a;
    partials[2 * (long long)blockIdx.x + 1] = b;
  }
}

// Lines per workgroup, threads and LDS bytes of a strided pass over lines of n elements of `es` bytes.  Neighbouring lines are
// neighbours in memory, so tc lines give segments of tc * es bytes: the widest tile up to 16 lines that fits a CU's LDS (128-byte
// segments with one workgroup per CU beat 64-byte ones with two: 1000^3 14.6 against 16.3 ms); enough threads for 16 waves per CU --
// the stages are chains of dependent LDS and table reads with a barrier between them, and latency is what bounds them.
struct StridedShape { int tc, threads; size_t lds; int tw_lds; };
inline StridedShape strided_shape(const GenericAxis& ax, int es, bool neighbours) {
  const int n = ax.n, bufs = generic_bufs(ax);
  auto lds = [&](int tc) { return (size_t)bufs * n * tc * es + generic_extra_bytes(ax, es); };      // (+ the stage and position tables)
  int tc = neighbours ? generic_strided_tile(ax, es) : 4;
  while (tc > 1 && lds(tc) > (size_t)GENERIC_LDS_MAX) tc >>= 1;
  if (lds(tc) > (size_t)GENERIC_LDS_MAX) return {1, 256, (size_t)bufs * n * es, 0};   // the longest lines that are not smooth: no room for the table
  const int per_cu = (int)((size_t)(GENERIC_LDS_MAX + 256) / (lds(tc) > 0 ? lds(tc) : 1));
  int threads = 256;
  while (threads < 1024 && per_cu * threads < 1024 && (long long)n * tc >= 4LL * threads) threads <<= 1;
  return {tc, threads, lds(tc), 1};
}

template <typename T>
hipError_t lines_t(const void* src, void* dst, const GenericLines& L, const void* root, hipStream_t s) {
  // sub-lines whose parents are neighbours in memory (inner > 1) go 16 to a block, as in axis_t
  const StridedShape sh = strided_shape(L.ax, (int)sizeof(cplx<T>), L.inner_s > 1);
  const int tc = sh.tc;
  const long long nblk = (L.nlines() + tc - 1) / tc;
  if (nblk <= 0) return hipSuccess;
  if (nblk > 0x7fffffffLL || L.nparent <= 0 || L.nsub <= 0) return hipErrorInvalidValue;
  const size_t lds = sh.lds;
  if (lds > (size_t)GENERIC_LDS_MAX) return hipErrorInvalidValue;
  static LdsAttrLatch latch;
  if (lds > 65536)
    if (hipError_t e = latch.ensure((const void*)generic_lines_kernel<T>, GENERIC_LDS_MAX); e != hipSuccess) return e;
  hipLaunchKernelGGL(generic_lines_kernel<T>, dim3((unsigned)nblk), dim3(sh.threads), lds, s, (const cplx<T>*)src, (cplx<T>*)dst, L, tc, (const cplx<T>*)root, sh.tw_lds);
  return hipGetLastError();
}

template <typename T>
hipError_t axis_t(const void* src, void* dst, const GenericAxis& ax, long long stride, long long inner, long long outer,
                  long long nlines, const void* root, int sign, double scale, hipStream_t s) {
  // lines that are neighbours in memory (inner > 1) are transformed 16 at a time: 128-byte (float32) segments
  const StridedShape sh = strided_shape(ax, (int)sizeof(cplx<T>), inner > 1);
  const int tc = sh.tc;
  const long long nblk = (nlines + tc - 1) / tc;
  if (nblk <= 0) return hipSuccess;
  if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
  const size_t lds = sh.lds;
  if (lds > (size_t)GENERIC_LDS_MAX) return hipErrorInvalidValue;
  static LdsAttrLatch latch;
  if (lds > 65536)
    if (hipError_t e = latch.ensure((const void*)generic_axis_kernel<T>, GENERIC_LDS_MAX); e != hipSuccess) return e;
  hipLaunchKernelGGL(generic_axis_kernel<T>, dim3((unsigned)nblk), dim3(sh.threads), lds, s, (const cplx<T>*)src, (cplx<T>*)dst, ax, stride,
                     inner, outer, nlines, tc, (const cplx<T>*)root, sign, (T)scale, sh.tw_lds);
  return hipGetLastError();
}

template <typename T> int rows_per_block(const GenericAxis& ax) { return generic_lines_per_block(ax.n, (int)sizeof(cplx<T>), 8, 49152, generic_bufs(ax), true); }   // + the reduction's static LDS

}  // namespace

hipError_t launch_generic_axis(int f64, const void* src, void* dst, const GenericAxis& ax, long long stride, long long inner,
                               long long outer, long long nlines, const void* root, int sign, double scale, hipStream_t s) {
  return f64 ? axis_t<double>(src, dst, ax, stride, inner, outer, nlines, root, sign, scale, s)
             : axis_t<float>(src, dst, ax, stride, inner, outer, nlines, root, sign, scale, s);
}

hipError_t launch_generic_lines(int f64, const void* src, void* dst, const GenericLines& L, const void* root, hipStream_t s) {
  return f64 ? lines_t<double>(src, dst, L, root, s) : lines_t<float>(src, dst, L, root, s);
}
hipError_t launch_generic_untangle(int f64, const void* G, void* Z, int M, long long nrows, const void* root, hipStream_t s) {
  const long long total = nrows * M;
  const unsigned grid = (unsigned)(total / 256 + 1 < 65536 ? total / 256 + 1 : 65536);
  if (f64) hipLaunchKernelGGL(generic_untangle_kernel<double>, dim3(grid), dim3(256), 0, s, (const cplx<double>*)G, (cplx<double>*)Z, M, total, (const cplx<double>*)root);
  else hipLaunchKernelGGL(generic_untangle_kernel<float>, dim3(grid), dim3(256), 0, s, (const cplx<float>*)G, (cplx<float>*)Z, M, total, (const cplx<float>*)root);
  return hipGetLastError();
}
hipError_t launch_generic_tangle(int f64, const void* Z, void* G, int M, long long nrows, const void* root, hipStream_t s) {
  const long long total = nrows * (M + 1);
  const unsigned grid = (unsigned)(total / 256 + 1 < 65536 ? total / 256 + 1 : 65536);
  if (f64) hipLaunchKernelGGL(generic_tangle_kernel<double>, dim3(grid), dim3(256), 0, s, (const cplx<double>*)Z, (cplx<double>*)G, M, total, (const cplx<double>*)root);
  else hipLaunchKernelGGL(generic_tangle_kernel<float>, dim3(grid), dim3(256), 0, s, (const cplx<float>*)Z, (cplx<float>*)G, M, total, (const cplx<float>*)root);
  return hipGetLastError();
}
hipError_t launch_generic_moments(int f64, const void* W, long long n, double* partials, long long nblocks, hipStream_t s) {
  if (nblocks < 1 || nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
  if (f64) hipLaunchKernelGGL(generic_moments_kernel<double>, dim3((unsigned)nblocks), dim3(256), 0, s, (const double*)W, n, partials);
  else hipLaunchKernelGGL(generic_moments_kernel<float>, dim3((unsigned)nblocks), dim3(256), 0, s, (const float*)W, n, partials);
  return hipGetLastError();
}

long long generic_row_blocks(int f64, const GenericAxis& ax, long long nrows) {
  const int tr = f64 ? rows_per_block<double>(ax) : rows_per_block<float>(ax);
  return (nrows + tr - 1) / tr;
}

namespace {
// rows per workgroup, LDS bytes and whether the stage table fits behind the line image(s)
struct RowShape { int tr; size_t lds; int tw_lds; };
template <typename T> RowShape row_shape(const GenericAxis& ax) {
  const int tr = rows_per_block<T>(ax);
  const size_t base = (size_t)generic_bufs(ax) * ax.n * generic_row_pitch(tr) * sizeof(cplx<T>), with = base + generic_extra_bytes(ax, (int)sizeof(cplx<T>));
  const int tw = with <= (size_t)GENERIC_LDS_MAX ? 1 : 0;
  return {tr, tw ? with : base, tw};
}
template <typename T>
hipError_t row_c2r_t(const void* G, void* W, const GenericAxis& ax, long long nrows, const void* root, double scale, double* partials, long long nblk, hipStream_t s) {
  const RowShape sh = row_shape<T>(ax);
  if (sh.lds > (size_t)GENERIC_LDS_MAX) return hipErrorInvalidValue;
  static LdsAttrLatch latch;
  if (sh.lds > 49152)
    if (hipError_t e = latch.ensure((const void*)generic_row_c2r_kernel<T>, GENERIC_LDS_MAX); e != hipSuccess) return e;
  hipLaunchKernelGGL(generic_row_c2r_kernel<T>, dim3((unsigned)nblk), dim3(256), sh.lds, s, (const cplx<T>*)G, (T*)W, ax, nrows, sh.tr, (const cplx<T>*)root, (T)scale,
                     partials, sh.tw_lds);
  return hipGetLastError();
}
template <typename T>
hipError_t row_r2c_t(const void* W, void* G, const GenericAxis& ax, long long nrows, const void* root, long long nblk, hipStream_t s) {
  const RowShape sh = row_shape<T>(ax);
  if (sh.lds > (size_t)GENERIC_LDS_MAX) return hipErrorInvalidValue;
  static LdsAttrLatch latch;
  if (sh.lds > 49152)
    if (hipError_t e = latch.ensure((const void*)generic_row_r2c_kernel<T>, GENERIC_LDS_MAX); e != hipSuccess) return e;
  hipLaunchKernelGGL(generic_row_r2c_kernel<T>, dim3((unsigned)nblk), dim3(256), sh.lds, s, (const T*)W, (cplx<T>*)G, ax, nrows, sh.tr, (const cplx<T>*)root, sh.tw_lds);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_generic_row_c2r(int f64, const void* G, void* W, const GenericAxis& ax, long long nrows, const void* root,
                                  double scale, double* partials, hipStream_t s) {
  const long long nblk = generic_row_blocks(f64, ax, nrows);
  if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
  return f64 ? row_c2r_t<double>(G, W, ax, nrows, root, scale, partials, nblk, s) : row_c2r_t<float>(G, W, ax, nrows, root, scale, partials, nblk, s);
}

hipError_t launch_generic_row_r2c(int f64, const void* W, void* G, const GenericAxis& ax, long long nrows, const void* root,
                                  hipStream_t s) {
  const long long nblk = generic_row_blocks(f64, ax, nrows);
  if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
  return f64 ? row_r2c_t<double>(W, G, ax, nrows, root, nblk, s) : row_r2c_t<float>(W, G, ax, nrows, root, nblk, s);
}

}  // namespace rf
