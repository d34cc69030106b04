// z pass of the c2r transform (contiguous rows, Hermitian untangle fused, moments)
#include "rf_kernels.h"
#include "rf_launch.h"


namespace rf {
namespace {
template <class C>
constexpr int row_lds_bytes() { return C::LDS_BYTES > 2 * (C::NT / 64) * 8 ? C::LDS_BYTES : 2 * (C::NT / 64) * 8; }

template <class C, class IO>
hipError_t launch_one(const IO& io, long long nrows, const cplx<typename C::T>* tw,
                      double* partials, hipStream_t s, bool prepare_only) {
  const long long ntiles = (nrows + C::NRT - 1) / C::NRT;
  auto k = row_c2r_kernel<C, IO>;
  constexpr int lds = row_lds_bytes<C>();
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds); e != hipSuccess) return e;
  if (prepare_only) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), lds, s, io, tw, nrows, partials);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_t(int M, cplx<T>* W, long long nrows, double scale, const cplx<T>* tw, double* partials, hipStream_t s, bool po) {
  PlainRowIO<T> io; io.base = W; io.scale = (T)scale; io.M_of = M;
  switch (M) {
#define X(MM) case MM: return launch_one<typename RowSel<T, MM>::type, PlainRowIO<T>>(io, nrows, tw, partials, s, po);
    RF_ROW_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
template <typename T>
hipError_t launch_gather_t(int M, const cplx<T>* src, cplx<T>* dst, long long nrows, double scale, int nzl,
                           long long seg_stride, const cplx<T>* tw, double* partials, hipStream_t s, bool po) {
  GatherRowIO<T> io; io.src = src; io.dst = dst; io.scale = (T)scale; io.M_of = M; io.nzl = nzl; io.seg_stride = seg_stride;
  if (!po && (nzl <= 0 || (nzl & (nzl - 1)))) return hipErrorInvalidValue;       // (the kernel splits k by shift and mask)
  switch (M) {
#define X(MM) case MM: return launch_one<typename RowSel<T, MM>::type, GatherRowIO<T>>(io, nrows, tw, partials, s, po);
    RF_ROW_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
template <typename T>
hipError_t launch_xgather_t(int M, const cplx<T>* src, cplx<T>* dst, long long nrows, double scale, int tc, int rb, int ny,
                            const cplx<T>* tw, double* partials, hipStream_t s, bool po) {
  XGatherRowIO<T> io; io.src = src; io.dst = dst; io.scale = (T)scale; io.M_of = M;
  auto pow2 = [](long long v) { return v > 0 && (v & (v - 1)) == 0; };
  if (!po && (!pow2(tc) || !pow2(rb) || !pow2(ny) || M % tc || nrows % rb)) return hipErrorInvalidValue;
  io.seg_shift = ilog2ll(tc > 0 ? tc : 1); io.rb_shift = ilog2ll(rb > 0 ? rb : 1); io.ny_shift = ilog2ll(ny > 0 ? ny : 1);
  io.kt_stride = (long long)ny * rb * tc; io.xb_stride = (long long)(M / (tc > 0 ? tc : 1)) * io.kt_stride;
  switch (M) {
#define X(MM) case MM: { using C = typename RowSel<T, MM>::type;                                                           \
    if (!po && rb % C::NRT) return hipErrorInvalidValue;                   /* whole workgroups per (x block, iy) */              \
    return launch_one<C, XGatherRowIO<T>>(io, nrows, tw, partials, s, po); }
    RF_ROW_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
// can the gathering z pass of rows of M complex serve tiles of tc columns and x blocks of rb rows?
template <typename T> bool xgather_ok_t(int M, int tc, int rb) {
  switch (M) {
#define X(MM) case MM: { using C = typename RowSel<T, MM>::type; return tc > 0 && rb > 0 && rb % C::NRT == 0 && M % tc == 0; }
    RF_ROW_SIZES(X)
#undef X
    default: return false;
  }
}
template <typename T>
hipError_t launch_lognormal_t(int M, cplx<T>* W, long long nrows, double scale, const double* Ap, const double* Bp,
                              const cplx<T>* tw, double* partials, hipStream_t s, bool po) {
  // float64 rows of >= 512 complex: the exp table in the tile's spare LDS slots (rf_fft.h LognormalRowIO)
  switch (M) {
#define X(MM) case MM: { constexpr int SP = (sizeof(T) == 8 && MM >= 512) ? 1 : 0;                       \
    LognormalRowIO<T, SP> io; io.base = W; io.scale = (T)scale; io.M_of = M; io.Ap = Ap; io.Bp = Bp;                     \
    return launch_one<typename RowSel<T, MM>::type, LognormalRowIO<T, SP>>(io, nrows, tw, partials, s, po); }
    RF_ROW_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
template <typename T>
hipError_t launch_zscale_t(int M, cplx<T>* W, long long nrows, double scale, const double* Sz, const cplx<T>* tw, double* partials, hipStream_t s, bool po) {
  ScaleZRowIO<T> io; io.base = W; io.scale = (T)scale; io.M_of = M; io.Sz = Sz;
  switch (M) {
#define X(MM) case MM: return launch_one<typename RowSel<T, MM>::type, ScaleZRowIO<T>>(io, nrows, tw, partials, s, po);
    RF_ROW_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
template <class C>
hipError_t launch_fwd_one(const PlainRowFwdIO<typename C::T>& io, long long nrows, const cplx<typename C::T>* tw, hipStream_t s, bool po) {
  const long long ntiles = (nrows + C::NRT - 1) / C::NRT;
  auto k = row_r2c_kernel<C, PlainRowFwdIO<typename C::T>>;
  constexpr int lds = C::LDS_BYTES > 64 ? C::LDS_BYTES : 64;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds); e != hipSuccess) return e;
  if (po) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), lds, s, io, tw, nrows);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_fwd_t(int M, cplx<T>* W, long long nrows, const cplx<T>* tw, hipStream_t s, bool po) {
  PlainRowFwdIO<T> io; io.base = W; io.M_of = M;
  switch (M) {
#define X(MM) case MM: return launch_fwd_one<typename RowSel<T, MM>::type>(io, nrows, tw, s, po);
    RF_ROW_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
template <typename T> long long tiles_t(int M, long long nrows) {
  switch (M) {
#define X(MM) case MM: return (nrows + RowSel<T, MM>::type::NRT - 1) / RowSel<T, MM>::type::NRT;
    RF_ROW_SIZES(X)
#undef X
    default: return -1;
  }
}
}  // namespace

hipError_t launch_row_c2r(int f64, int M, void* W, long long nrows, double scale, const void* tw,
                          double* partials, hipStream_t s, bool po) {
  if (f64) return launch_t<double>(M, (cplx<double>*)W, nrows, scale, (const cplx<double>*)tw, partials, s, po);
  return launch_t<float>(M, (cplx<float>*)W, nrows, scale, (const cplx<float>*)tw, partials, s, po);
}
hipError_t launch_row_c2r_gather(int f64, int M, const void* src, void* dst, long long nrows, double scale, int nzl,
                                 long long seg_stride, const void* tw, double* partials, hipStream_t s, bool po) {
  if (f64) return launch_gather_t<double>(M, (const cplx<double>*)src, (cplx<double>*)dst, nrows, scale, nzl, seg_stride, (const cplx<double>*)tw, partials, s, po);
  return launch_gather_t<float>(M, (const cplx<float>*)src, (cplx<float>*)dst, nrows, scale, nzl, seg_stride, (const cplx<float>*)tw, partials, s, po);
}
hipError_t launch_row_c2r_xgather(int f64, int M, const void* src, void* dst, long long nrows, double scale, int tc, int rb, int ny,
                                  const void* tw, double* partials, hipStream_t s, bool po) {
  if (f64) return launch_xgather_t<double>(M, (const cplx<double>*)src, (cplx<double>*)dst, nrows, scale, tc, rb, ny, (const cplx<double>*)tw, partials, s, po);
  return launch_xgather_t<float>(M, (const cplx<float>*)src, (cplx<float>*)dst, nrows, scale, tc, rb, ny, (const cplx<float>*)tw, partials, s, po);
}
hipError_t launch_row_c2r_lognormal(int f64, int M, void* W, long long nrows, double scale, const double* Ap, const double* Bp,
                                    const void* tw, double* partials, hipStream_t s, bool po) {
  if (f64) return launch_lognormal_t<double>(M, (cplx<double>*)W, nrows, scale, Ap, Bp, (const cplx<double>*)tw, partials, s, po);
  return launch_lognormal_t<float>(M, (cplx<float>*)W, nrows, scale, Ap, Bp, (const cplx<float>*)tw, partials, s, po);
}
hipError_t launch_row_c2r_zscale(int f64, int M, void* W, long long nrows, double scale, const double* Sz, const void* tw, double* partials,
                                 hipStream_t s, bool po) {
  if (f64) return launch_zscale_t<double>(M, (cplx<double>*)W, nrows, scale, Sz, (const cplx<double>*)tw, partials, s, po);
  return launch_zscale_t<float>(M, (cplx<float>*)W, nrows, scale, Sz, (const cplx<float>*)tw, partials, s, po);
}
bool row_c2r_xgather_ok(int f64, int M, int tc, int rb) { return f64 ? xgather_ok_t<double>(M, tc, rb) : xgather_ok_t<float>(M, tc, rb); }
hipError_t launch_row_r2c(int f64, int M, void* W, long long nrows, const void* tw, hipStream_t s, bool po) {
  if (f64) return launch_fwd_t<double>(M, (cplx<double>*)W, nrows, (const cplx<double>*)tw, s, po);
  return launch_fwd_t<float>(M, (cplx<float>*)W, nrows, (const cplx<float>*)tw, s, po);
}
long long row_c2r_tiles(int f64, int M, long long nrows) {
  return f64 ? tiles_t<double>(M, nrows) : tiles_t<float>(M, nrows);
}
}  // namespace rf
