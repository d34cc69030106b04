// contiguous-axis pass of the unpacked c2c transform (transform.py:207-213): plain complex rows, both directions
#include "rf_kernels.h"
#include "rf_launch.h"

namespace rf {
namespace {
template <class C, int DIR>
hipError_t launch_one(const ScaledRowIO<typename C::T>& io, long long nrows, const cplx<typename C::T>* tw, hipStream_t s, bool po) {
  const long long ntiles = (nrows + C::NRT - 1) / C::NRT;
  auto k = row_c2c_kernel<C, DIR, ScaledRowIO<typename C::T>>;
  constexpr int lds = C::LDS_BYTES > 64 ? C::LDS_BYTES : 64;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds); e != hipSuccess) return e;
  if (po) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), lds, s, io, tw, nrows);
  return hipGetLastError();
}
template <typename T, int DIR>
hipError_t launch_t(int M, cplx<T>* W, long long nrows, double scale, const cplx<T>* tw, hipStream_t s, bool po) {
  ScaledRowIO<T> io; io.base = W; io.M_of = M; io.scale = (T)scale;
  switch (M) {
#define X(MM) case MM: return launch_one<typename RowSel<T, MM>::type, DIR>(io, nrows, tw, s, po);
    RF_ROWC_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t launch_row_c2c(int f64, int M, int dir, void* W, long long nrows, double scale, const void* tw, hipStream_t s, bool po) {
  if (f64) return dir > 0 ? launch_t<double, +1>(M, (cplx<double>*)W, nrows, scale, (const cplx<double>*)tw, s, po)
                          : launch_t<double, -1>(M, (cplx<double>*)W, nrows, scale, (const cplx<double>*)tw, s, po);
  return dir > 0 ? launch_t<float, +1>(M, (cplx<float>*)W, nrows, scale, (const cplx<float>*)tw, s, po)
                 : launch_t<float, -1>(M, (cplx<float>*)W, nrows, scale, (const cplx<float>*)tw, s, po);
}
}  // namespace rf
