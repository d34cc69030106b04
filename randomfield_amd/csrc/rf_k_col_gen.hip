// x pass of the c2r transform with generation (rows K,T,R,S) fused into its load
#include "rf_kernels.h"
#include "rf_launch.h"

namespace rf {
namespace {
template <class C, class IO>
hipError_t launch_one(const IO& io, long long ncols, const cplx<typename C::T>* tw, hipStream_t s, bool prepare_only) {
  if (ncols % C::TC) return hipErrorInvalidValue;
  const long long ntiles = ncols / C::TC;
  auto k = col_kernel<C, +1, IO>;
  static bool prepared = false;
  if (!prepared) {
    if (C::LDS_BYTES > 65536) {
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
      if (e != hipSuccess) return e;
    }
    prepared = true;
  }
  if (prepare_only) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), C::LDS_BYTES, s, io, tw, ntiles);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_t(int N, cplx<T>* W, ColGeom g, long long ncols, const GenParams& gp, const cplx<T>* kspace,
                    int kz0, int nzl, const cplx<T>* tw, hipStream_t s, bool po) {
  GenColIO<T> io; io.base = W; io.g = g; io.gp = gp; io.kspace = kspace; io.kz0 = kz0; io.nzl = nzl;
  switch (N) {
#define X(NN) case NN: return launch_one<typename ColSel<T, NN>::type, GenColIO<T>>(io, ncols, tw, s, po);
    RF_COL_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t launch_col_fastgen(int N, void* W, ColGeom g, long long ncols, const FastGenParams& gp, int kz0, int nzl,
                              const void* tw, hipStream_t s, bool po) {
  FastGenColIO io; io.base = (cplx<float>*)W; io.g = g; io.gp = gp; io.kz0 = kz0; io.nzl = nzl;
  switch (N) {
#define X(NN) case NN: return launch_one<typename ColSel<float, NN>::type, FastGenColIO>(io, ncols, (const cplx<float>*)tw, s, po);
    RF_COL_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_col_gen(int f64, int N, void* W, ColGeom g, long long ncols, const GenParams& gp,
                          const void* kspace, int kz0, int nzl, const void* tw, hipStream_t s, bool po) {
  if (f64) return launch_t<double>(N, (cplx<double>*)W, g, ncols, gp, (const cplx<double>*)kspace, kz0, nzl, (const cplx<double>*)tw, s, po);
  return launch_t<float>(N, (cplx<float>*)W, g, ncols, gp, (const cplx<float>*)kspace, kz0, nzl, (const cplx<float>*)tw, s, po);
}
}  // namespace rf
