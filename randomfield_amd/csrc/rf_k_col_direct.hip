// rf_k_col_direct.hip -- the inverse y pass of a kz-slab rank with its output stored straight into the receive buffers of the
// exchange (rf_fft.h DirectColIO / Pair2DirectColIO; DESIGN.md section 5, "direct" mode).  Same configurations and the same
// arithmetic per tile as the in-place pass (rf_k_col_plain.hip launch_col_plain): only the base pointer of the stores differs,
// chosen once per workgroup from a device table of per-destination pointers.
#include "rf_kernels.h"
#include "rf_launch.h"

namespace rf {
namespace {

// Tile order: xcd_tile gives XCD j the j-th eighth of the tiles, i.e. of the x planes -- with 8 ranks XCD j stores to rank j, and all
// 8 destinations (7 links + the local segment) are being written at any one time.
template <class C, class IO>
__global__ __launch_bounds__(C::NT, (col_min_waves<C, IO>())) void col_direct_kernel(IO io, const cplx<typename C::T>* __restrict__ tw, long long ntiles) {
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  const long long tile = xcd_tile(blockIdx.x, ntiles);
  io.bind_tile(tile * C::TC);
  col_body<C, +1, IO>(io, tw, tile, rf_smem);
}

template <class C1, class IO>
__global__ __launch_bounds__(C1::NT, (col_min_waves<C1, IO>())) void col2_direct_kernel(IO io, const cplx<typename C1::T>* __restrict__ tw2, long long ntiles) {
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  const long long tile = xcd_tile(blockIdx.x, ntiles);
  io.bind_tile(tile * C1::TC);
  col2_body<C1, +1, IO>(io, tw2, tile, rf_smem);
}

template <class C, class IO>
hipError_t launch_one(const IO& io, long long ncols, const cplx<typename C::T>* tw, hipStream_t s, bool po) {
  // a tile must lie inside ONE x plane (one destination): whole tiles per run of `inner` columns
  if (!po && (ncols % C::TC || io.g.inner <= 0 || (io.g.inner & (io.g.inner - 1)) || io.g.inner % C::TC)) return hipErrorInvalidValue;
  const long long ntiles = ncols / C::TC;
  auto k = col_direct_kernel<C, IO>;
  constexpr int lds_bytes = C::LDS_BYTES + IO::LDS_EXTRA;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds_bytes); e != hipSuccess) return e;
  if (po) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), lds_bytes, s, io, tw, ntiles);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_t(int N, const cplx<T>* src, ColGeom g, cplx<T>* const* tab, int dest_shift, long long ncols, const cplx<T>* tw, hipStream_t s, bool po) {
  switch (N) {
#define X(NN)                                                                                                    \
  case NN: {                                                                                                     \
    using C = typename ColSel<T, NN>::type;                                                                      \
    if constexpr (NN >= 1024) {                                                                                  \
      if (po || g.needs_wide(C::LMAX, C::TC, (int)sizeof(cplx<T>))) {                                            \
        DirectColIO<T, true> iow; iow.base = const_cast<cplx<T>*>(src); iow.g = g; iow.tab = tab; iow.dest_shift = dest_shift; \
        hipError_t e = launch_one<C, DirectColIO<T, true>>(iow, ncols, tw, s, po);                               \
        if (!po || e != hipSuccess) return e;                                                                    \
      }                                                                                                          \
    } else if (!po && g.needs_wide(C::LMAX, C::TC, (int)sizeof(cplx<T>))) {                                      \
      return hipErrorInvalidValue;                                                                               \
    }                                                                                                            \
    DirectColIO<T> io; io.base = const_cast<cplx<T>*>(src); io.g = g; io.tab = tab; io.dest_shift = dest_shift;  \
    return launch_one<C, DirectColIO<T>>(io, ncols, tw, s, po);                                                  \
  }
    RF_COL_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

template <class C1>
hipError_t launch_pair(const cplx<float>* src, ColGeom g, cplx<float>* const* tab, int dest_shift, long long ncols, const cplx<float>* tw2, hipStream_t s, bool po) {
  using IO = Pair2DirectColIO<float>;
  if (!po && (ncols % C1::TC || g.inner <= 0 || (g.inner & (g.inner - 1)) || g.inner % C1::TC)) return hipErrorInvalidValue;
  IO io; io.base = const_cast<cplx<float>*>(src); io.g = g; io.gin = g; io.gin.row_stride = 2 * g.row_stride; io.par_off = g.row_stride;
  io.tab = tab; io.dest_shift = dest_shift;
  const long long ntiles = ncols / C1::TC;
  auto k = col2_direct_kernel<C1, IO>;
  constexpr int lds_bytes = C1::LDS_BYTES + IO::LDS_EXTRA;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds_bytes); e != hipSuccess) return e;
  if (po) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C1::NT), lds_bytes, s, io, tw2, ntiles);
  return hipGetLastError();
}

template <class C1> bool pair_fits(ColGeom g) {
  ColGeom gin = g;
  gin.row_stride = 2 * g.row_stride;
  return !gin.needs_wide(C1::LMAX, C1::TC, 8) && !g.needs_wide(C1::LMAX, C1::TC, 8) && g.row_shift >= 30 && g.hi_shift >= 62 && g.sub_shift == 0;
}
}  // namespace

// can the pass of length N store tile by tile to per-x-plane destinations?  (a tile must not straddle two x planes)
bool col_direct_supported(int f64, int N, long long inner) {
  const int tc = col_tile_cols(f64, N);
  return tc > 0 && inner > 0 && (inner & (inner - 1)) == 0 && inner % tc == 0;
}

hipError_t launch_col_direct(int f64, int N, const void* src, ColGeom g, void* const* tab, int dest_shift, long long ncols, const void* tw,
                             hipStream_t s, bool po) {
  if (!po && !col_direct_supported(f64, N, g.inner)) return hipErrorInvalidValue;
  // the same choice of kernel as launch_col_plain makes for the in-place pass, so that the arithmetic per tile is the same
  if (RF_COL2_2048 && N == 2048 && !f64) {
    using C1 = GenSel<float, 1024>::type;
    if (po || pair_fits<C1>(g)) {
      hipError_t e = launch_pair<C1>((const cplx<float>*)src, g, (cplx<float>* const*)tab, dest_shift, ncols, (const cplx<float>*)tw, s, po);
      if (!po || e != hipSuccess) return e;
    }
  }
  if (RF_Y_COL2_1024 && N == 1024 && !f64) {
    using C1 = PairSel1024::type;
    if (po || pair_fits<C1>(g)) {
      hipError_t e = launch_pair<C1>((const cplx<float>*)src, g, (cplx<float>* const*)tab, dest_shift, ncols, (const cplx<float>*)tw, s, po);
      if (!po || e != hipSuccess) return e;
    }
  }
  if (f64) return launch_t<double>(N, (const cplx<double>*)src, g, (cplx<double>* const*)tab, dest_shift, ncols, (const cplx<double>*)tw, s, po);
  return launch_t<float>(N, (const cplx<float>*)src, g, (cplx<float>* const*)tab, dest_shift, ncols, (const cplx<float>*)tw, s, po);
}
}  // namespace rf
