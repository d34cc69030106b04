// strided in-place FFT passes (PlainColIO), float32 + float64, both directions
#include "rf_kernels.h"
#include "rf_launch.h"

namespace rf {
namespace {
template <class C, int DIR, class IO>
hipError_t launch_one(const IO& io_in, long long ncols, const cplx<typename C::T>* tw, hipStream_t s, bool prepare_only) {
  if (ncols % C::TC || io_in.g.inner <= 0 || (io_in.g.inner & (io_in.g.inner - 1))) return hipErrorInvalidValue;
  const IO& io = io_in;
  const long long ntiles = ncols / C::TC;
  auto k = col_kernel<C, DIR, IO>;
  constexpr int lds_bytes = C::LDS_BYTES + IO::LDS_EXTRA;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds_bytes); e != hipSuccess) return e;
  if (prepare_only) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), lds_bytes, s, io, tw, ntiles, 1LL, 0LL, 0);
  return hipGetLastError();
}
template <typename T, int DIR>
hipError_t launch_t(int N, cplx<T>* base, ColGeom g, long long ncols, const cplx<T>* tw, hipStream_t s, bool po) {
  PlainColIO<T> io; io.base = base; io.g = g;
  switch (N) {
#define X(NN)                                                                                                    \
  case NN: {                                                                                                     \
    using C = typename ColSel<T, NN>::type;                                                                      \
    if constexpr (NN >= 1024) {   /* 64-bit lane offsets: the long axes of the largest (unpacked c2c, float64) arrays */ \
      if (po || g.needs_wide(C::LMAX, C::TC, (int)sizeof(cplx<T>))) {                                            \
        PlainColIO<T, true> iow; iow.base = base; iow.g = g;                                                     \
        hipError_t e = launch_one<C, DIR, PlainColIO<T, true>>(iow, ncols, tw, s, po);                           \
        if (!po || e != hipSuccess) return e;                                                                    \
      }                                                                                                          \
    } else if (g.needs_wide(C::LMAX, C::TC, (int)sizeof(cplx<T>))) {                                             \
      return hipErrorInvalidValue;                                                                               \
    }                                                                                                            \
    return launch_one<C, DIR, PlainColIO<T>>(io, ncols, tw, s, po);                                              \
  }
    RF_COL_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
// a pass of length 2 C1::N in place as two C1 transforms per tile (rf_fft.h Col2 / Pair2ColIO); tw2 = the 2 C1::N-point table
template <class C1, int DIR>
hipError_t launch_pair(cplx<typename C1::T>* base, ColGeom g, long long ncols, const cplx<typename C1::T>* tw2, hipStream_t s, bool po) {
  using T = typename C1::T;
  using IO = Pair2ColIO<T>;
  if (ncols % C1::TC || g.inner <= 0 || (g.inner & (g.inner - 1))) return hipErrorInvalidValue;
  IO io; io.base = base; io.g = g; io.gin = g; io.gin.row_stride = 2 * g.row_stride; io.par_off = g.row_stride;
  const long long ntiles = ncols / C1::TC;
  auto k = col2_kernel<C1, DIR, IO>;
  constexpr int lds_bytes = C1::LDS_BYTES + IO::LDS_EXTRA;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds_bytes); e != hipSuccess) return e;
  if (po) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C1::NT), lds_bytes, s, io, tw2, ntiles, 1LL, 0LL, 0);
  return hipGetLastError();
}

// inverse pass out of place: src (geometry gs) -> dst (geometry gd)
template <typename T>
hipError_t launch_xp(int N, const cplx<T>* src, ColGeom gs, cplx<T>* dst, ColGeom gd, long long ncols, const cplx<T>* tw, hipStream_t s, bool po) {
  XposeColIO<T> io; io.src = src; io.gs = gs; io.base = dst; io.g = gd;
  if (gd.inner <= 0 || ncols % gd.inner) return hipErrorInvalidValue;
  const long long nhi = ncols / gd.inner;                 // values of the slow index (ix); gd.inner = kz planes per run
  if (nhi & (nhi - 1)) return hipErrorInvalidValue;
  switch (N) {
#define X(NN)                                                                                                    \
  case NN: {                                                                                                     \
    using C = typename ColSel<T, NN>::type;                                                                      \
    if (!po && (gs.needs_wide(C::LMAX, C::TC, (int)sizeof(cplx<T>)) || gd.needs_wide(C::LMAX, C::TC, (int)sizeof(cplx<T>)) || \
                gs.inner <= 0 || (gs.inner & (gs.inner - 1)) || (gs.sub_shift > 0 && (1 << gs.sub_shift) < C::TC)))  \
      return hipErrorInvalidValue;                                                                               \
    if (gd.inner % C::TC) return po ? hipSuccess : hipErrorInvalidValue;                                        \
    set_xpose_order(io, nhi, gd.inner / C::TC);                                                                  \
    return launch_one<C, +1, XposeColIO<T>>(io, ncols, tw, s, po);                                               \
  }
    RF_COL_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
// inverse pass in place + the Parseval partials (AccColIO)
template <typename T>
hipError_t launch_acc_t(int N, cplx<T>* base, ColGeom g, long long ncols, int kz0, int nzl, double* partials, const cplx<T>* tw, hipStream_t s, bool po) {
  AccColIO<T> io; io.base = base; io.g = g; io.partials = partials; io.kz0 = kz0; io.nzl = nzl;
  if (!po && (nzl <= 0 || (nzl & (nzl - 1)))) return hipErrorInvalidValue;
  switch (N) {
#define X(NN)                                                                                                    \
  case NN: {                                                                                                     \
    using C = typename ColSel<T, NN>::type;                                                                      \
    if (!po && g.needs_wide(C::LMAX, C::TC, (int)sizeof(cplx<T>))) return hipErrorInvalidValue;                  \
    return launch_one<C, +1, AccColIO<T>>(io, ncols, tw, s, po);                                                 \
  }
    RF_COL_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t launch_col_plain_acc(int f64, int N, void* base, ColGeom g, long long ncols, int kz0, int nzl, double* partials, const void* tw,
                                hipStream_t s, bool po) {
  if (f64) return launch_acc_t<double>(N, (cplx<double>*)base, g, ncols, kz0, nzl, partials, (const cplx<double>*)tw, s, po);
  return launch_acc_t<float>(N, (cplx<float>*)base, g, ncols, kz0, nzl, partials, (const cplx<float>*)tw, s, po);
}

hipError_t launch_col_xpose(int f64, int N, const void* src, ColGeom gs, void* dst, ColGeom gd, long long ncols, const void* tw,
                            hipStream_t s, bool po) {
  if (f64) return launch_xp<double>(N, (const cplx<double>*)src, gs, (cplx<double>*)dst, gd, ncols, (const cplx<double>*)tw, s, po);
  return launch_xp<float>(N, (const cplx<float>*)src, gs, (cplx<float>*)dst, gd, ncols, (const cplx<float>*)tw, s, po);
}

// can the strided pass of length N address this geometry?  (32-bit lane offsets everywhere; 64-bit ones exist for N >= 1024)
bool col_plain_addressable(int f64, int N, ColGeom g) {
  switch (N) {
#define X(NN) case NN: { const int L = f64 ? ColSel<double, NN>::type::LMAX : ColSel<float, NN>::type::LMAX,               \
                                   tc = f64 ? ColSel<double, NN>::type::TC : ColSel<float, NN>::type::TC;                    \
                         return NN >= 1024 || !g.needs_wide(L, tc, f64 ? 16 : 8); }
    RF_COL_SIZES(X)
#undef X
    default: return false;
  }
}

int col_tile_cols(int f64, int N) {
  switch (N) {
#define X(NN) case NN: return f64 ? ColSel<double, NN>::type::TC : ColSel<float, NN>::type::TC;
    RF_COL_SIZES(X)
#undef X
    default: return 0;
  }
}

hipError_t launch_col_plain(int f64, int N, int dir, void* base, ColGeom g, long long ncols, const void* tw,
                            hipStream_t s, bool po) {
  if (RF_COL2_2048 && N == 2048 && !f64) {
    // (the radix-8-first 1024-point configuration: with 32 parked registers the 16-first one would not fit 128 VGPRs)
    using C1 = GenSel<float, 1024>::type;
    ColGeom gin = g;
    gin.row_stride = 2 * g.row_stride;
    const bool fits = !gin.needs_wide(C1::LMAX, C1::TC, 8) && !g.needs_wide(C1::LMAX, C1::TC, 8) && g.row_shift >= 30 && g.hi_shift >= 62 && g.sub_shift == 0;
    if (po || fits) {
      hipError_t e = dir > 0 ? launch_pair<C1, +1>((cplx<float>*)base, g, ncols, (const cplx<float>*)tw, s, po)
                             : launch_pair<C1, -1>((cplx<float>*)base, g, ncols, (const cplx<float>*)tw, s, po);
      if (!po || e != hipSuccess) return e;
    }
  }
                                       // RF_Y_COL2_1024 (rf_configs.h): the float32 in-place pass of length 1024 as two 512-point transforms per 8-column tile (Col2): 256 threads, 36 KB of
                                       // LDS, 109 registers -- FOUR workgroups per CU where the whole-column kernel (72 KB) has two.  The pass is latency-bound,
                                       // not HBM-bound (its slab is read once from HBM and handed to the z pass through the Infinity Cache): more tiles in
                                       // flight per CU, y pass 1.70 -> 1.61 ms per 1024^3 (profiles/r04_ab/ycol2.log)
  if (RF_Y_COL2_1024 && N == 1024 && !f64) {
    using C1 = PairSel1024::type;
    ColGeom gin = g;
    gin.row_stride = 2 * g.row_stride;
    const bool fits = !gin.needs_wide(C1::LMAX, C1::TC, 8) && !g.needs_wide(C1::LMAX, C1::TC, 8) && g.row_shift >= 30 && g.hi_shift >= 62 && g.sub_shift == 0;
    if (po || fits) {
      hipError_t e = dir > 0 ? launch_pair<C1, +1>((cplx<float>*)base, g, ncols, (const cplx<float>*)tw, s, po)
                             : launch_pair<C1, -1>((cplx<float>*)base, g, ncols, (const cplx<float>*)tw, s, po);
      if (!po || e != hipSuccess) return e;
    }
  }
  if (f64) return dir > 0 ? launch_t<double, +1>(N, (cplx<double>*)base, g, ncols, (const cplx<double>*)tw, s, po)
                          : launch_t<double, -1>(N, (cplx<double>*)base, g, ncols, (const cplx<double>*)tw, s, po);
  return dir > 0 ? launch_t<float, +1>(N, (cplx<float>*)base, g, ncols, (const cplx<float>*)tw, s, po)
                 : launch_t<float, -1>(N, (cplx<float>*)base, g, ncols, (const cplx<float>*)tw, s, po);
}
}  // namespace rf
