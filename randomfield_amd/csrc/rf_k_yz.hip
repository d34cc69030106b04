// rf_k_yz.hip -- the z pass of slab s and the y pass of slab s + 1 in ONE launch.
//
// A single-GPU realisation runs its y and z passes slab by slab (64 x planes = 256 MiB at 1024^3: the slab goes from the y pass to the
// z pass through the Infinity Cache), i.e. 16 + 16 launches of 0.08 - 0.1 ms, and every launch pays its own ramp
// and drain: the CUs idle while the last tiles of a launch finish (a tile takes ~25 us from its first load to its last store, a launch
// ~4 rounds of them).  The z pass of slab s depends on the y pass of slab s -- a kernel boundary -- but the y pass of slab s + 1 depends
// on neither.  So the tiles of both go into one grid, the z tiles first: the hardware hands out workgroups in order, and the y tiles of
// the next slab start on the CUs the draining z pass leaves free.  No flags, no spinning: the two halves are independent.
// Both bodies are the product kernels' own (rf_kernels.h row_c2r_body / col2_body): same arithmetic, same field bit for bit.
#include "rf_kernels.h"
#include "rf_launch.h"

namespace rf {
namespace {

template <class CR, class RIO, class C1, class CIO>
__global__ __launch_bounds__(CR::NT) void yz_merged_kernel(RIO rio, const cplx<typename CR::T>* __restrict__ twz, long long nrows,
                                                           double* __restrict__ partials, unsigned nz_tiles, CIO cio,
                                                           const cplx<typename CR::T>* __restrict__ tw2, long long ny_tiles) {
  static_assert(CR::NT == C1::NT, "one workgroup size for both kinds of tile");
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  if (blockIdx.x < nz_tiles) {
    const long long tile = RF_Z_REVERSE ? (long long)nz_tiles - 1 - blockIdx.x : (long long)blockIdx.x;
    row_c2r_body<CR, RIO>(rio, twz, nrows, partials, tile, rf_smem);
  } else {
    // (nz_tiles is a multiple of 8, so workgroup b of the y half still lands on XCD b mod 8: xcd_tile's assumption)
    col2_body<C1, +1, CIO>(cio, tw2, xcd_tile((long long)blockIdx.x - nz_tiles, ny_tiles), rf_smem);
  }
}

template <class CR, class C1>
hipError_t launch_merged(void* Wz, long long nrows, double scale, const void* twz, double* partials, void* Wy, ColGeom gy, long long ncols,
                         const void* twy, hipStream_t s, bool po) {
  using T = typename CR::T;
  using RIO = PlainRowIO<T>;
  using CIO = Pair2ColIO<T>;
  const long long nz_tiles = (nrows + CR::NRT - 1) / CR::NRT, ny_tiles = ncols / C1::TC;
  if (!po && (ncols % C1::TC || nz_tiles % 8 || nz_tiles + ny_tiles > 0x7fffffffLL || gy.inner <= 0 || (gy.inner & (gy.inner - 1)))) return hipErrorInvalidValue;
  RIO rio; rio.base = (cplx<T>*)Wz; rio.scale = (T)scale; rio.M_of = CR::M;
  CIO cio; cio.base = (cplx<T>*)Wy; cio.g = gy; cio.gin = gy; cio.gin.row_stride = 2 * gy.row_stride; cio.par_off = gy.row_stride;
  auto k = yz_merged_kernel<CR, RIO, C1, CIO>;
  constexpr int lds_z = CR::LDS_BYTES, lds_y = C1::LDS_BYTES + CIO::LDS_EXTRA, lds = lds_z > lds_y ? lds_z : lds_y;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds); e != hipSuccess) return e;
  if (po) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)(nz_tiles + ny_tiles)), dim3(CR::NT), lds, s, rio, (const cplx<T>*)twz, nrows, partials, (unsigned)nz_tiles,
                     cio, (const cplx<T>*)twy, ny_tiles);
  return hipGetLastError();
}
}  // namespace

// which (ny, nz / 2) the merged launch serves: float32, the in-place y pass of length 1024 as two 512-point halves (256 threads) next to
// the 256-thread z pass of rows of 512 complex -- the 1024^3 pipeline
bool yz_merged_supported(int f64, int ny, int M) { return !f64 && RF_Y_COL2_1024 && ny == 1024 && M == 512; }

// ... and this geometry of the y pass (32-bit lane offsets, plain column layout), this many rows and columns per launch?  (queue_yz asks before
// it commits to the merged sequence; a plan it does not fit keeps one launch per pass)
bool yz_merged_fits(int f64, int ny, int M, ColGeom gy, long long nrows, long long ncols) {
  if (!yz_merged_supported(f64, ny, M)) return false;
  using C1 = PairSel1024::type;
  using CR = RowSel<float, 512>::type;
  ColGeom gin = gy;
  gin.row_stride = 2 * gy.row_stride;
  if (gin.needs_wide(C1::LMAX, C1::TC, 8) || gy.needs_wide(C1::LMAX, C1::TC, 8) || gy.row_shift < 30 || gy.hi_shift < 62 || gy.sub_shift != 0) return false;
  const long long nz_tiles = (nrows + CR::NRT - 1) / CR::NRT;
  return ncols % C1::TC == 0 && nz_tiles % 8 == 0 && nz_tiles + ncols / C1::TC <= 0x7fffffffLL && gy.inner > 0 && (gy.inner & (gy.inner - 1)) == 0;
}

hipError_t launch_yz_merged(int f64, int ny, int M, void* Wz, long long nrows, double scale, const void* twz, double* partials, void* Wy, ColGeom gy,
                            long long ncols, const void* twy, hipStream_t s, bool po) {
  if (!yz_merged_supported(f64, ny, M)) return hipErrorInvalidValue;
  if (!po && !yz_merged_fits(f64, ny, M, gy, nrows, ncols)) return hipErrorInvalidValue;
  return launch_merged<RowSel<float, 512>::type, PairSel1024::type>(Wz, nrows, scale, twz, partials, Wy, gy, ncols, twy, s, po);
}
}  // namespace rf
