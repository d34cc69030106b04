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

// the y half's tile body: the in-place pass as two half-length transforms per tile (Col2)
template <class C1> struct YHalves {
  using IO = Pair2ColIO<float>;
  static constexpr int NT = C1::NT, TC = C1::TC, LMAX = C1::LMAX, LDS = C1::LDS_BYTES + IO::LDS_EXTRA;
  static IO make(void* W, ColGeom g) { IO io; io.base = (cplx<float>*)W; io.g = g; io.gin = g; io.gin.row_stride = 2 * g.row_stride; io.par_off = g.row_stride; return io; }
  static bool addressable(ColGeom g) { ColGeom gin = g; gin.row_stride = 2 * g.row_stride; return !gin.needs_wide(LMAX, TC, 8) && !g.needs_wide(LMAX, TC, 8); }
  __device__ static void run(IO& io, const cplx<float>* tw, long long tile, char* smem) { col2_body<C1, +1, IO>(io, tw, tile, smem); }
};

// One grid, two kinds of workgroup.  The launch has max(z threads, y threads) threads per workgroup; the surplus waves of the narrower
// kind leave at once (whole waves: a workgroup barrier only counts the waves that are still alive), so a 256-thread z tile and a
// 512-thread y tile can share a launch.  LDS = the larger of the two footprints.
template <class CR, class RIO, class YB>
__global__ __launch_bounds__((CR::NT > YB::NT ? CR::NT : YB::NT)) void yz_merged_kernel(RIO rio, const cplx<typename CR::T>* __restrict__ twz, long long nrows,
                                                           double* __restrict__ partials, unsigned nz_tiles, typename YB::IO cio,
                                                           const cplx<typename CR::T>* __restrict__ tw2, long long ny_tiles) {
  static_assert(CR::NT % 64 == 0 && YB::NT % 64 == 0, "whole waves of either kind");
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  if (blockIdx.x < nz_tiles) {
    if (CR::NT < YB::NT && threadIdx.x >= CR::NT) return;
    const long long tile = (long long)nz_tiles - 1 - blockIdx.x;          // (rows last to first, as row_c2r_kernel)
    row_c2r_body<CR, RIO>(rio, twz, nrows, partials, tile, rf_smem);
  } else {
    if (YB::NT < CR::NT && threadIdx.x >= YB::NT) return;
    // (nz_tiles is a multiple of 8, so workgroup b of the y half still lands on XCD b mod 8: xcd_tile's assumption)
    YB::run(cio, tw2, xcd_tile((long long)blockIdx.x - nz_tiles, ny_tiles), rf_smem);
  }
}

template <class CR, class YB>
bool fits(ColGeom gy, long long nrows, long long ncols) {
  if (!YB::addressable(gy) || gy.row_shift < 30 || gy.hi_shift < 62 || gy.sub_shift != 0) return false;
  const long long nz_tiles = (nrows + CR::NRT - 1) / CR::NRT;
  return ncols % YB::TC == 0 && nz_tiles % 8 == 0 && nz_tiles + ncols / YB::TC <= 0x7fffffffLL && gy.inner > 0 && (gy.inner & (gy.inner - 1)) == 0;
}

template <class CR, class YB>
hipError_t launch_merged(void* Wz, long long nrows, double scale, const void* twz, double* partials, void* Wy, ColGeom gy, long long ncols,
                         const void* twy, hipStream_t s, bool po) {
  using T = typename CR::T;
  using RIO = PlainRowIO<T>;
  const long long nz_tiles = (nrows + CR::NRT - 1) / CR::NRT, ny_tiles = ncols / YB::TC;
  if (!po && !fits<CR, YB>(gy, nrows, ncols)) return hipErrorInvalidValue;
  RIO rio; rio.base = (cplx<T>*)Wz; rio.scale = (T)scale; rio.M_of = CR::M;
  typename YB::IO cio = YB::make(Wy, gy);
  auto k = yz_merged_kernel<CR, RIO, YB>;
  constexpr int lds_z = CR::LDS_BYTES, lds = lds_z > YB::LDS ? lds_z : YB::LDS, nt = CR::NT > YB::NT ? CR::NT : YB::NT;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds); e != hipSuccess) return e;
  if (po) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)(nz_tiles + ny_tiles)), dim3(nt), lds, s, rio, (const cplx<T>*)twz, nrows, partials, (unsigned)nz_tiles,
                     cio, (const cplx<T>*)twy, ny_tiles);
  return hipGetLastError();
}

// the (ny, nz / 2) pairs that are served, float32: the y pass's product kernel for that length next to the z pass's
template <int MM> using ZRows = typename RowSel<float, MM>::type;
using Y1024 = YHalves<PairSel1024::type>;
// Served: ny = 1024 next to rows of 512 complex (both 256 threads: the 1024^3 pipeline).  Round 6 built and measured the pairs with UNEQUAL
// workgroup sizes as well -- the 512-thread y pass of length 2048 (YHalves<GenSel<float, 1024>::type>) next to the 256-thread z pass of rows
// of 1024 complex, and the 512-thread whole-column y pass of length 512 (YWhole<ColSel<float, 512>::type>) next to rows of 256 -- bit-identical
// fields, but SLOWER than one launch per pass: 2048^3 35.1 -> 35.9 ms, 512^3 0.485 -> 0.490 ms per graph-replayed realisation (MI355X,
// profiles/r06_ab/r06_b_merge.log).  One launch has one LDS size: the z tiles take the y tile's 72 KB and two of them share a CU where
// the z pass alone has three (45 KB) -- that costs the z half more than the overlapped ramp and drain return.  Not instantiated.
#define RF_YZ_PAIRS(X) X(1024, 512, Y1024)
}  // namespace

// which (ny, nz / 2) the merged launch serves: float32, the in-place y pass of length 1024 as two 512-point halves (256 threads) next to
// the 256-thread z pass of rows of 512 complex -- the 1024^3 pipeline
bool yz_merged_supported(int f64, int ny, int M) {
  if (f64 || !RF_Y_COL2_1024 || !RF_COL2_2048) return false;
#define X(NY, MM, YB) if (ny == NY && M == MM) return true;
  RF_YZ_PAIRS(X)
#undef X
  return false;
}

// ... and this geometry of the y pass (32-bit lane offsets, plain column layout), this many rows and columns per launch?  (queue_yz asks before
// it commits to the merged sequence; a plan it does not fit keeps one launch per pass)
bool yz_merged_fits(int f64, int ny, int M, ColGeom gy, long long nrows, long long ncols) {
  if (!yz_merged_supported(f64, ny, M)) return false;
#define X(NY, MM, YB) if (ny == NY && M == MM) return fits<ZRows<MM>, YB>(gy, nrows, ncols);
  RF_YZ_PAIRS(X)
#undef X
  return false;
}

hipError_t launch_yz_merged(int f64, int ny, int M, void* Wz, long long nrows, double scale, const void* twz, double* partials, void* Wy, ColGeom gy,
                            long long ncols, const void* twy, hipStream_t s, bool po) {
  if (!yz_merged_supported(f64, ny, M)) return hipErrorInvalidValue;
#define X(NY, MM, YB) if (ny == NY && M == MM) return launch_merged<ZRows<MM>, YB>(Wz, nrows, scale, twz, partials, Wy, gy, ncols, twy, s, po);
  RF_YZ_PAIRS(X)
#undef X
  return hipErrorInvalidValue;
}
}  // namespace rf
