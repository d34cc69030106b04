// rf_col_gen_launch.h -- launch helpers shared by the two translation units of the generation pass (rf_k_col_gen.hip: float32,
// compiled with the max-ILP scheduling strategy; rf_k_col_gen64.hip: float64, default strategy -- measured on MI355X: the float32
// whole-column kernels gain 7 % from it, the float64 half-transform kernels lose 8 %)
#pragma once
#include "rf_kernels.h"
#include "rf_launch.h"

// (Launch structure of the kz = 0 repair, settled by measurement -- DESIGN_HISTORY.md: the tiles that hold slot kz = 0 run as a launch of
// their own in FRONT of the lean kernel's launch over all other tiles.  One launch of the repairing kernel over all tiles, the lean
// kernel over all tiles followed by the kz = 0 tiles again, and both kinds of tile in one grid were each measured and are gone.)

namespace rf {
namespace {
template <class IO, class = void> struct io_noise_src { static constexpr int value = 0; };
template <class IO> struct io_noise_src<IO, typename std::enable_if<(IO::NOISE_SRC >= 0)>::type> { static constexpr int value = IO::NOISE_SRC; };
// runs tiles  b * tile_mul + tile_add,  b in [0, ntiles)
template <class C, class IO>
hipError_t launch_one(const IO& io_in, long long ncols, const cplx<typename C::T>* tw, hipStream_t s, bool prepare_only,
                      long long ntiles_sub = -1, long long tile_mul = 1, long long tile_add = 0, int skip_period = 0) {
  if (ncols % C::TC || io_in.g.inner <= 0 || (io_in.g.inner & (io_in.g.inner - 1))) return hipErrorInvalidValue;
  if (!prepare_only && (!io_in.g.rows_ok(C::N / C::RL, C::NPASS) || (io_in.g.sub_shift > 0 && (1 << io_in.g.sub_shift) < C::TC))) return hipErrorInvalidValue;
  const IO& io = io_in;
  const long long ntiles = ntiles_sub >= 0 ? ntiles_sub : ncols / C::TC;
  auto k = col_kernel<C, +1, IO>;
  constexpr int lds_bytes = C::LDS_BYTES + IO::LDS_EXTRA;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds_bytes); e != hipSuccess) return e;
  if (prepare_only) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), lds_bytes, s, io, tw, ntiles, tile_mul, tile_add, skip_period);
  return hipGetLastError();
}

// pairs of adjacent tiles, whole-line stores (rf_kernels.h colpair_kernel): runs pairs b * pair_mul + pair_add, b in [0, npairs)
// (the float32 generation pass of length 1024 -- native generator or replayed deviates, whole grid or kz slab -- runs on tile pairs)
template <class C, class IO>
hipError_t launch_onepair(const IO& io_in, long long ncols, const cplx<typename C::T>* tw, hipStream_t s, bool prepare_only,
                          long long npairs, long long pair_mul = 1, long long pair_add = 0, int skip_period = 0) {
  if (ncols % (2 * C::TC) || io_in.g.inner <= 0 || (io_in.g.inner & (io_in.g.inner - 1))) return hipErrorInvalidValue;
  if (!prepare_only && (!io_in.g.rows_ok(C::N / C::RL, C::NPASS) || io_in.g.sub_shift > 0 || io_in.g.inner % (2 * C::TC))) return hipErrorInvalidValue;
  const IO& io = io_in;
  auto k = colpair_kernel<C, +1, IO>;
  constexpr int lds_bytes = C::LDS_BYTES + IO::LDS_EXTRA;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds_bytes); e != hipSuccess) return e;
  if (prepare_only) return hipSuccess;
  if (npairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)npairs), dim3(C::NT), lds_bytes, s, io, tw, npairs, pair_mul, pair_add, skip_period);
  return hipGetLastError();
}

// the same through col2_kernel: a pass of length 2 C1::N as two C1 transforms per tile (rf_fft.h Col2); tw2 = the 2 C1::N-point table
template <class C1, class IO>
hipError_t launch_one2(const IO& io_in, long long ncols, const cplx<typename C1::T>* tw2, hipStream_t s, bool prepare_only,
                       long long ntiles_sub = -1, long long tile_mul = 1, long long tile_add = 0, int skip_period = 0) {
  if (ncols % C1::TC || io_in.g.inner <= 0 || (io_in.g.inner & (io_in.g.inner - 1))) return hipErrorInvalidValue;
  const IO& io = io_in;
  const long long ntiles = ntiles_sub >= 0 ? ntiles_sub : ncols / C1::TC;
  auto k = col2_kernel<C1, +1, IO>;
  constexpr int lds_bytes = C1::LDS_BYTES + IO::LDS_EXTRA;
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)k, lds_bytes); e != hipSuccess) return e;
  if (prepare_only) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C1::NT), lds_bytes, s, io, tw2, ntiles, tile_mul, tile_add, skip_period);
  return hipGetLastError();
}

// the side buffer of repaired kz = 0 slots for a FIX = 3 launch: out[iy * nx + ix], one thread per mode (rf_kernels.h fix_fill_kernel)
template <class IOF, class CT>
hipError_t launch_fix_fill(const IOF& iof, CT* out, hipStream_t s) {
  const long long n = (long long)iof.gp.nx * iof.gp.ny;
  hipLaunchKernelGGL((fix_fill_kernel<IOF, CT>), dim3((unsigned)((n + 255) / 256)), dim3(256), IOF::LDS_EXTRA, s, iof, out, iof.gp.nx, iof.gp.ny);
  return hipGetLastError();
}

// fast float32 generation + x pass of length 2 C1::N through Col2 (native generator, whole grid or kz slab; no potential store,
// no resident deviates, no x-slab restriction: those keep the whole-column kernel)
template <class C1, class IO0, class IO1>
hipError_t launch_fast_one2(const FastGenParams& gp, cplx<typename C1::T>* W, ColGeom g, long long ncols, int kz0, int nzl,
                            const cplx<typename C1::T>* tw2, hipStream_t s, bool po, hipEvent_t after_repair, cplx<typename C1::T>* fixbuf) {
  if (nzl <= 0 || (nzl & (nzl - 1)) || ncols >= (1LL << 31) || g.needs_wide(C1::LMAX, C1::TC, (int)sizeof(cplx<typename C1::T>)) || !g.rows_ok(C1::N / C1::RL, C1::NPASS))
    return hipErrorInvalidValue;
  IO0 io0; io0.base = W; io0.g = g; io0.gp = gp; io0.kz0 = kz0; io0.nzl = nzl; io0.rec = nullptr; io0.pot = nullptr;
  IO1 io1; io1.base = W; io1.g = g; io1.gp = gp; io1.kz0 = kz0; io1.nzl = nzl; io1.rec = nullptr; io1.pot = nullptr;
  const bool split = nzl > C1::TC && nzl % C1::TC == 0;
  const long long tiles_per_iy = nzl / C1::TC, ntiles = ncols / C1::TC;
  // the split launch's repair kernel takes the repaired slots from the side buffer (FIX = 3, filled by fix_fill_kernel just before)
  using IOC = typename IO1::template with_fix<3>;
  using IOF = typename IO1::fill_io;
  IOC ioc; ioc.base = W; ioc.g = g; ioc.gp = gp; ioc.kz0 = kz0; ioc.nzl = nzl; ioc.rec = nullptr; ioc.pot = nullptr; ioc.fixbuf = fixbuf;
  if (po) {
    hipError_t e = launch_one2<C1, IO0>(io0, ncols, tw2, s, true);
    if (e == hipSuccess) e = launch_one2<C1, IOC>(ioc, ncols, tw2, s, true);
    return e != hipSuccess ? e : launch_one2<C1, IO1>(io1, ncols, tw2, s, true);
  }
  if (!split) return launch_one2<C1, IO1>(io1, ncols, tw2, s, false);
  if (kz0 != 0) return launch_one2<C1, IO0>(io0, ncols, tw2, s, false);
  if (!fixbuf) return hipErrorInvalidValue;
  IOF iof; iof.base = W; iof.g = g; iof.gp = gp; iof.kz0 = kz0; iof.nzl = nzl; iof.rec = nullptr; iof.pot = nullptr;
  hipError_t e = launch_fix_fill(iof, fixbuf, s);
  if (e == hipSuccess) e = launch_one2<C1, IOC>(ioc, ncols, tw2, s, false, ncols / nzl, tiles_per_iy, 0);
  if (e != hipSuccess || tiles_per_iy >= (1LL << 30) || ntiles >= (1LL << 31)) return e != hipSuccess ? e : hipErrorInvalidValue;
  if (after_repair && (e = hipEventRecord(after_repair, s)) != hipSuccess) return e;
  return launch_one2<C1, IO0>(io0, ncols, tw2, s, false, ntiles - ntiles / tiles_per_iy, 1, 0, (int)tiles_per_iy);
}

// Fast float32 generation.  The tiles that contain the kz = 0 slot (one per iy when a tile is narrower than a
// kz row) are run by the kernel WITH the Hermitian repair, which carries the extra register pressure only where
// it is needed; every other tile by the kernel WITHOUT it (skip_period = tiles per iy).
template <class C, class IO0, class IO1, class CT>
hipError_t launch_fast_one(const FastGenParams& gp, CT* W, ColGeom g, long long ncols, int kz0, int nzl,
                           const CT* tw, hipStream_t s, bool po, hipEvent_t after_repair, int x0, int x1, CT* fixbuf, CT* pot = nullptr) {
  // the slab-restricted instantiations test the workgroup-uniform row offset m * L of the last pass: the slab
  // boundaries must be multiples of L = N / (radix of the last pass)
  if (nzl <= 0 || (nzl & (nzl - 1)) || ncols >= (1LL << 31) || g.needs_wide(C::LMAX, C::TC, (int)sizeof(CT))) return hipErrorInvalidValue;   // the IO splits a column index by shift and mask
  if ((x0 > 0 || x1 < C::N) && (C::NPASS < 2 || x0 % (C::N / C::RL) || x1 % (C::N / C::RL))) return hipErrorInvalidValue;
  CT* base = x0 > 0 ? W - (long long)x0 * g.row_stride : W;      // row x0 of the transform lands on row 0 of W
  IO0 io0; io0.base = base; io0.g = g; io0.gp = gp; io0.kz0 = kz0; io0.nzl = nzl; io0.rec = nullptr; io0.x0 = x0; io0.x1 = x1;
  io0.pot = pot;
  IO1 io1; io1.base = base; io1.g = g; io1.gp = gp; io1.kz0 = kz0; io1.nzl = nzl; io1.rec = nullptr; io1.x0 = x0; io1.x1 = x1; io1.pot = pot;
  const bool split = nzl > C::TC && nzl % C::TC == 0;
  const long long tiles_per_iy = nzl / C::TC, ntiles = ncols / C::TC;
  // the split launch's repair kernel of the long passes takes the repaired slots from the side buffer (FIX = 3); the short passes
  // (a few tiles, one pass) keep the owning lane's own repair
  constexpr bool side = C::N >= 512 && C::NPASS >= 2;
  using IOC = typename IO1::template with_fix<side ? 3 : 1>;
  using IOF = typename IO1::fill_io;
  IOC ioc; ioc.base = base; ioc.g = g; ioc.gp = gp; ioc.kz0 = kz0; ioc.nzl = nzl; ioc.rec = nullptr; ioc.x0 = x0; ioc.x1 = x1; ioc.pot = pot;
  if constexpr (side) ioc.fixbuf = fixbuf;
  // tile pairs with whole-line stores (ColPair): the 1024-point float32 pass, every tile of a kz row in a pair of its own row
  constexpr bool pairs = side && C::N == 1024 && sizeof(CT) == 8 && C::NPASS == 3 && (io_noise_src<IO0>::value == 0 || io_noise_src<IO0>::value == 2);
  const bool use_pairs = pairs && split && x0 <= 0 && x1 >= C::N && tiles_per_iy % 2 == 0 && g.inner % (2 * C::TC) == 0 && g.sub_shift == 0;
  if (po) {
    hipError_t e = launch_one<C, IO0>(io0, ncols, tw, s, true);
    if constexpr (pairs) {
      if (e == hipSuccess) e = launch_onepair<C, IO0>(io0, ncols, tw, s, true, 0);
      if (e == hipSuccess) e = launch_onepair<C, IOC>(ioc, ncols, tw, s, true, 0);
    }
    if (e == hipSuccess && side) e = launch_one<C, IOC>(ioc, ncols, tw, s, true);
    return e != hipSuccess ? e : launch_one<C, IO1>(io1, ncols, tw, s, true);
  }
  if (!split) return launch_one<C, IO1>(io1, ncols, tw, s, false);
  if (kz0 != 0) {                                                       // only the slab that owns kz = 0 needs the repair
    if constexpr (pairs)
      if (use_pairs && ntiles / 2 < (1LL << 31)) return launch_onepair<C, IO0>(io0, ncols, tw, s, false, ntiles / 2);
    return launch_one<C, IO0>(io0, ncols, tw, s, false);
  }
  // first the (few) tiles that hold slot kz = 0, with the repair; then every other tile without it
  hipError_t e = hipSuccess;
  if constexpr (side) {
    if (!fixbuf) return hipErrorInvalidValue;
    IOF iof; iof.base = base; iof.g = g; iof.gp = gp; iof.kz0 = kz0; iof.nzl = nzl; iof.rec = nullptr; iof.x0 = x0; iof.x1 = x1; iof.pot = pot;
    e = launch_fix_fill(iof, fixbuf, s);
  }
  if constexpr (pairs) {
    if (use_pairs) {
      // the pair that holds the kz = 0 tile of every ky row (repair from the side buffer: only that tile's owning lanes take it), then all others
      const long long ppi = tiles_per_iy / 2, npairs = ntiles / 2;
      if (ppi >= (1LL << 30) || npairs >= (1LL << 31)) return hipErrorInvalidValue;
      if (e == hipSuccess) e = launch_onepair<C, IOC>(ioc, ncols, tw, s, false, ncols / nzl, ppi, 0);
      if (e != hipSuccess) return e;
      if (after_repair && (e = hipEventRecord(after_repair, s)) != hipSuccess) return e;
      if (ppi < 2) return hipSuccess;                    // (two tiles per ky row: the repair launch has covered everything)
      return launch_onepair<C, IO0>(io0, ncols, tw, s, false, npairs - npairs / ppi, 1, 0, (int)ppi);
    }
  }
  if (e == hipSuccess) e = launch_one<C, IOC>(ioc, ncols, tw, s, false, ncols / nzl, tiles_per_iy, 0);
  if (e != hipSuccess || tiles_per_iy >= (1LL << 30) || ntiles >= (1LL << 31)) return e != hipSuccess ? e : hipErrorInvalidValue;
  if (after_repair && (e = hipEventRecord(after_repair, s)) != hipSuccess) return e;
  return launch_one<C, IO0>(io0, ncols, tw, s, false, ntiles - ntiles / tiles_per_iy, 1, 0, (int)tiles_per_iy);
}

}  // namespace

// the float64 branches of launch_col_fastgen (rf_k_col_gen64.hip)
hipError_t launch_col_fastgen64(int N, void* W, ColGeom g, long long ncols, const FastGenParams& gp, int kz0, int nzl,
                                const void* tw, hipStream_t s, bool po, hipEvent_t after_repair, int x0, int x1, void* pot, void* fixbuf);
}  // namespace rf
