// rf_launch.h -- host-callable launchers; each .hip translation unit instantiates
// one family of kernels so that the files compile in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include "rf_configs.h"
#include "rf_generic.h"

namespace rf {

// Kernels with more than 64 KB of dynamic LDS need hipFuncSetAttribute(MaxDynamicSharedMemorySize) once PER DEVICE:
// every launcher keeps one of these per kernel instantiation (bit d = device d is prepared; thread safe).
struct LdsAttrLatch {
  std::atomic<unsigned long long> done{0};
  hipError_t ensure(const void* kernel, int lds_bytes) {
    if (lds_bytes <= 65536) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

// compute units of the current device (cached per device): persistent kernels launch a multiple of it
inline int device_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int v = cache[dev & 63].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  cache[dev & 63].store(v, std::memory_order_relaxed);
  return v;
}

// strided FFT pass in place (y pass of c2r; x/y passes of r2c). dir = +1 inverse, -1 forward.
// prepare_only = true sets the kernel's function attributes (dynamic LDS > 64 KB) without launching
hipError_t launch_col_plain(int f64, int N, int dir, void* base, ColGeom g, long long ncols, const void* tw,
                            hipStream_t s, bool prepare_only = false);
// inverse strided pass out of place: reads src through geometry gs, writes dst through gd (the y pass of the c2r transform
// when the x pass left a transposed intermediate: DESIGN.md section 3.8)
hipError_t launch_col_xpose(int f64, int N, const void* src, ColGeom gs, void* dst, ColGeom gd, long long ncols, const void* tw,
                            hipStream_t s, bool prepare_only = false);
// the inverse pass out of place with every output tile stored through tab[ix >> dest_shift] instead of `src`'s own base (device table of
// per-destination base pointers; ix = the tile's index of the slow column axis: rf_fft.h DirectColIO) -- the y pass of a kz-slab rank
// writing straight into the receive buffers of the exchange
bool col_direct_supported(int f64, int N, long long inner);
hipError_t launch_col_direct(int f64, int N, const void* src, ColGeom g, void* const* tab, int dest_shift, long long ncols, const void* tw,
                             hipStream_t s, bool prepare_only = false);
int col_gen_tile_cols(int f64, int N);   // tile width of the generation-fused x pass of length N
int col_gen_row_block(int f64, int N, int want);   // x rows per block of the transposed intermediate: `want`, or N when the pass cannot block
// x pass of c2r fused with generation (kspace == nullptr) or reading an API-layout k array
// [kz0, kz0+nzl) is the slab of packed kz planes this rank owns (0, nz/2 on one GPU)
hipError_t launch_col_gen(int f64, int N, void* W, ColGeom g, long long ncols, const GenParams& gp,
                          const void* kspace, int kz0, int nzl, const void* tw, hipStream_t s,
                          bool prepare_only = false);
// x pass fused with the fast native generation (float32 arithmetic; float64 plans widen the result)
// (when the kz = 0 tiles run as a separate repairing launch first, `after_repair` is recorded between the two)
bool col_fastgen_supported(int f64, int N);   // false when tile + twiddles + generation tables exceed a CU's LDS (float64, N = 2048)
bool col_replicate_supported(int f64, int N, int nranks);   // can the x pass of length N keep 1/nranks of its rows?
// [x0, x1): rows that are stored (replicated-generation mode: W is then the local x slab, row x0 at its start);
// the default keeps every row.
hipError_t launch_col_fastgen(int f64, int N, void* W, ColGeom g, long long ncols, const FastGenParams& gp, int kz0, int nzl,
                              const void* tw, hipStream_t s, bool prepare_only = false, hipEvent_t after_repair = nullptr,
                              int x0 = 0, int x1 = 1 << 30, void* pot = nullptr, void* fixbuf = nullptr);
// (fixbuf: nx * ny complex of the plan's dtype -- the side buffer of repaired kz = 0 slots the pass fills and reads when it runs the
// kz = 0 tiles as a launch of their own: rf_kernels.h fix_fill_kernel; required for N >= 512 on the rank that owns kz = 0)
// (pot != nullptr, float32 only: the pass also stores delta(k) / k^2 into the API-layout array `pot` -- generate.py:200-217)
bool col_plain_addressable(int f64, int N, ColGeom g);   // false: the pass would need 64-bit lane offsets and has none (N < 1024)
int col_tile_cols(int f64, int N);   // tile width (columns) of the strided pass of length N, 0 if unsupported
// z pass of c2r: rows of M = nz/2 complex -> nz reals, scaled; partials[2*tile] = (sum, sumsq)
// the z pass of one slab (Wz, nrows rows) and the in-place y pass of another (Wy, ncols columns) in one launch (rf_k_yz.hip)
bool yz_merged_supported(int f64, int ny, int M);
bool yz_merged_fits(int f64, int ny, int M, ColGeom gy, long long nrows, long long ncols);
hipError_t launch_yz_merged(int f64, int ny, int M, void* Wz, long long nrows, double scale, const void* twz, double* partials, void* Wy, ColGeom gy,
                            long long ncols, const void* twy, hipStream_t s, bool prepare_only = false);
hipError_t launch_row_c2r(int f64, int M, void* W, long long nrows, double scale, const void* tw,
                          double* partials, hipStream_t s, bool prepare_only = false);
// z pass of a slab-decomposed plan: rows gathered from P received blocks of nzl kz planes each
hipError_t launch_row_c2r_gather(int f64, int M, const void* src, void* dst, long long nrows, double scale, int nzl,
                                 long long seg_stride, const void* tw, double* partials, hipStream_t s,
                                 bool prepare_only = false);
// z pass reading the blocked intermediate X [x block][kz tile][ny][rb][tc] (rf_fft.h xblock_*_geom) of the slab at `src` and writing
// the dense rows of W at `dst`; nrows = (x planes of the slab) * ny, in the order (x block, iy, row of the block)
hipError_t launch_row_c2r_xgather(int f64, int M, const void* src, void* dst, long long nrows, double scale, int tc, int rb, int ny,
                                  const void* tw, double* partials, hipStream_t s, bool prepare_only = false);
bool row_c2r_xgather_ok(int f64, int M, int tc, int rb);
// y pass (inverse, in place) that also leaves the Parseval partials of its output, one per tile (rf_fft.h AccColIO): columns
// C = hi * nzl + (kz - kz0)
hipError_t launch_col_plain_acc(int f64, int N, void* base, ColGeom g, long long ncols, int kz0, int nzl, double* partials, const void* tw,
                                hipStream_t s, bool prepare_only = false);
// sig[0] = sigma = sqrt(norm * sum of the n partials) (rounded to float32 first when sigma_as_float), Ap[z] = sqrt(log t) / sigma,
// Bp[z] = density[z] / sqrt(t) (density may be null: 1), t = 1 + (sigma growth[z])^2
hipError_t launch_lognormal_tables(const double* partials, long long n, double norm, const double* growth, const double* density, int nz,
                                   int sigma_as_float, double ap_unit, double* sig, double* Ap, double* Bp, hipStream_t s);
// z pass with rho = exp(delta Ap_z) Bp_z in its epilogue (rf_fft.h LognormalRowIO)
hipError_t launch_row_c2r_lognormal(int f64, int M, void* W, long long nrows, double scale, const double* Ap, const double* Bp,
                                    const void* tw, double* partials, hipStream_t s, bool prepare_only = false);
// the z pass with a per-z factor applied in its store (ScaleZRowIO)
hipError_t launch_row_c2r_zscale(int f64, int M, void* W, long long nrows, double scale, const double* Sz, const void* tw, double* partials,
                                 hipStream_t s, bool po = false);
// forward z pass (r2c rows, in place: nz reals -> nz/2 complex with (X[0], X[nz/2]) packed in element 0)
hipError_t launch_row_r2c(int f64, int M, void* W, long long nrows, const void* tw, hipStream_t s,
                          bool prepare_only = false);
// contiguous-axis pass of an unpacked c2c plan: rows of M = nz complex, in place; dir = +1 inverse (scaled), -1 forward;
// tw = exp(2 pi i q / M), q in [0, M)
hipError_t launch_row_c2c(int f64, int M, int dir, void* W, long long nrows, double scale, const void* tw, hipStream_t s,
                          bool prepare_only = false);
// packed device array [nx][ny][nz/2] -> API layout [nx][ny][nz/2+1] (separates the kz = 0 and nz/2 planes)
hipError_t launch_unpack_kspace(int f64, const void* W, void* K, int nx, int ny, int nzl, int kz0, hipStream_t s);
long long row_c2r_tiles(int f64, int M, long long nrows);

// stand-in for the all-to-all of a kz-slab rank without a communicator (diagnostics): nblk <= 16 blocks of `bytes` bytes (a multiple of
// 16) copied src[b] -> dst[b] by `workgroups` 256-thread workgroups
// (read_pct / write_pct: the share of every block that is read / written; 100 / 100 = the copy; sink: a device word nobody reads)
hipError_t launch_exchange_standin(const void* const* src, void* const* dst, int nblk, size_t bytes, int workgroups, int read_pct, int write_pct,
                                   unsigned* sink, hipStream_t s);
// word `slot` of each of the n <= 64 buffers bases_dev[t] (device table; null entries skipped) := value, stored from a kernel
hipError_t launch_peer_mark(void* const* bases_dev, int n, int slot, unsigned long long value, hipStream_t s);
// rows K,T,R,S into an API-layout k array [nx][ny][nz/2+1]
hipError_t launch_gen_kspace(int f64, void* K, const GenParams& gp, hipStream_t s);
// stats[0] = sum of partials[2i], stats[1] = sum of partials[2i+1]  (two levels through `scratch`, 512 doubles)
hipError_t launch_reduce_partials(const double* partials, long long n, double* stats, double* scratch, hipStream_t s);
hipError_t launch_lognormal(int f64, void* W, long long nrows, int nz, const double* a_z, const double* b_z,
                            double sigma, hipStream_t s);
hipError_t launch_affine_z(int f64, void* W, long long nrows, int nz, const double* mul_z, double add,
                           hipStream_t s);
// psi[x][y][e] = -2 * Simpson_{j in [i_min, e]} (cot_z[j] - cot_z[e]) phi[x][y][j] h  (generate.py:397-411), dense real arrays
hipError_t launch_lensing(int f64, const void* phi, void* psi, long long nrows, int nz, const double* cot_z, double h, int i_min,
                          hipStream_t s);
// P = K / k^2 (0 at DC), API layout; and K = scale * P
// (side arrays of a kz-slab rank: rows of zpitch slots, first plane zoff -- GenParams)
hipError_t launch_save_potential(int f64, const void* K, void* P, int nx, int ny, int nz, const double* kx2,
                                 const double* ky2, const double* kz2, int zpitch, int zoff, int ppitch, hipStream_t s);
// (n = cells of K; P's rows hold ppitch >= zpitch cells)
hipError_t launch_scale_copy(int f64, const void* P, void* K, long long n, int zpitch, int ppitch, double scale, hipStream_t s);

// on-GPU replay of RandomState(seed).normal (rf_k_mt.hip)
// one stage of the jump tree: states[i + m * dist] = states[i] advanced by m * dist segments, i < nsrc, m = 1 .. nmult
// (pos: nmult rows of pos_stride positions, npos: their lengths, both in device memory)
hipError_t launch_mt_jump(uint32_t* states, const uint32_t* pos, const int* npos, int pos_stride, int nsrc, long long dist, int nmult,
                          int nseg, hipStream_t s);
// every segment (one wave each) generates its blocks once and writes the deviate pairs of its accepted polar attempts
// densely from slot seg * cap of `runs` (float64 pairs, or float32 pairs with `single`); counts[seg] = their number
hipError_t launch_mt_polar(bool single, const uint32_t* states, int nseg, int blocks_per_segment, long long total_blocks,
                           unsigned long long* counts, void* runs, unsigned long long cap, hipStream_t s);
// moves the runs into cell order (offsets = exclusive scan of counts); a kz-slab rank keeps its own planes (nzh = nz/2 + 1
// cells per row of the stream; zpitch / zoff: the destination's rows, see GenParams)
hipError_t launch_mt_compact(bool single, const void* scratch, const unsigned long long* counts, const unsigned long long* offsets,
                             int nseg, unsigned long long cap, void* noise, unsigned long long ncells, int nzh, int zpitch, int zoff,
                             hipStream_t s);
// (single: `noise` is an array of float32 pairs instead of float64 pairs)
// (nzh = nz/2 + 1 cells per row of the stream; zpitch / zoff: the noise buffer's rows, see GenParams)
hipError_t launch_mt_scan(const unsigned long long* counts, unsigned long long* offsets, int n, hipStream_t s);
// float32 form: the row table the generation pass locates its pairs with (rf_core.h RowLoc; tab: nx * ny entries of 8 bytes, index
// iy * nx + ix; offsets: the scan, nseg + 1 entries; flags[0] |= 1 if a row would span more than two segments)
hipError_t launch_mt_rowtab(const unsigned long long* offsets, int nseg, void* tab, int nx, int ny, int nzh, int* flags, hipStream_t s);
// distributed replay: pack the local segments' pairs by destination rank / widen a received float32 stream into the resident deviates
hipError_t launch_mt_share_pack(bool single, const void* scratch, const unsigned long long* counts, const unsigned long long* first_cell,
                                int nseg, unsigned long long cap, void* send, unsigned long long ncells, int nzh, int nzl, int nranks,
                                const long long* sbase_dev, hipStream_t s);
hipError_t launch_mt_share_widen(const void* recv, double* noise, long long npairs, hipStream_t s);

// non-power-of-two grids (rf_generic.h, rf_k_generic.hip); root = exp(2 pi i t / n) tables as made by make_twiddles
// complex pass along one axis: line l starts at (l / inner) * outer + l % inner, elements `stride` apart; src == dst allowed
hipError_t launch_generic_axis(int f64, const void* src, void* dst, const GenericAxis& ax, long long stride, long long inner,
                               long long outer, long long nlines, const void* root, int sign, double scale, hipStream_t s);
// axes too long for the LDS (rf_generic.h GenericLong): one step of the four-step transform (lines with sub-lines); the Hermitian
// (un)tangle of long rows as passes of their own; (sum, sum of squares) of a real array into nblocks partial pairs
hipError_t launch_generic_lines(int f64, const void* src, void* dst, const GenericLines& L, const void* root, hipStream_t s);
hipError_t launch_generic_untangle(int f64, const void* G, void* Z, int M, long long nrows, const void* root, hipStream_t s);
hipError_t launch_generic_tangle(int f64, const void* Z, void* G, int M, long long nrows, const void* root, hipStream_t s);
hipError_t launch_generic_moments(int f64, const void* W, long long n, double* partials, long long nblocks, hipStream_t s);
// rows of nz/2+1 half-spectrum bins (G) <-> dense rows of nz reals (W); ax factors nz/2, root_nz has nz entries;
// the c2r pass leaves (sum, sumsq) of block b in partials[2b], partials[2b+1]
long long generic_row_blocks(int f64, const GenericAxis& ax, long long nrows);
hipError_t launch_generic_row_c2r(int f64, const void* G, void* W, const GenericAxis& ax, long long nrows, const void* root_nz,
                                  double scale, double* partials, hipStream_t s);
hipError_t launch_generic_row_r2c(int f64, const void* W, void* G, const GenericAxis& ax, long long nrows, const void* root_nz,
                                  hipStream_t s);

}  // namespace rf
