// rf_kernels.h -- __global__ wrappers around the phase functions of rf_fft.h
// (gfx950 only; compiled by hipcc).  One workgroup = one tile.
#pragma once
#include <hip/hip_runtime.h>
#include "rf_configs.h"

namespace rf {

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one L2).  Give
// each XCD a contiguous run of tiles so that neighbouring tiles -- which share
// 128-byte lines when a tile row is 64 B wide -- meet in the same L2.
__device__ __forceinline__ long long xcd_tile(long long b, long long nb) {
  return (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
}

// waves per SIMD the register allocator must leave room for: as many workgroups per CU as the LDS
// footprint admits (160 KiB), capped at 4 waves/SIMD (128 VGPRs) so radix-16 butterflies never spill
template <class C, class IO>
constexpr int col_min_waves() {
  constexpr int lds = C::LDS_BYTES + IO::LDS_EXTRA;
  constexpr int blocks = lds > 0 ? (163840 / lds > 0 ? 163840 / lds : 1) : 8;
  constexpr int w = blocks * C::NT / 256;
  return w < 1 ? 1 : (w > 4 ? 4 : w);
}

// one tile of a strided pass (the body of col_kernel)
template <class C, int DIR, class IO>
__device__ __forceinline__ void col_body(IO& io, const cplx<typename C::T>* __restrict__ tw, long long tile, char* rf_smem) {
  using F = ColFFT<C, DIR, IO>;
  using cx = cplx<typename C::T>;
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  const int tid = threadIdx.x;
  const cx* ltw = tw;
  io.bind_seed();
  typename F::TwRegs twr;
  typename F::PreRegs pre;
  F::preload(tid, tile, io, pre);                     // (IOs that read memory in pass 1 AND stage tables: their loads go out first)
  if (IO::LDS_EXTRA > 0) {
    io.prologue(tid, C::NT, F::lds_io(lds));          // the IO's tables -> LDS: pass 1 reads them
    __syncthreads();
  }
  if (C::NPASS >= 2) F::tw_fetch(tid, tw, twr);       // twiddles global -> registers: issued here, landed under pass 1
  F::pass_first(tid, tile, io, lds, pre, true);
  if (C::NPASS >= 2) {
    F::tw_stage(tid, lds, twr);                       // -> LDS, in front of the barrier that precedes their first use
    ltw = F::lds_tw(lds);
  }
  if (C::NPASS == 3) {
    typename F::Regs r;
    __syncthreads();
    F::pass_mid_read(tid, ltw, lds, r);
    __syncthreads();
    F::pass_mid_write(tid, lds, r);
  }
  if (C::NPASS >= 2) {
    __syncthreads();
    F::pass_last(tid, tile, io, ltw, lds);
  }
  if constexpr (IO::HAS_FINISH) {            // (AccColIO: the workgroup's sum of w |Y|^2 -> partials[tile])
    double a = io.weighted_sum();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    __syncthreads();                         // the LDS tile is no longer read
    io.finish(tid, C::NT, reinterpret_cast<double*>(rf_smem), tile, a, [] { __syncthreads(); });
  }
}

template <class C, int DIR, class IO>
__global__ __launch_bounds__(C::NT, (col_min_waves<C, IO>())) void col_kernel(IO io, const cplx<typename C::T>* __restrict__ tw,
                                                                           long long ntiles, long long tile_mul,
                                                                           long long tile_add, int skip_period) {
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  long long tile = xcd_tile(blockIdx.x, ntiles) * tile_mul + tile_add;          // (1, 0) unless a tile subset is run
  if (skip_period > 0) {                                                          // all tiles except those = 0 mod skip_period
    const unsigned t = (unsigned)tile;
    tile = (long long)(t + t / (unsigned)(skip_period - 1) + 1u);
  }
  col_body<C, DIR, IO>(io, tw, io.remap_tile(tile), rf_smem);                     // (remap: identity except for XposeColIO)
}

// The repaired slots kz = 0 of every mode (ix, iy) -- (plane kz = 0) + i (plane kz = nz/2), each Hermitian-symmetrised
// (transform.py:141-158; rf_core.h fast_fix_kz0) -- into the side buffer out[iy * nx + ix] that the FIX = 3 launch of the generation
// pass reads.  IOF = the pass's IO with FIX = 1 and rows = modes (fill_io): the SAME fix_value() the pass itself would evaluate,
// including the potential's two planes when POT = 1 -- but by one thread per mode, all lanes busy, in a kernel of its own
// (nx ny threads: 10 - 40 us; the repair inside the pass cost 0.05 ms per 1024^3 realisation and 0.25 ms on rank 0 of 2048^3 / 8).
template <class IOF, class CT>
__global__ __launch_bounds__(256) void fix_fill_kernel(IOF io, CT* __restrict__ out, int nx, int ny) {
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  io.bind_seed();
  if (IOF::LDS_EXTRA > 0) {
    io.prologue(threadIdx.x, 256, rf_smem);            // the sigma records -> LDS
    __syncthreads();
  }
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)nx * ny) return;
  const int iy = (int)(e / nx), ix = (int)(e - (long long)iy * nx);
  out[e] = io.fix_value((long long)iy * io.nzl, ix, 0);
}

// strided pass of length 2 C1::N as two C1 transforms per tile + a radix-2 step in registers (rf_fft.h Col2)
// (the kernels that repair the kz = 0 slot -- FIX = 1 computes it, FIX = 3 loads it from the side buffer -- hold the eight values next
// to the parked half: 140 - 260 registers.  They run few tiles and get the budget of two waves per SIMD instead of 190 - 300 bytes
// of scratch per thread at four)
// one tile of a Col2 pass (the body of col2_kernel; yz_merged_kernel runs it for its y tiles too)
template <class C1, int DIR, class IO>
__device__ __forceinline__ void col2_body(IO& io, const cplx<typename C1::T>* __restrict__ tw2, long long tile, char* rf_smem) {
  using X = Col2<C1, DIR, IO>;
  using F = typename X::F;
  using cx = cplx<typename C1::T>;
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  const int tid = threadIdx.x;
  io.bind_seed();
  typename F::TwRegs twr;
  typename X::Park pk;
  if (IO::LDS_EXTRA > 0) {
    io.prologue(tid, C1::NT, F::lds_io(lds));
    __syncthreads();
  }
  X::tw_fetch(tid, tw2, twr);
  const cx* ltw = F::lds_tw(lds);
#pragma unroll
  for (int phase = 0; phase < 2; ++phase) {
    io.set_phase(phase);
    if (phase == 1) __syncthreads();                 // the tile in LDS is free again
    // both phases run the same passes on the same LDS addresses: hidden from the optimiser (an opaque copy of the thread index
    // per phase), or it keeps the first phase's ~40 addresses and twiddles live through the whole kernel, next to the 32 parked
    // registers, and spills (248 bytes of scratch per thread, 2.8x slower on MI355X)
    // (the copy is opaque, its RANGE is not: masked back to [0, NT), so that t / LPR, t % TC ... stay single shifts and masks --
    // without the mask the compiler treats t as an arbitrary signed integer and spends ~630 of the kernel's 3079 vector
    // instructions on sign-correct divisions and un-folded LDS addresses)
    unsigned tu = (unsigned)tid;
    asm volatile("" : "+v"(tu));
    // float32 passes see the unmasked index in both phases: at 122 registers next to the 32 parked ones the folded addresses cost 12 - 60
    // bytes of scratch per thread and the 2048^3 x pass runs 9 % SLOWER (12.4 -> 13.6 ms); the float64 generation pass (104 registers)
    // takes the mask in both phases: 2.63 -> 2.47 ms per 1024^3
    constexpr int mask_phases = sizeof(typename C1::T) == 8 ? 3 : 0;
    const int t = ((mask_phases >> phase) & 1) ? (int)(tu & (unsigned)(C1::NT - 1)) : (int)tu;
    static_assert((C1::NT & (C1::NT - 1)) == 0, "the thread count of a Col2 pass is a power of two");
    if constexpr (F::PRELOAD) {                      // (IOs that read memory in pass 1: this phase's loads go out first)
      typename F::PreRegs pre;
      F::preload(t, tile, io, pre);
      F::pass_first(t, tile, io, lds, pre, true);
    } else {
      F::pass_first(t, tile, io, lds);
    }
    if (phase == 0) F::tw_stage(tid, lds, twr);
    if (C1::NPASS == 3) {
      typename F::Regs r;
      __syncthreads();
      F::pass_mid_read(t, ltw, lds, r);
      __syncthreads();
      F::pass_mid_write(t, lds, r);
    }
    __syncthreads();
    if (phase == 0) X::last_park(t, ltw, lds, pk);
    else X::last_combine(t, tile, io, ltw, tw2, lds, pk);
  }
}

template <class C1, int DIR, class IO>
__global__ __launch_bounds__(C1::NT, (IO::FIX_MODE != 0 ? 2 : col_min_waves<C1, IO>())) void col2_kernel(IO io, const cplx<typename C1::T>* __restrict__ tw2,
                                                                             long long ntiles, long long tile_mul,
                                                                             long long tile_add, int skip_period) {
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  long long tile = xcd_tile(blockIdx.x, ntiles) * tile_mul + tile_add;
  if (skip_period > 0) {
    const unsigned t = (unsigned)tile;
    tile = (long long)(t + t / (unsigned)(skip_period - 1) + 1u);
  }
  col2_body<C1, DIR, IO>(io, tw2, tile, rf_smem);
}

// two adjacent tiles per workgroup, whole-line stores (rf_fft.h ColPair): pair p = tiles 2p and 2p + 1
// one phase of a pair: the passes of tile 2 pair + PHASE up to its last butterfly; phase 0 parks, phase 1 stores both tiles
template <class C, int DIR, class IO, int PHASE>
__device__ __forceinline__ void colpair_phase(IO& io, const cplx<typename C::T>* ltw, long long pair, cplx<typename C::T>* lds,
                                              typename ColFFT<C, DIR, IO>::TwRegs& twr, typename ColPair<C, DIR, IO>::Park& pk) {
  using X = ColPair<C, DIR, IO>;
  using F = typename X::F;
  const int tid = threadIdx.x;
  const long long tile = 2 * pair + PHASE;
  if (PHASE == 1) __syncthreads();                   // the tile in LDS is free again
  // (the two phases run the same passes on the same LDS addresses: an opaque copy of the thread index per phase keeps the optimiser
  // from carrying the first phase's addresses and twiddles through the second next to the parked registers -- see col2_body)
  unsigned tu = (unsigned)tid;
  asm volatile("" : "+v"(tu));
  const int t = (int)tu;
  // (the second tile of a pair has an odd index: it never holds the slot kz = 0, so its pass is built without the repair)
  typename F::PreRegs pre;
  if constexpr (F::PRELOAD) F::preload(t, tile, io, pre);
  F::template pass_first<PHASE == 0>(t, tile, io, lds, pre, F::PRELOAD);
  if (PHASE == 0) F::tw_stage(tid, lds, twr);
  if (C::NPASS == 3) {
    typename F::Regs r;
    __syncthreads();
    F::pass_mid_read(t, ltw, lds, r);
    __syncthreads();
    F::pass_mid_write(t, lds, r);
  }
  __syncthreads();
  if (PHASE == 0) X::last_park(t, ltw, lds, pk);
  else X::last_store(t, 2 * pair, io, ltw, lds, pk);
}

template <class C, int DIR, class IO>
__device__ __forceinline__ void colpair_body(IO& io, const cplx<typename C::T>* __restrict__ tw, long long pair, char* rf_smem) {
  using X = ColPair<C, DIR, IO>;
  using F = typename X::F;
  using cx = cplx<typename C::T>;
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  const int tid = threadIdx.x;
  io.bind_seed();
  typename F::TwRegs twr;
  typename X::Park pk;
  if (IO::LDS_EXTRA > 0) {
    io.prologue(tid, C::NT, F::lds_io(lds));
    __syncthreads();
  }
  F::tw_fetch(tid, tw, twr);
  const cx* ltw = F::lds_tw(lds);
  colpair_phase<C, DIR, IO, 0>(io, ltw, pair, lds, twr, pk);
  colpair_phase<C, DIR, IO, 1>(io, ltw, pair, lds, twr, pk);
}

// pairs b * pair_mul + pair_add, b in [0, npairs); skip_period > 0: all pairs except those = 0 mod skip_period
template <class C, int DIR, class IO>
__global__ __launch_bounds__(C::NT, (col_min_waves<C, IO>())) void colpair_kernel(IO io, const cplx<typename C::T>* __restrict__ tw,
                                                                               long long npairs, long long pair_mul, long long pair_add, int skip_period) {
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  long long pair = xcd_tile(blockIdx.x, npairs) * pair_mul + pair_add;
  if (skip_period > 0) {
    const unsigned t = (unsigned)pair;
    pair = (long long)(t + t / (unsigned)(skip_period - 1) + 1u);
  }
  colpair_body<C, DIR, IO>(io, tw, pair, rf_smem);
}

// one tile of the z pass (the body of row_c2r_kernel; yz_merged_kernel runs it for its z tiles): c2r rows + the workgroup's
// (sum, sum of squares) into partials[2 tile]
template <class C, class IO>
__device__ __forceinline__ void row_c2r_body(IO& io, const cplx<typename C::T>* __restrict__ tw, long long nrows, double* __restrict__ partials,
                                             long long tile, char* rf_smem) {
  using F = RowC2R<C, IO>;
  using cx = cplx<typename C::T>;
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  const int tid = threadIdx.x;
  typename F::Regs r;
  typename F::TwRegs twr;
  const cx* ltw = F::lds_tw(lds);
  F::tw_fetch(tid, tw, twr);                 // twiddles global -> registers
  if constexpr (row_io_wants_stage<IO>::value) io.template stage<C>(tid, rf_smem);   // (the IO's table -> spare LDS slots of the tile, in front of the first barrier)
  if constexpr (sizeof(typename C::T) == 4) {
    // ... then the rows global -> registers: both trips to memory in flight together; the (older) table loads are waited for first
    typename F::In in;
    F::pass_first_load(tid, tile, nrows, io, in);
    F::tw_stage(tid, lds, twr);              // twiddles -> LDS
    __syncthreads();
    F::pass_first_compute(tid, tile, nrows, io, ltw, lds, r, in);
  } else {
    // (float64: holding the 16 complex128 inputs across the barrier spills -- 72 / 136 bytes of scratch; one trip after the other)
    F::tw_stage(tid, lds, twr);
    __syncthreads();
    F::pass_first(tid, tile, nrows, io, ltw, lds, r);
  }
  // (the exchange between the middle and the last radix-8 stage through the wave's cross-lane network instead of the LDS row image was
  // measured in round 3 -- 24 ds_bpermute per thread against 8 + 8 LDS accesses and two barriers -- and lost: DESIGN_HISTORY.md)
  if (C::NPASS == 3) {
    __syncthreads();
    F::pass_mid_read(tid, ltw, lds, r);
    __syncthreads();
    F::pass_mid_write(tid, lds, r);
  }
  if (C::NPASS >= 2) {
    __syncthreads();
    F::pass_last(tid, tile, nrows, io, ltw, lds, r);
  }
  // workgroup reduction of the moments: wave shuffle, then one slot per wave in LDS
  double s1 = r.mom.sum(), s2 = r.mom.sumsq();
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_down(s1, off);
    s2 += __shfl_down(s2, off);
  }
  __syncthreads();  // LDS tile no longer needed
  double* red = reinterpret_cast<double*>(rf_smem);
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; }
  __syncthreads();
  if (tid == 0) {
    double a = 0, b = 0;
#pragma unroll
    for (int w = 0; w < C::NT / 64; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
    partials[2 * tile] = a;
    partials[2 * tile + 1] = b;
  }
}

// z pass: c2r rows + per-workgroup (sum, sum of squares) partials
// Waves per SIMD the register allocator must leave room for.  The float64 pass of rows of 512 complex allocates 182 VGPRs when left
// alone: two waves per SIMD, although its LDS footprint (53 KB) admits three workgroups per CU.  Held to three waves (168 VGPRs, 20 - 28
// bytes of scratch per thread) the passes WITH an epilogue (LognormalRowIO, ScaleZRowIO: row_io_pre) gain -- the fused lognormal z pass
// 4.00 -> 3.70 ms per 1024^3 float64 on MI355X, profiles/r05_ab/r05_b_ln_*.log -- while the plain pass, which already sits at the copy
// ceiling, loses 4 % (3.18 -> 3.32 ms) and keeps its registers.  (Rows of 1024 complex128: 184 bytes of scratch at three waves; left alone.)
template <class C, class IO>
constexpr int row_min_waves() {
  return (sizeof(typename C::T) == 8 && C::M == 512 && row_io_pre<IO, C::RL>::value) ? 3 : 1;
}
template <class C, class IO>
__global__ __launch_bounds__(C::NT, (row_min_waves<C, IO>())) void row_c2r_kernel(IO io, const cplx<typename C::T>* __restrict__ tw,
                                                        long long nrows, double* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  // last rows first: the pass before this one wrote the array front to back, so its end is what the 256 MiB Infinity
  // Cache still holds (measured: DESIGN.md section 3.8)
  const long long tile = (long long)gridDim.x - 1 - blockIdx.x;
  row_c2r_body<C, IO>(io, tw, nrows, partials, tile, rf_smem);
}

// forward z pass: r2c rows in place
template <class C, class IO>
__global__ __launch_bounds__(C::NT) void row_r2c_kernel(IO io, const cplx<typename C::T>* __restrict__ tw, long long nrows) {
  using F = RowR2C<C, IO>;
  using cx = cplx<typename C::T>;
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  const int tid = threadIdx.x;
  const long long tile = blockIdx.x;
  F::prologue(tid, tw, lds);                 // twiddles -> LDS
  const cx* ltw = F::lds_tw(lds);
  if (C::NPASS >= 2) F::pass_first(tid, tile, nrows, io, lds);
  if (C::NPASS == 3) {
    typename F::Regs r;
    __syncthreads();
    F::pass_mid_read(tid, ltw, lds, r);
    __syncthreads();
    F::pass_mid_write(tid, lds, r);
  }
  __syncthreads();
  F::pass_last(tid, tile, nrows, io, ltw, lds);
}

// unpacked c2c row pass (either direction), in place
template <class C, int DIR, class IO>
__global__ __launch_bounds__(C::NT) void row_c2c_kernel(IO io, const cplx<typename C::T>* __restrict__ tw, long long nrows) {
  using F = RowC2C<C, DIR, IO>;
  using cx = cplx<typename C::T>;
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  const int tid = threadIdx.x;
  const long long tile = blockIdx.x;
  F::prologue(tid, tw, lds);                 // twiddles -> LDS (first read after the next barrier)
  const cx* ltw = F::lds_tw(lds);
  F::pass_first(tid, tile, nrows, io, lds);
  if (C::NPASS == 3) {
    typename F::Regs r;
    __syncthreads();
    F::pass_mid_read(tid, ltw, lds, r);
    __syncthreads();
    F::pass_mid_write(tid, lds, r);
  }
  if (C::NPASS >= 2) {
    __syncthreads();
    F::pass_last(tid, tile, nrows, io, ltw, lds);
  }
}

}  // namespace rf
