// rf_fused.h -- x pass (generation) and y pass of the c2r transform in ONE launch.
//
// Why: the generation x pass is VALU-bound (it reads nothing), the y pass is HBM-bound, and two kernels on
// two streams do not overlap on this chip (each grid fills every CU before the other gets a slot; measured).
// Here one grid carries both kinds of work items, ordered so that the hardware dispatcher interleaves x tiles
// of kz-chunk c+1 with y tiles of chunk c: a CU typically holds one VALU-heavy and one memory-heavy workgroup,
// and the y pass reads data that was written about a chunk ago (Infinity-Cache distance).
//
//   chunk c  = G kz tiles (G / LT whole 128-byte lines spread evenly over the kz row), all ix, all iy
//   X item   = (c, iy): generate + x-FFT the G tiles of row iy           (no dependency)
//   Y item   = (c, ix): y-FFT the G tiles of plane ix, in place          (needs ALL X items of chunk c)
//   X stream = X items chunk-major, Y stream likewise; the grid order is
//              X[0 .. D) | 8 of X[D + ..], 8 of Y[..], alternating | rest of Y,   D = (items per chunk) + slack
//   one workgroup = one item, item = blockIdx.x
//
// Dependencies.  An X item waits for its own stores (s_waitcnt vmcnt(0): the data is then in its XCD's L2),
// records its XCD in xmask[c] and increments done[c].  The XCD L2s do not snoop each other and plain (or
// sc0/sc1) stores to ordinary device memory stay in the writer's L2 (measured), so before chunk c is read every
// XCD that produced part of it must write its L2 back: the first Y item of the chunk that runs on XCD k does
// ONE buffer_wbl2 for that XCD and sets bit k of flushed[c]; Y items wait until done[c] is complete and
// flushed[c] covers xmask[c].  (A write-back per X item instead costs 3 ms per 1024^3 launch; measured.)
// No invalidate is needed on the reading side: an item covers whole 128-byte lines, every line of W is written
// by exactly one X item and read by exactly one Y item per launch, and caches start a kernel clean.
//
// Progress.  Workgroups are dispatched in block order within each XCD.  Take the unfinished item with the
// smallest index: everything before it has finished, so it is resident (or next in line for a free slot), and
// what it waits for -- X items with smaller indices and a write-back by Y items of its own chunk, which the
// first resident Y item of every XCD performs itself before waiting -- can complete.  The waits are bounded all
// the same (spin_limit): on a timeout the kernel sets the sticky abort flag and the host reports an error.
#pragma once
#include "rf_fft.h"

namespace rf {

struct FusedSched {
  unsigned* ctrl;        // per chunk c: ctrl[4c] = finished X items, ctrl[4c+1] = xmask, ctrl[4c+2] = flushed   (zeroed before every launch)
  unsigned* abort_flag;  // sticky: set when a dependency wait timed out
  int nchunks, G;        // kz-tile chunks, kz tiles per chunk
  int ktiles;            // kz tiles per row = nzl / TC  (== nchunks * G)
  int per;               // items per chunk in either stream = nx = ny
  unsigned delay;        // D: how many X items the Y stream lags behind
  unsigned spin_limit;
};

struct FusedItem { int role /* 0 = X, 1 = Y */, chunk, t; };

RF_HD unsigned fused_total_items(const FusedSched& s) { return 2u * (unsigned)s.nchunks * (unsigned)s.per; }

// grid index -> work item.  The streams alternate in blocks of FUSED_BLOCK items: workgroups go to the XCDs
// round-robin, so a 1:1 alternation would put every X item on the odd XCDs and every Y item on the even ones
// (and no Y item would ever write back the odd XCDs' L2s: observed).  n and D are multiples of FUSED_BLOCK.
enum { FUSED_BLOCK = 8 };
RF_HD FusedItem fused_decode(const FusedSched& s, unsigned item) {
  const unsigned n = (unsigned)s.nchunks * (unsigned)s.per;      // items per stream
  const unsigned D = s.delay < n ? s.delay : n;
  unsigned role, k;
  if (item < D) { role = 0; k = item; }
  else if (item - D < 2 * (n - D)) {
    const unsigned r = item - D, grp = r / (2 * FUSED_BLOCK), in = r % (2 * FUSED_BLOCK);
    role = in >= FUSED_BLOCK ? 1u : 0u;
    k = role ? grp * FUSED_BLOCK + (in - FUSED_BLOCK) : D + grp * FUSED_BLOCK + in;
  } else { role = 1; k = (n - D) + (item - D - 2 * (n - D)); }
  FusedItem w;
  w.role = (int)role; w.chunk = (int)(k / (unsigned)s.per); w.t = (int)(k % (unsigned)s.per);
  return w;
}

#if defined(__HIPCC__)
}  // namespace rf
#include "rf_kernels.h"
namespace rf {

// one tile through the three phases of a column pass (barriers in between)
template <class F, class C, class IO>
__device__ __forceinline__ void col_run_tile(int tid, long long tile, const IO& io, const cplx<typename C::T>* ltw,
                                             cplx<typename C::T>* lds) {
  F::pass_first(tid, tile, io, lds);
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
}

// kz tile g of item t of chunk c.  A chunk is NOT a contiguous kz range: its G / LT lines (LT tiles = one 128-byte
// line each) are spread evenly over the kz row, and item t starts at line t % (G / LT), so the workgroups that
// run together touch every part of a row -- a contiguous chunk puts all concurrent traffic on the same 128-byte
// window of each 4-KiB row and camps on a few HBM channels (measured: 4.4 ms instead of 3.5 ms per 1024^3 x+y).
template <int LT> __device__ __forceinline__ int fused_tile_of(const FusedSched& s, int c, int t, int g) {
  const int L = s.G / LT;                       // lines per item
  const int line = (g / LT + t) % L, h = g % LT;
  return (line * s.nchunks + c) * LT + h;
}

// XCD the wave runs on (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ unsigned fused_xcd_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu; }

// An opaque per-tile copy of the thread index (with its range): keeps the compiler from carrying one tile's
// per-thread invariants (LDS addresses, twiddle registers) across the tile loop, which costs ~30 spilled VGPRs.
template <int NT> __device__ __forceinline__ int fused_fresh_tid(int tid) {
  int t = tid;
  asm volatile("" : "+v"(t));
  __builtin_assume(t >= 0 && t < NT);
  return t;
}

// bounded wait of wave 0 until pred(load(addr)) holds; returns 0 on timeout / abort
template <class Pred>
__device__ __forceinline__ unsigned fused_wait(const unsigned* addr, Pred pred, const FusedSched& s) {
  unsigned spins = 0;
  while (!pred(__hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
    __builtin_amdgcn_s_sleep(4);
    if (++spins > s.spin_limit || __hip_atomic_load(s.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      __hip_atomic_store(s.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0u;
    }
  }
  return 1u;
}

// IOX: the generation IO WITHOUT the kz = 0 repair (the repair's register pressure would spill the whole kernel):
// when skip_kz0_tile is set, the tiles that hold slot kz = 0 have been produced by the repairing x-pass kernel
// in an earlier launch on the same stream and the X role leaves them alone.  IOY: the plain in-place IO.
template <class CX, class CY, class IOX, class IOY>
__global__ __launch_bounds__(CX::NT, (col_min_waves<CX, IOX>())) void xy_fused_kernel(IOX iox, IOY ioy,
                                                                                    const cplx<typename CX::T>* __restrict__ tw,
                                                                                    FusedSched s, int skip_kz0_tile) {
  static_assert(CX::N == CY::N && CX::TC == CY::TC && CX::NT == CY::NT && CX::NPASS >= 2 && CY::NPASS >= 2,
                "the fused kernel shares one LDS carve and one twiddle table between the two passes");
  using FX = ColFFT<CX, +1, IOX>;
  using FY = ColFFT<CY, +1, IOY>;
  using cx = cplx<typename CX::T>;
  constexpr int LT = 128 / (CX::TC * (int)sizeof(cx)) > 1 ? 128 / (CX::TC * (int)sizeof(cx)) : 1;   // tiles per 128-byte line
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  const int tid = threadIdx.x;
  const FusedItem w = fused_decode(s, blockIdx.x);
  unsigned* done = s.ctrl + 4 * w.chunk;
  unsigned* xmask = done + 1;
  unsigned* flushed = done + 2;
  // the verdict of wave 0's wait reaches the other waves through the last 4 bytes of the LDS allocation (the
  // last record slot of the generation table: the launcher only takes this kernel when that slot is unused;
  // 2 x (tile + twiddles + records) is exactly the 160 KiB of a CU)
  static_assert(IOX::LDS_EXTRA >= 16, "the generation IO must carry its LDS table");
  volatile unsigned* mail = reinterpret_cast<volatile unsigned*>(rf_smem + CX::LDS_BYTES + IOX::LDS_EXTRA - 16);
  const cx* ltw = FX::lds_tw(lds);
  if (w.role == 0) {
    FX::prologue(tid, iox, tw, lds);                         // twiddles + generation tables -> LDS
    __syncthreads();
    for (int g = 0; g < s.G; ++g) {
      const int kt = fused_tile_of<LT>(s, w.chunk, w.t, g);
      if (kt == 0 && skip_kz0_tile) continue;                // uniform; produced by the repairing launch
      const int t_it = fused_fresh_tid<CX::NT>(tid);
      col_run_tile<FX, CX>(t_it, (long long)w.t * s.ktiles + kt, iox, ltw, lds);
      __syncthreads();                                       // the LDS tile is reused
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's stores have reached the XCD's L2
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_or(xmask, 1u << fused_xcd_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the mask before the count
      __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    if (CY::NPASS >= 2) {                                     // twiddles -> LDS while wave 0 waits
      cx* l = FY::lds_tw(lds);
      for (int i = tid; i < CY::N; i += CY::NT) l[i] = tw[i];
    }
    if (tid < 64) {
      const unsigned need = (unsigned)s.per;
      unsigned ok = fused_wait(done, [need](unsigned v) { return v >= need; }, s);
      if (ok) {
        const unsigned bit = 1u << fused_xcd_id();
        if ((__hip_atomic_load(flushed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0u) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // buffer_wbl2: this XCD's share of the chunk goes to memory
          __hip_atomic_fetch_or(flushed, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned want = __hip_atomic_load(xmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = fused_wait(flushed, [want](unsigned v) { return (v & want) == want; }, s);
      }
      if (tid == 0) mail[0] = ok;
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(mail[0]) != 0u) {
      for (int g = 0; g < s.G; ++g) {
        const int t_it = fused_fresh_tid<CY::NT>(tid);
        col_run_tile<FY, CY>(t_it, (long long)w.t * s.ktiles + fused_tile_of<LT>(s, w.chunk, w.t, g), ioy, ltw, lds);
        __syncthreads();
      }
    }
  }
}
#endif  // __HIPCC__

}  // namespace rf
