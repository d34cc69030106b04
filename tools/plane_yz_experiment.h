// plane_yz_experiment.h -- EXPERIMENT (round 2), not part of the product: y and z passes fused plane by plane through
// L2 / Infinity-Cache resident scratch planes, with per-XCD ticket queues and dataflow flags.  Correct by construction
// of the dependency order and it runs, but on MI355X it is not faster than the two separate passes yet: 3.23 ms against
// 1.58 + 1.49 ms at 1024^3 (tools/xbench.hip `f`; ticket + flag overhead alone 0.54 ms; each role alone 2.0 ms); the
// non-persistent v2 at the end of this file (`g`) takes 5.05 ms.  The
// per-item critical path (load -> three LDS passes -> store, ~12 us) times twice as many items over the same 512
// resident workgroups is what bounds it, not HBM.  See DESIGN.md section 3.5.
#pragma once
#include "rf_kernels.h"

namespace rf {

// y pass of ONE x-plane: reads the plane of W (row = iy, stride nzc; one tile = TC adjacent kz columns) and writes the
// same layout into a scratch plane that stays in the XCD's L2 / the memory-side cache until the z items read it.
template <typename T> struct PlaneYIO {
  const cplx<T>* src;            // plane of W
  cplx<T>* dst;                  // scratch plane
  ColGeom g;                     // {inner = nzc, outer_stride = 0, row_stride = nzc}
  RF_HD V16<T> load(long long C0, int cl, int rb, int ro) const { return v16_load<T>(g.at<false>(src, C0, cl, rb, ro)); }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<T>& v) const { v16_store<T>(g.at<false>(dst, C0, cl, rb, ro), v); }
  static constexpr int FIX_MODE = 0;
  RF_HD bool needs_fix(long long) const { return false; }
  RF_HD cplx<T> fix_value(long long, int, int) const { return cplx<T>(); }
  static constexpr int LDS_EXTRA = 0;
  RF_HD void prologue(int, int, void*) {}
  RF_HD void bind_seed() {}
  RF_HD static void sched_fence(int = 0) {}
  static constexpr bool ROLLED_LOAD = false;
};
// z pass of rows of that plane: complex rows from the scratch plane (written by OTHER compute units of the same XCD:
// the loads must not be served from this CU's vector L1 -- non-temporal loads bypass it), real rows into W.
template <typename T> struct PlaneZIO {
  const cplx<T>* src;            // scratch plane [ny][M]
  cplx<T>* dst;                  // plane of W, viewed as [ny][M] complex = [ny][2M] real
  T scale;
  int M_of;
  RF_HD cplx<T> load(long long row, int k) const {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef T vt __attribute__((ext_vector_type(2)));
    const vt v = __builtin_nontemporal_load(reinterpret_cast<const vt*>(src + row * (long long)M_of + k));
    return mk<T>(v.x, v.y);
#else
    return src[row * (long long)M_of + k];
#endif
  }
  RF_HD void store(long long row, int n, cplx<T> z, double& s1, double& s2) const {
    z.x *= scale; z.y *= scale;
    stream_store(dst + row * (long long)M_of + n, z);
    s1 += (double)z.x + (double)z.y;
    s2 += (double)z.x * (double)z.x + (double)z.y * (double)z.y;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Fused y + z passes (single GPU).  The y pass of x-plane p needs the whole plane; the z pass of its rows needs the
// whole y output of that plane -- and nothing else.  So instead of writing the y output to HBM and reading it back
// (2 of the pipeline's 5 sweeps), a persistent grid walks the planes: "Y items" (one tile of TC kz-columns of a
// plane) write their output into a small scratch plane, "Z items" (ZR rows of the same plane) read it back while it
// is still in the XCD's L2 / the 256 MB memory-side cache (measured: bouncing every byte of a 4.3 GB sweep through a
// 96 MB ring costs +10 % of the sweep, tools/xbench.hip `m`).
//
// All workgroups that share an XCD (= share an L2, XCC_ID register) form a GROUP with its own ticket counter; a group
// claims planes one at a time from a global counter, so the partition adapts to however the dispatcher placed the
// workgroups.  Tickets are handed out in the order Y(s0) Y(s1) Z(s0) Y(s2) Z(s1) ...; an item only ever waits for items
// with EARLIER tickets of its own group (Z(s) for the NYI Y items of s; Y(s + NS) for the Z items of s, whose scratch
// slot it reuses; a claim for the previous claim), so there is no cycle and a workgroup that holds a ticket is by
// construction running.  Waits are bounded: on a timeout the sticky `error` word is set and every workgroup leaves.
// Memory ordering inside an XCD needs no cache maintenance: a store is acknowledged (vmcnt) by the L2, the signalling
// atomic goes to the same L2, and the consumers read scratch and flags with L1-bypassing accesses.
struct PlaneCtl {
  static constexpr int MAX_GROUPS = 16, RING = 8;
  unsigned next_plane;           // next unclaimed x-plane (all groups)
  unsigned error;                // sticky: a bounded wait ran out
  unsigned pad0[30];
  struct Group {
    unsigned ticket;  unsigned padA[31];
    unsigned claimed; unsigned padB[31];        // number of plane sequence numbers published in plane_of[]
    int plane_of[RING]; unsigned padC[24];      // sequence number s -> x-plane (or -1: no planes left), slot s % RING
    unsigned done_y[RING]; unsigned padD[24];   // cumulative Y items finished per ring slot
    unsigned done_z[RING]; unsigned padE[24];   // cumulative Z items that have finished READING the scratch slot
  } g[MAX_GROUPS];
};

// the XCD (accelerator complex die) this wave runs on: hardware register XCC_ID (id 20), bits 3:0 on gfx942 / gfx950
__device__ __forceinline__ unsigned xcc_id() {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
#else
  return 0;
#endif
}
__device__ __forceinline__ unsigned ctl_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifndef RF_PLANE_SPIN_LIMIT
#define RF_PLANE_SPIN_LIMIT (1u << 20)
#endif
// thread 0 of a workgroup: wait until *p >= want; false (and the sticky error) on timeout or if another workgroup failed
__device__ __forceinline__ bool ctl_wait_ge(const unsigned* p, unsigned want, unsigned* error) {
  for (unsigned spin = 0;; ++spin) {
    if (ctl_load(p) >= want) return true;
    if (spin > RF_PLANE_SPIN_LIMIT || ctl_load(error) != 0u) {
      __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(16);
  }
}

// CY: ColCfg of the y axis (NT threads), CZ: RowCfg of the z rows with the SAME thread count; ZR = rows per Z item
// (a multiple of CZ::NRT).  NS scratch planes per group.  LDS: [tile: max(y tile, z tile)][y twiddles][z twiddles].
template <class CY, class CZ, int ZR, int NS>
__global__ __launch_bounds__(CY::NT, 4) void plane_yz_kernel(cplx<typename CY::T>* __restrict__ W, int nx,
                                                             cplx<typename CY::T>* __restrict__ scratch,
                                                             const cplx<typename CY::T>* __restrict__ tw_y,
                                                             const cplx<typename CY::T>* __restrict__ tw_z,
                                                             typename CY::T scale, double* __restrict__ partials,
                                                             PlaneCtl* __restrict__ ctl, int debug_skip, unsigned* dbg_marker) {
  using T = typename CY::T;
  using cx = cplx<T>;
  using FY = ColFFT<CY, +1, PlaneYIO<T>>;
  using FZ = RowC2R<CZ, PlaneZIO<T>>;
  static_assert(CY::NT == CZ::NT && CY::NPASS == 3 && ZR % CZ::NRT == 0, "plane kernel: configuration");
  constexpr int NY = CY::N, M = CZ::M;
  constexpr int NYI = M / CY::TC, NZI = NY / ZR;             // Y / Z items per plane
  constexpr int TILE = CY::TILE_BYTES > CZ::TILE_BYTES ? CY::TILE_BYTES : CZ::TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  cx* ltw_y = reinterpret_cast<cx*>(rf_smem + TILE);
  cx* ltw_z = reinterpret_cast<cx*>(rf_smem + TILE + CY::TW_BYTES);
  unsigned* mail = reinterpret_cast<unsigned*>(rf_smem);    // 2 words in the (idle) tile area; barriers order the accesses
  const int tid = threadIdx.x;
  for (int i = tid; i < NY; i += CY::NT) ltw_y[i] = tw_y[i];
  for (int i = tid; i < 2 * M; i += CY::NT) ltw_z[i] = tw_z[i];
  const unsigned xcc = xcc_id() & (PlaneCtl::MAX_GROUPS - 1);
  PlaneCtl::Group* grp = &ctl->g[xcc];
  const long long plane_elems = (long long)NY * M;
  cx* my_scratch = scratch + (long long)xcc * NS * plane_elems;
  __syncthreads();
  for (;;) {
    // ---- take a ticket, resolve it to (kind, sequence number, item), wait for what the item depends on ----------
    if (tid == 0) {
      const unsigned t = atomicAdd(&grp->ticket, 1u);
      const unsigned r = t / (unsigned)(NYI + NZI), i = t % (unsigned)(NYI + NZI);
      const bool is_y = i < (unsigned)NYI;
      const int seq = is_y ? (int)r : (int)r - 1;
      const unsigned item = is_y ? i : i - (unsigned)NYI;
      int plane = -2;                                       // -2: skip (Z of sequence -1), -1: no planes left, -3: failed
      if (seq >= 0) {
        const int slot = seq % PlaneCtl::RING;
        bool ok = true;
        if (is_y && item == 0) {                            // this workgroup claims the group's next plane
          ok = ctl_wait_ge(&grp->claimed, (unsigned)seq, &ctl->error);
          if (ok) {
            const unsigned p = atomicAdd(&ctl->next_plane, 1u);
            __hip_atomic_store(&grp->plane_of[slot], p < (unsigned)nx ? (int)p : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the plane number is in the L2 before the count
            __hip_atomic_store(&grp->claimed, (unsigned)seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        ok = ok && ctl_wait_ge(&grp->claimed, (unsigned)seq + 1u, &ctl->error);
        if (ok) {
          plane = __hip_atomic_load(&grp->plane_of[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (plane >= 0) {
            if (is_y) {                                      // the scratch slot was last read by the Z items of seq - NS
              if (seq >= NS) ok = ctl_wait_ge(&grp->done_z[(seq - NS) % PlaneCtl::RING], (unsigned)NZI * (unsigned)((seq - NS) / PlaneCtl::RING + 1), &ctl->error);
            } else {
              ok = ctl_wait_ge(&grp->done_y[slot], (unsigned)NYI * (unsigned)(seq / PlaneCtl::RING + 1), &ctl->error);
            }
          }
        }
        if (!ok) plane = -3;
      }
      mail[0] = (unsigned)plane;
      mail[1] = (is_y ? 0x80000000u : 0u) | ((unsigned)(seq < 0 ? 0 : seq) << 12) | item;
    }
    __syncthreads();
    // (workgroup-uniform: told to the compiler, so that the branches below are scalar branches around the barriers)
    const int plane = __builtin_amdgcn_readfirstlane((int)mail[0]);
    const unsigned word = (unsigned)__builtin_amdgcn_readfirstlane((int)mail[1]);
    __syncthreads();                                         // the mail words live in the tile area
    const bool is_y = (word >> 31) != 0u;
    const int seq = (int)((word >> 12) & 0x7FFFFu), item = (int)(word & 0xFFFu);
    if (plane == -3) break;                                  // a wait failed somewhere: leave
    if (plane == -1 && !is_y) break;                         // tickets are ordered: after the first idle Z item nothing is left
    if (plane < 0) continue;                                 // Z items of sequence -1, idle Y items
    cx* Wp = W + (long long)plane * plane_elems;
    cx* Sp = my_scratch + (long long)(seq % NS) * plane_elems;
    const int slot = seq % PlaneCtl::RING;
    if (dbg_marker != nullptr && tid == 0) dbg_marker[blockIdx.x] = word;
    int tl = tid;
    asm volatile("" : "+v"(tl));
    __builtin_assume(tl >= 0 && tl < CY::NT);
    if ((is_y && (debug_skip & 1)) || (!is_y && (debug_skip & 2))) {   // development: this role only signals
      __syncthreads();
      if (tid == 0) atomicAdd(is_y ? &grp->done_y[slot] : &grp->done_z[slot], 1u);
    } else
    if (is_y) {
      PlaneYIO<T> io;
      io.src = Wp; io.dst = Sp; io.g = ColGeom{(long long)M, 0, (long long)M};
      FY::pass_first(tl, item, io, lds);
      typename FY::Regs r;
      __syncthreads();
      FY::pass_mid_read(tl, ltw_y, lds, r);
      __syncthreads();
      FY::pass_mid_write(tl, lds, r);
      __syncthreads();
      FY::pass_last(tl, item, io, ltw_y, lds);
      __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this thread's scratch stores are in the L2
      __syncthreads();
      if (tid == 0) atomicAdd(&grp->done_y[slot], 1u);
    } else {
      PlaneZIO<T> io;
      io.src = Sp; io.dst = Wp; io.scale = scale; io.M_of = M;
      typename FZ::Regs r;
      double s1 = 0, s2 = 0;
#pragma unroll 1
      for (int sub = 0; sub < ZR / CZ::NRT; ++sub) {
        const long long tile = (long long)item * (ZR / CZ::NRT) + sub;
        FZ::pass_first(tl, tile, NY, io, ltw_z, lds, r);     // (resets r.s1, r.s2)
        if (sub == ZR / CZ::NRT - 1) {                        // every scratch read of this item has returned its data
          __syncthreads();
          if (tid == 0) atomicAdd(&grp->done_z[slot], 1u);
        } else {
          __syncthreads();
        }
        FZ::pass_mid_read(tl, ltw_z, lds, r);
        __syncthreads();
        FZ::pass_mid_write(tl, lds, r);
        __syncthreads();
        FZ::pass_last(tl, tile, NY, io, ltw_z, lds, r);
        s1 += r.s1; s2 += r.s2;
        __syncthreads();
      }
      // workgroup reduction of the moments of this item: wave shuffle, then one slot per wave in the (idle) tile area
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
      }
      double* red = reinterpret_cast<double*>(rf_smem);
      const int wave = tid >> 6, lane = tid & 63;
      if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; }
      __syncthreads();
      if (tid == 0) {
        double a = 0, b = 0;
#pragma unroll
        for (int w = 0; w < CY::NT / 64; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
        const long long slot_p = (long long)plane * NZI + item;
        partials[2 * slot_p] = a;
        partials[2 * slot_p + 1] = b;
      }
      __syncthreads();
    }
  }
}



// ---------------------------------------------------------------------------------------------------------------
// v2: the same plane pipeline WITHOUT persistent workgroups and tickets.  One workgroup per item; blockIdx b belongs to
// group b % 8 (blocks b and b + 8 share an XCD: observed dispatch rule, VERIFIED at run time through XCC_ID -- a
// mismatch sets the sticky error) and q = b / 8 is its position in the group's item order Y(s0) Y(s1) Z(s0) Y(s2)
// Z(s1) ...; group g owns the planes g, g + 8, g + 16, ...  An item waits (bounded) for items with smaller q of its
// own group only: Z(s) for the NYI Y items of s, Y(s + NS) for the NZI Z items of s.  Workgroups are dispatched in
// blockIdx order, so the producers of a waiting item hold their slots already.  Fresh workgroups per item keep what
// made the separate passes fast (a finished workgroup's stores drain while the next one starts; no vmcnt ordering
// between items), at the price of staging the twiddle tables (16 KB from L2) per item.
struct PlaneCtl2 {
  static constexpr int GROUPS = 8, MAXSEQ = 512;
  unsigned error;
  unsigned pad0[31];
  unsigned group_xcc[GROUPS];    // XCC_ID + 1 of the first workgroup of each group (0 = not yet registered)
  unsigned pad1[24];
  unsigned done_y[GROUPS][MAXSEQ];
  unsigned done_z[GROUPS][MAXSEQ];
};

template <class CY, class CZ, int ZR, int NS>
__global__ __launch_bounds__(CY::NT, 4) void plane_yz_kernel_v2(cplx<typename CY::T>* __restrict__ W, int nx,
                                                                cplx<typename CY::T>* __restrict__ scratch,
                                                                const cplx<typename CY::T>* __restrict__ tw_y,
                                                                const cplx<typename CY::T>* __restrict__ tw_z,
                                                                typename CY::T scale, double* __restrict__ partials,
                                                                PlaneCtl2* __restrict__ ctl, int debug_skip) {
  using T = typename CY::T;
  using cx = cplx<T>;
  using FY = ColFFT<CY, +1, PlaneYIO<T>>;
  using FZ = RowC2R<CZ, PlaneZIO<T>>;
  static_assert(CY::NT == CZ::NT && CY::NPASS == 3 && ZR % CZ::NRT == 0, "plane kernel: configuration");
  constexpr int NY = CY::N, M = CZ::M;
  constexpr int NYI = M / CY::TC, NZI = NY / ZR;
  constexpr int TILE = CY::TILE_BYTES > CZ::TILE_BYTES ? CY::TILE_BYTES : CZ::TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  cx* ltw_y = reinterpret_cast<cx*>(rf_smem + TILE);
  cx* ltw_z = reinterpret_cast<cx*>(rf_smem + TILE + CY::TW_BYTES);
  unsigned* mail = reinterpret_cast<unsigned*>(rf_smem);
  const int tid = threadIdx.x;
  const unsigned g = blockIdx.x & 7u, q = blockIdx.x >> 3;
  const unsigned r = q / (unsigned)(NYI + NZI), i = q % (unsigned)(NYI + NZI);
  const bool is_y = i < (unsigned)NYI;
  const int seq = is_y ? (int)r : (int)r - 1;
  const int item = (int)(is_y ? i : i - (unsigned)NYI);
  if (seq < 0) return;                                              // Z items of sequence -1
  const long long plane = (long long)g + 8LL * seq;
  if (plane >= nx) return;                                          // idle tail
  // stage only the table this item needs
  if (is_y) { for (int k = tid; k < NY; k += CY::NT) ltw_y[k] = tw_y[k]; }
  else      { for (int k = tid; k < 2 * M; k += CY::NT) ltw_z[k] = tw_z[k]; }
  if (tid == 0) {
    bool ok = true;
    const unsigned xcc = xcc_id() + 1u;
    const unsigned prev = atomicCAS(&ctl->group_xcc[g], 0u, xcc);
    if (prev != 0u && prev != xcc) ok = false;                      // this group spans two XCDs: its scratch is not coherent
    if (ok) {
      if (is_y) { if (seq >= NS) ok = ctl_wait_ge(&ctl->done_z[g][seq - NS], (unsigned)NZI, &ctl->error); }
      else ok = ctl_wait_ge(&ctl->done_y[g][seq], (unsigned)NYI, &ctl->error);
    }
    if (!ok) __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mail[0] = ok ? 1u : 0u;
  }
  __syncthreads();
  const unsigned go = (unsigned)__builtin_amdgcn_readfirstlane((int)mail[0]);
  __syncthreads();
  if (!go) return;
  const long long plane_elems = (long long)NY * M;
  cx* Wp = W + plane * plane_elems;
  cx* Sp = scratch + ((long long)g * NS + (seq % NS)) * plane_elems;
  const int tl = tid;
  if ((is_y && (debug_skip & 1)) || (!is_y && (debug_skip & 2))) {
    if (tid == 0) atomicAdd(is_y ? &ctl->done_y[g][seq] : &ctl->done_z[g][seq], 1u);
    return;
  }
  if (is_y) {
    PlaneYIO<T> io;
    io.src = Wp; io.dst = Sp; io.g = ColGeom{(long long)M, 0, (long long)M};
    FY::pass_first(tl, item, io, lds);
    typename FY::Regs rr;
    __syncthreads();
    FY::pass_mid_read(tl, ltw_y, lds, rr);
    __syncthreads();
    FY::pass_mid_write(tl, lds, rr);
    __syncthreads();
    FY::pass_last(tl, item, io, ltw_y, lds);
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0): this thread's scratch stores are in the L2
    __syncthreads();
    if (tid == 0) atomicAdd(&ctl->done_y[g][seq], 1u);
  } else {
    PlaneZIO<T> io;
    io.src = Sp; io.dst = Wp; io.scale = scale; io.M_of = M;
    typename FZ::Regs rr;
    double s1 = 0, s2 = 0;
#pragma unroll 1
    for (int sub = 0; sub < ZR / CZ::NRT; ++sub) {
      const long long tile = (long long)item * (ZR / CZ::NRT) + sub;
      FZ::pass_first(tl, tile, NY, io, ltw_z, lds, rr);
      __syncthreads();
      if (sub == ZR / CZ::NRT - 1 && tid == 0) atomicAdd(&ctl->done_z[g][seq], 1u);   // all scratch reads have returned
      FZ::pass_mid_read(tl, ltw_z, lds, rr);
      __syncthreads();
      FZ::pass_mid_write(tl, lds, rr);
      __syncthreads();
      FZ::pass_last(tl, tile, NY, io, ltw_z, lds, rr);
      s1 += rr.s1; s2 += rr.s2;
      __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s1 += __shfl_down(s1, off);
      s2 += __shfl_down(s2, off);
    }
    double* red = reinterpret_cast<double*>(rf_smem);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; }
    __syncthreads();
    if (tid == 0) {
      double a = 0, b = 0;
#pragma unroll
      for (int w = 0; w < CY::NT / 64; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
      const long long slot_p = plane * NZI + item;
      partials[2 * slot_p] = a;
      partials[2 * slot_p + 1] = b;
    }
  }
}

}  // namespace rf
