// rf_fft.h -- LDS-staged Stockham FFT passes of the 3-D c2r / r2c transform.
//
// Two pass shapes, both written as *phase functions* separated by workgroup
// barriers so that the HIP kernels (rf_kernels.hip) and the CPU emulator
// (emu/rf_emu.cpp) run the very same code:
//
//   ColFFT : FFT along a strided axis (x or y).  A workgroup owns a tile of TC
//            adjacent columns (kz fastest => every global access is a 16-byte
//            vector covering CPL adjacent columns, tile rows are TC*8 B
//            contiguous segments).  Pass 1 goes global -> registers -> LDS,
//            the last pass LDS -> registers -> global, so the tile crosses LDS
//            only NPASS-1 times.  LDS image: [padded row][TC] (t fastest); the middle
//            (LDS-only) pass works one column per lane so that all threads stay busy.
//   RowFFT : c2r (or r2c) along the contiguous z axis.  A workgroup owns NRT
//            whole rows of M = nz/2 complex.  The Hermitian untangle is folded
//            into pass 1: one thread owns the butterfly pair (j, M/R1 - j),
//            whose inputs are each other's mirror images.
//
// Device-internal layout: W[nx][ny][nz/2] complex == [nx][ny][nz]
// real; slot kz=0 of the complex view carries (kz=0 plane) + i (kz=nz/2 plane).
#pragma once
#include <type_traits>
#include "rf_core.h"
#include "rf_exp2_tab.h"

namespace rf {

constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <typename T> struct alignas(16) V16 {
  static constexpr int CPL = 16 / (int)sizeof(cplx<T>);
  cplx<T> c[CPL];
};
// One 16-byte global access per V16: through a 4 x 32-bit vector type, so that the compiler cannot split it into two
// 8-byte accesses when the halves sit in non-adjacent registers (it did: half-width stores, twice as many of them).
// The pointer is cast to the GLOBAL address space: the arrays reach the kernels inside by-value structs, the compiler
// could not prove where they point and emitted flat_load / flat_store (64-bit VGPR addresses, counted on the LDS
// counter as well) -- with the cast they are global_load / global_store with an SGPR base and a 32-bit lane offset.
template <typename T> RF_HD void v16_store(void* p, const V16<T>& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) u4 gu4;
  union { V16<T> s; u4 q; } u;
  u.s = v;
  *(gu4*)p = u.q;
#else
  *reinterpret_cast<V16<T>*>(p) = v;
#endif
}
template <typename T> RF_HD V16<T> v16_load(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) u4 gu4;
  union { V16<T> s; u4 q; } u;
  u.q = *(const gu4*)p;
  return u.s;
#else
  return *reinterpret_cast<const V16<T>*>(p);
#endif
}

// ---------------------------------------------------------------------------
// Column pass configuration
// ---------------------------------------------------------------------------
template <typename T_, int N_, int R1_, int R2_, int R3_, int TC_, int NT_>
struct ColCfg {
  using T = T_;
  static constexpr int N = N_, R1 = R1_, R2 = R2_, R3 = R3_, TC = TC_, NT = NT_;
  static_assert(R1_ * R2_ * R3_ == N_, "radices must multiply to N");
  static constexpr int CPL = V16<T>::CPL;        // columns per lane in the global-memory passes (16 B per lane)
  static constexpr int LPR = TC / CPL;           // lanes per tile row in those passes
  static constexpr int BPI = NT / LPR;           // butterflies (per column) per iteration, first / last pass
  static constexpr int BPM = NT / TC;            // butterflies per iteration in the middle pass (1 column per lane)
  static_assert(NT_ % TC_ == 0, "NT must be a multiple of the tile width");
  static constexpr int NPASS = (R2 == 1 ? 1 : (R3 == 1 ? 2 : 3));
  static constexpr int RL = (NPASS == 1 ? R1 : (NPASS == 2 ? R2 : R3));  // radix of the last pass
  // LDS image [row ^ swizzle][TC] -- no padding: bit 0 of the row is flipped by bit log2(R1), so the
  // stride-R1 rows written by pass 1 alternate between the two 128-byte bank halves while runs of
  // consecutive rows stay conflict free.  Behind the tile: the twiddle table, then the IO's tables.
  static constexpr int TILE_BYTES = (NPASS == 1 ? 0 : N * TC * (int)sizeof(cplx<T>));
  static constexpr int TW_BYTES = (NPASS == 1 ? 0 : N * (int)sizeof(cplx<T>));
  static constexpr int LDS_BYTES = TILE_BYTES + TW_BYTES;      // + IO::LDS_EXTRA, added by the launcher
  static constexpr int IT1 = ceil_div(N / R1, BPI);
  static constexpr int IT2 = (NPASS == 3 ? ceil_div(N / R2, BPM) : 1);
  static constexpr int ITL = ceil_div(N / RL, BPI);
  RF_HD static int prow(int r) { return r ^ ((r / R1) & 1); }
  // most butterflies per column of any global-memory pass (first: N / R1, last: N / RL): bounds the lane offsets
  static constexpr int LMAX = N / (R1 < RL ? R1 : RL);
};

// Addressing of a column pass over the packed device array.
//   flattened column C in [0, ncols); element (row, C) lives at
//   (C / inner) * outer_stride + (C % inner) + row * row_stride   (complex units; inner is a power of two)
// The IO calls name a cell as (C0, cl, rb, ro): column C0 + cl with C0 the first column of the workgroup's tile
// (a tile never straddles `inner`), row rb + ro with rb the lane's butterfly index and ro = m * L the same for
// all lanes.  So an address splits into a workgroup-uniform 64-bit part (scalar ALU; it becomes the SGPR base of
// the memory instruction) and a lane part that does not depend on m and fits 32 bits: ONE address VGPR serves all
// R accesses of a butterfly (64-bit lane addresses cost two VGPRs and a 64-bit add per access).  WIDE (a template
// flag of the IO) = the lane part may exceed 32 bits of bytes: only float64 passes of length 2048 over 16-MiB-plus
// row strides need it (needs_wide(), checked by the launchers).
struct ColGeom {
  long long inner, outer_stride, row_stride;
  // Optional further levels (the blocked, transposed intermediate of the c2r transform).  With
  // hi = C / inner, lo = C % inner the element (row, C) lives at
  //     (hi >> hi_shift) * hi_stride + (hi & (2^hi_shift - 1)) * outer_stride
  //   + (lo >> sub_shift) * sub_stride + (lo & (2^sub_shift - 1))
  //   + (row >> row_shift) * row_hi_stride + (row & (2^row_shift - 1)) * row_stride.
  // The defaults (sub_shift = 0, sub_stride = 1, hi_shift = 62, row_shift = 30) are the plain form above.  A tile is at most
  // 2^sub_shift columns wide when sub_shift > 0, and the uniform row offsets ro of a pass are multiples of 2^row_shift
  // whenever row_shift < 30 (so the row splits of rb and ro add without a carry): the launchers check both.
  int sub_shift = 0;
  long long sub_stride = 1;
  int hi_shift = 62;
  long long hi_stride = 0;
  int row_shift = 30;
  long long row_hi_stride = 0;
  // (tiles start at multiples of their width, and widths and `inner` are powers of two: either a tile lies inside
  // one run of `inner` columns, or it covers whole runs and the lane's column offset cl selects the run)
  RF_HD int inner_shift() const { return 63 - __builtin_clzll((unsigned long long)inner); }
  RF_HD long long uniform_part(long long C0, int ro) const {
    const long long lo = C0 & (inner - 1), hi = C0 >> inner_shift();
    return (hi >> hi_shift) * hi_stride + (hi & ((1LL << hi_shift) - 1)) * outer_stride + (lo >> sub_shift) * sub_stride +
           (lo & ((1LL << sub_shift) - 1)) + (long long)(ro >> row_shift) * row_hi_stride + (long long)(ro & ((1 << row_shift) - 1)) * row_stride;
  }
  RF_HD uint32_t lane_part(int cl, int rb) const {
    const uint32_t lo = (uint32_t)cl & (uint32_t)(inner - 1);
    return (uint32_t)(cl >> inner_shift()) * (uint32_t)outer_stride + (lo >> sub_shift) * (uint32_t)sub_stride + (lo & ((1u << sub_shift) - 1u)) +
           (uint32_t)(rb >> row_shift) * (uint32_t)row_hi_stride + ((uint32_t)rb & ((1u << row_shift) - 1u)) * (uint32_t)row_stride;
  }
  RF_HD long long lane_part_wide(int cl, int rb) const {
    const long long lo = (long long)cl & (inner - 1);
    return (long long)(cl >> inner_shift()) * outer_stride + (lo >> sub_shift) * sub_stride + (lo & ((1LL << sub_shift) - 1)) +
           (long long)(rb >> row_shift) * row_hi_stride + (long long)(rb & ((1 << row_shift) - 1)) * row_stride;
  }
  // does the lane part of a pass with L butterflies per column and TC columns per tile need 64 bits?
  bool needs_wide(int L, int TC, int elem_bytes) const {
    const unsigned long long runs = inner < TC ? (unsigned long long)(TC / inner) : 0;
    const unsigned long long subs = sub_shift > 0 ? (unsigned long long)((TC - 1) >> sub_shift) : 0;
    const unsigned long long rlo = (unsigned long long)(L - 1) < (1ull << row_shift) ? (unsigned long long)(L - 1) : (1ull << row_shift) - 1;
    return (((unsigned long long)(L - 1) >> row_shift) * (unsigned long long)row_hi_stride + rlo * (unsigned long long)row_stride +
            runs * (unsigned long long)outer_stride + subs * (unsigned long long)sub_stride + (unsigned long long)TC) * (unsigned long long)elem_bytes >= (1ull << 32);
  }
  // are the uniform row offsets of a pass (multiples of L, or every row when the pass is the only one) compatible with the row split?
  bool rows_ok(int L, int npass) const { return row_shift >= 30 || (npass >= 2 && L % (1 << row_shift) == 0); }
  template <bool WIDE, typename E> RF_HD E* at(E* base, long long C0, int cl, int rb, int ro) const {
    E* ub = base + uniform_part(C0, ro);
    if (WIDE) return ub + lane_part_wide(cl, rb);
    return reinterpret_cast<E*>((size_t)ub + (size_t)(lane_part(cl, rb) * (uint32_t)sizeof(E)));
  }
};

}  // namespace rf

// the pass families (each header includes this one first: include guards make the order irrelevant)
#include "rf_fft_gen.h"
#include "rf_fft_col.h"
#include "rf_fft_row.h"
