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

template <typename T, bool WIDE = false> struct PlainColIO {
  cplx<T>* base;
  ColGeom g;
  RF_HD V16<T> load(long long C0, int cl, int rb, int ro) const { return v16_load<T>(g.at<WIDE>(base, C0, cl, rb, ro)); }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<T>& v) const { v16_store<T>(g.at<WIDE>(base, C0, cl, rb, ro), v); }
  static constexpr int FIX_MODE = 0;
  RF_HD bool needs_fix(long long) const { return false; }
  RF_HD cplx<T> fix_value(long long, int, int) const { return cplx<T>(); }
  static constexpr int LDS_EXTRA = 0;
  RF_HD void prologue(int, int, void*) {}
  RF_HD void bind_seed() {}
  RF_HD static void sched_fence(int = 0) {}      // loads of one butterfly are meant to be issued back to back
  static constexpr bool ROLLED_LOAD = false;
  RF_HD long long remap_tile(long long t) const { return t; }
  static constexpr bool HAS_FINISH = false;
};

// y pass of the c2r transform that also accumulates  S = sum over its OUTPUT Y(x, y, kz) of w(kz) |Y|^2  (w = 1 for slot kz = 0,
// which holds the two REAL planes kz = 0 and nz/2 as A0 + i Anyq, so |slot|^2 = A0^2 + Anyq^2; w = 2 for every other kz: its
// conjugate half of k space).  Y is the unnormalised inverse transform over (kx, ky), so by Parseval S = nx ny sum_k |delta_k|^2
// over the FULL k space, and for the real field delta(x) = (1 / N3) sum_k delta_k e^{ikx}, N3 = nx ny nz:
//     sum_x delta(x)^2 = S / (nx ny N3),    mean = 0 (the DC mode is 0)    =>    rms = sqrt(S / (nx ny)) / N3
// -- the field's rms is known BEFORE the z pass runs, so that pass can apply the lognormal map (cosmotools.py:206-221,
// generate.py:266-273) in its epilogue instead of two more sweeps and a host round trip.  One partial per workgroup (tile),
// float64, fixed order: deterministic.
template <typename T> struct AccColIO {
  cplx<T>* base;
  ColGeom g;
  double* partials;              // [ntiles]
  int kz0, nzl;                  // the kz planes of this rank's columns: column C = hi * nzl + (kz - kz0)
  // A lane stores the same CPL columns in every call, so the weight is a property of the lane: the squares are summed unweighted per
  // column (two fused multiply-adds per complex; the weighted form cost a multiply, a select and an add more, 2 x 10^9 times per
  // 1024^3 field) and weighted once in weighted_sum()
  mutable double accs[V16<T>::CPL] = {};
  mutable bool first_is_dc = false;      // the lane's first column is the slot kz = 0 (weight 1); every other column has weight 2
  RF_HD V16<T> load(long long C0, int cl, int rb, int ro) const { return v16_load<T>(g.at<false>(base, C0, cl, rb, ro)); }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<T>& v) const {
    first_is_dc = kz0 + (int)((C0 + cl) & (long long)(nzl - 1)) == 0;
#pragma unroll
    for (int c = 0; c < V16<T>::CPL; ++c) {
      const double re = (double)v.c[c].x, im = (double)v.c[c].y;
      accs[c] = __builtin_fma(im, im, __builtin_fma(re, re, accs[c]));
    }
    v16_store<T>(g.at<false>(base, C0, cl, rb, ro), v);
  }
  RF_HD double weighted_sum() const {
    double a = (first_is_dc ? 1.0 : 2.0) * accs[0];
#pragma unroll
    for (int c = 1; c < V16<T>::CPL; ++c) a += 2.0 * accs[c];      // (columns kz + 1 ...: never the slot kz = 0, whose kz is even)
    return a;
  }
  static constexpr int FIX_MODE = 0;
  RF_HD bool needs_fix(long long) const { return false; }
  RF_HD cplx<T> fix_value(long long, int, int) const { return cplx<T>(); }
  static constexpr int LDS_EXTRA = 0;
  RF_HD void prologue(int, int, void*) {}
  RF_HD void bind_seed() {}
  RF_HD static void sched_fence(int = 0) {}
  static constexpr bool ROLLED_LOAD = false;
  RF_HD long long remap_tile(long long t) const { return t; }
  static constexpr bool HAS_FINISH = true;
  // workgroup sum of `acc` -> partials[tile]; `red` = NT / 64 doubles of LDS the workgroup no longer needs, `sync` = its barrier
  template <class Sync> RF_HD void finish(int tid, int nthreads, double* red, long long tile, double wave_sum, Sync sync) const {
    if ((tid & 63) == 0) red[tid >> 6] = wave_sum;
    sync();
    if (tid == 0) {
      double a = 0;
      for (int w = 0; w < nthreads / 64; ++w) a += red[w];
      partials[tile] = a;
    }
  }
};

// Strided pass with separate load and store geometries and its own tile order: the y pass of the c2r transform on the blocked
// intermediate X (in place: src == base, gs == g = xblock_y_geom), DESIGN.md section 3.8.
template <typename T> struct XposeColIO {
  const cplx<T>* src;
  ColGeom gs;
  cplx<T>* base;
  ColGeom g;
  // Order of the tiles.  A tile is (hi = ix, kz tile kt) with logical index hi * tiles_per_run + kt (columns C = hi * nzl + kz, as
  // in the plain layout).  In X the tiles of neighbouring ix are neighbouring tc-cell segments (two of them share a 128-byte
  // line when tc cells are 64 bytes) and the kz tiles of one ix are whole blocks apart, so in dispatch order ix is the fast
  // index: t = ((kg * nhi + hi) << grp_shift) + kl  ->  hi * tiles_per_run + (kg << grp_shift) + kl  (grp_shift = 0 in the product).
  int grp_shift = 0, nhi_shift = 0, tpr_shift = 0;
  static constexpr bool HAS_FINISH = false;
  RF_HD long long remap_tile(long long t) const {
    const long long kl = t & ((1LL << grp_shift) - 1), r = t >> grp_shift;
    const long long hi = r & ((1LL << nhi_shift) - 1), kg = r >> nhi_shift;
    return (hi << tpr_shift) + (kg << grp_shift) + kl;
  }
  RF_HD V16<T> load(long long C0, int cl, int rb, int ro) const { return v16_load<T>(gs.at<false>(src, C0, cl, rb, ro)); }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<T>& v) const { v16_store<T>(g.at<false>(base, C0, cl, rb, ro), v); }
  static constexpr int FIX_MODE = 0;
  RF_HD bool needs_fix(long long) const { return false; }
  RF_HD cplx<T> fix_value(long long, int, int) const { return cplx<T>(); }
  static constexpr int LDS_EXTRA = 0;
  RF_HD void prologue(int, int, void*) {}
  RF_HD void bind_seed() {}
  RF_HD static void sched_fence(int = 0) {}
  static constexpr bool ROLLED_LOAD = false;
};

// The blocked intermediate X of the c2r transform: [x block xb][kz tile kt][iy][rb rows of x][tc columns],
//   cell (ix, iy, kz)  at  ((xb * nkt + kt) * ny + iy) * rb * tc + (ix % rb) * tc + kz % tc,   xb = ix / rb, kt = kz / tc.
// * the x pass's tile (all nx rows of tc adjacent kz of one iy) is nx / rb contiguous chunks of rb * tc cells: whole 128-byte
//   lines (its stores into the plain layout are 64-byte half lines 4 MiB apart);
// * the y pass runs IN PLACE on X: its tile (all ny rows of tc kz of one ix) is tc-cell segments rb * tc cells apart inside one
//   block of ny * rb * tc cells -- with rb = 64 and 8-byte cells the 4-KiB stride and 4-MiB span of the plain layout;
// * the z pass gathers: its NRT rows are consecutive ix of one (xb, iy), so for every kz tile they are ONE contiguous chunk of
//   NRT * tc cells, and it writes the dense rows of W (XGatherRowIO);
// * an x block is contiguous: the y / z slabs of RF_FLAG_YZ_SLAB_PLANES are whole blocks.
inline int ilog2ll(long long v) { return 63 - __builtin_clzll((unsigned long long)v); }
inline ColGeom xblock_x_geom(long long nx, long long ny, long long nzl, long long tc, long long rb) {   // x pass: C = iy * nzl + kz, row = ix
  ColGeom g{nzl, rb * tc, tc};
  g.sub_shift = ilog2ll(tc); g.sub_stride = ny * rb * tc;
  if (rb < nx) { g.row_shift = ilog2ll(rb); g.row_hi_stride = (nzl / tc) * ny * rb * tc; }
  return g;
}
inline ColGeom xblock_y_geom(long long nx, long long ny, long long nzl, long long tc, long long rb) {   // y pass: C = ix * nzl + kz, row = iy
  ColGeom g{nzl, tc, rb * tc};
  g.sub_shift = ilog2ll(tc); g.sub_stride = ny * rb * tc;
  if (rb < nx) { g.hi_shift = ilog2ll(rb); g.hi_stride = (nzl / tc) * ny * rb * tc; }
  return g;
}

// dispatch order of XposeColIO's tiles: nhi values of the slow index (powers of two), tiles_per_run kz tiles each
template <class IO> inline void set_xpose_order(IO& io, long long nhi, long long tiles_per_run) {
  io.nhi_shift = 63 - __builtin_clzll((unsigned long long)nhi);
  io.tpr_shift = 63 - __builtin_clzll((unsigned long long)tiles_per_run);
  io.grp_shift = 0;                 // (kz tiles of one ix dispatched in groups of 2^grp_shift: measured, no gain -- DESIGN_HISTORY.md)
}

// x pass fused with generation (rows K,T,R,S): load() synthesises the packed
// k-space cell instead of reading memory.  Columns are the flattened (iy, kz).
// If `kspace` is non-null the cell is read from an API-layout array
// [nx][ny][nz/2+1] instead (unfused c2r of uploaded / separately generated data).
template <typename T, bool WIDE = false> struct GenColIO {
  cplx<T>* base;           // destination W
  ColGeom g;               // x-pass geometry: inner = ny*nzc, row_stride = ny*nzc
  GenParams gp;
  const cplx<T>* kspace;   // optional source in API layout
  int kz0, nzl;            // this rank's kz slab [kz0, kz0 + nzl) of the nz/2 packed planes
  RF_HD V16<T> load(long long C0, int cl, int rb, int ro) const {
    V16<T> v;
    const long long C = C0 + cl;
    const int nzc = gp.nz / 2;
    const uint64_t seed = gp.seed;
    const int ix = rb + ro;
#pragma unroll
    for (int c = 0; c < V16<T>::CPL; ++c) {
      const long long Cc = C + c;
      const int iy = (int)(Cc / nzl), kz = kz0 + (int)(Cc % nzl);
      if (kspace) {
        const cplx<T>* p = kspace + ((long long)ix * gp.ny + iy) * gp.zpitch;      // rows of this rank's planes + Nyquist
        cplx<T> a = p[kz - gp.zoff];
        if (kz == 0) {
          // The planes kz = 0 and kz = nz/2 travel as ONE complex plane (a + i n), which needs both to be 2-D
          // Hermitian.  np.fft.irfftn (transform.py:314) accepts anything there and, by discarding the imaginary part
          // after the x and y transforms, in effect uses the Hermitian part of each plane: so that is what is packed.
          // (Hermitian input, e.g. after symmetrize(), is reproduced bit for bit: (a + conj a*)/2 with a == conj a*.)
          const int mx = (gp.nx - ix) % gp.nx, my = (gp.ny - iy) % gp.ny;
          const cplx<T>* pm = kspace + ((long long)mx * gp.ny + my) * gp.zpitch;
          const cplx<T> am = pm[0], n0 = p[gp.zpitch - 1], nm = pm[gp.zpitch - 1];
          const cplx<T> ah = mk<T>((T)0.5 * (a.x + am.x), (T)0.5 * (a.y - am.y));
          const cplx<T> nh = mk<T>((T)0.5 * (n0.x + nm.x), (T)0.5 * (n0.y - nm.y));
          a = mk<T>(ah.x - nh.y, ah.y + nh.x);
        }
        v.c[c] = a;
      } else {
        v.c[c] = gen_packed<T>(gp, seed, ix, iy, kz);
      }
    }
    return v;
  }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<T>& v) const {
    v16_store<T>(g.at<WIDE>(base, C0, cl, rb, ro), v);
  }
  static constexpr int FIX_MODE = 0;
  RF_HD bool needs_fix(long long) const { return false; }
  RF_HD cplx<T> fix_value(long long, int, int) const { return cplx<T>(); }
  static constexpr int LDS_EXTRA = 0;
  RF_HD void prologue(int, int, void*) {}
  // the kernel calls this once before any load(): a seed kept in device memory (graph replay) is read once, through
  // the scalar unit, instead of once per cell
  RF_HD void bind_seed() { if (gp.seed_dev) { gp.seed = pin_uniform(*gp.seed_dev); gp.seed_dev = nullptr; } }
  RF_HD static void sched_fence(int = 0) {}
  // the exact-chain generation body (float64 lookups, libm-grade log10 / sin / cos) is far too big to be
  // replicated R times: the load loop stays rolled and parks its values in the thread's own LDS slots
  static constexpr bool ROLLED_LOAD = true;
  RF_HD long long remap_tile(long long t) const { return t; }
  static constexpr bool HAS_FINISH = false;
};

// x pass fused with the fast float32 native generation (one Philox call per lane load)
// SLAB: 1 = only rows [x0, x1) are stored (replicated-generation mode); a separate instantiation so that the
// guard costs the ordinary kernel nothing.
// FIX: 1 = this kernel repairs the kz = 0 slot itself (the owning lane, rolled loop through LDS: short passes, and the emulator's
// reference form); 3 = it takes the repaired slots from a side buffer [ny][nx] that fix_fill_kernel (rf_kernels.h: one thread per
// mode, every lane busy, the same fix_value() arithmetic) has filled just before -- 8 extra loads per owning lane instead of two
// Philox calls, Box-Muller pairs and sigma lookups per row in a kernel that then needs 128 - 244 registers and runs its tiles
// 2.5 - 19x slower than an ordinary one (rounds 1 - 3: FIX = 1, then FIX = 2 = the values computed by all lanes in a phase of
// their own); 0 = it does not repair (the tiles that hold kz = 0 are run by a FIX = 1 / 3 launch first).
// POT: 2 = the pass transforms pscale * delta(k) / k^2 instead of delta(k) (the saved potential regenerated on demand: each
// cell rounded as the stored one and its scaled copy would be); 1 = the pass also stores delta(k) / k^2 (0 at DC) of every generated cell into `pot`, an API-layout array
// [nx][ny][nz/2+1] -- the save_potential=True branch of generate_delta_field (generate.py:200-217) without ever
// materialising delta(k) itself.  (Rows of nz/2+1 cells are only 8-byte aligned: two 8-byte stores per lane.)
// SRC: 0 = native Philox + Box-Muller draws; 1 = deviates resident in device memory as float64 (2 = as float32 pairs;
// the reference's numpy stream,
// rng='reference'): same float32 |k| and sigma arithmetic, the draw replaced by two 16-byte loads per lane.  The field
// then differs from the exact-chain kernel's by the float32 sigma rounding only (<= 1e-6 relative, far inside the
// 1e-5 * rms parity tolerance) and the pass is HBM-bound (12.9 GB) instead of latency-bound on table lookups.
// XS: 1 = row r of the pass is mode ix = r; 2 = the pass is one HALF of a transform of twice its length (Col2 below: rows of the
// even / odd modes ix = 2 r + xp, xp = the phase set_phase() selects) -- native generation without the potential store only.
template <int FIX = 1, int SLAB = 0, int POT = 0, int SRC = 0, int XS = 1>
struct FastGenColIOT {
  static_assert(XS == 1 || (XS == 2 && SLAB == 0 && POT != 1 && SRC != 1), "half-transform rows: native generation or float32 deviate pairs, no potential store");
  static constexpr int NOISE_SRC = SRC;      // (0 native, 1 float64 deviates, 2 float32 pairs in the replay's runs)
  int xp = 0;
  RF_HD void set_phase(int p) { xp = p; }
  cplx<float>* base;
  ColGeom g;
  FastGenParams gp;
  cplx<float>* pot = nullptr;
  int kz0, nzl;
  int x0 = 0, x1 = 1 << 30;  // replicated-generation mode (multi-GPU without an exchange): only rows [x0, x1) are stored,
                             // and `base` has been moved back by x0 rows so that row x0 lands on the local array's row 0
  RF_HD int nzl_shift() const { return 31 - __builtin_clz((unsigned)nzl); }
  const FastRec* rec;      // set by prologue(): LDS copy of the sigma records (or the global one)
  static constexpr int LDS_EXTRA = FAST_LDS_BINS * (int)sizeof(FastRec);
  // (a scheduling fence between the R generation bodies of a butterfly was measured every 1, 2 and 4 rows: no gain with the max-ILP
  // strategy this file is compiled with; the hook stays because ColFFT calls it on every IO)
  RF_HD static void sched_fence(int = 0) {}
  // stage the sigma records in LDS (every thread copies its share; the kernel barriers afterwards)
  // (the host only selects this kernel when nbins <= FAST_LDS_BINS, so `rec` is always an LDS pointer
  // and the lookups compile to ds_read_b128, not flat loads)
  RF_HD void prologue(int tid, int nthreads, void* lds_extra) {
    FastRec* l = reinterpret_cast<FastRec*>(lds_extra);
    for (int i = tid; i < gp.nbins && i < FAST_LDS_BINS; i += nthreads) l[i] = gp.rec[i];
    rec = l;
  }
  RF_HD void bind_seed() { if (gp.seed_dev) { gp.seed = pin_uniform(*gp.seed_dev); gp.seed_dev = nullptr; } }
  // Cell pair (kz, kz + 1) of column (ix = rb + ro, iy).  What does not depend on m (= ro / L) is a common
  // subexpression of the R unrolled loads, and what does not depend on the lane runs on the scalar ALU: the
  // Philox counter is (lane part) + (uniform part), two vector adds per load instead of a 64-bit multiply chain.
  RF_HD V16<float> load_impl(long long C0, int cl, int rb, int ro, const V16<float>* raw) const {
    V16<float> v;
    const long long C = C0 + cl;
    const uint64_t seed = gp.seed;                                                      // bind_seed() ran first
    // nzl = (nz/2) / ranks is a power of two (the launcher checks it): shift and mask instead of a 64-bit division
    const int iy = (int)((unsigned)C >> nzl_shift()), kz = kz0 + (int)((unsigned)C & (unsigned)(nzl - 1));   // lane, m-invariant
    const uint64_t half_plane = ((uint64_t)gp.ny * (uint64_t)(gp.nz / 2)) >> 1;        // counters per unit of ix
    // mode index ix = XS (rb + ro) + xp = (lane part rbt) + (uniform part rot)
    const int rbt = XS * rb, rot = XS * ro + (XS == 2 ? xp : 0);
    const uint64_t ctr_l = (uint64_t)rbt * half_plane + (((uint64_t)iy * (uint64_t)(gp.nz / 2) + (uint64_t)kz) >> 1);
    const uint64_t ctr_u = pin_uniform((uint64_t)rot * half_plane);
    // signed fftfreq index: XS rb < XS L <= nx/2 and XS ro is a multiple of XS L, so the wrap depends on ro alone
    const int ro_s = XS * ro >= (gp.nx >> 1) ? rot - gp.nx : rot;
    const float kx = (float)(rbt + ro_s) * gp.dkx, ky = (float)fast_signed_index(iy, gp.ny) * gp.dky;
    const float kxy = fmaf(kx, kx, ky * ky);                                           // == fast_kxy2(gp, rb + ro, iy)
    const float k2a = fast_k2(gp, kxy, kz), k2b = fast_k2(gp, kxy, kz + 1);
    if (SRC == 0) {
      fast_gen_pair_at(gp, rec, seed, ctr_l + ctr_u, k2a, k2b, v.c[0], v.c[1]);
    } else if (SRC == 1) {
      // cells (ix, iy, kz) and (ix, iy, kz + 1) are adjacent in the reference's order: 4 doubles, 32 contiguous bytes
      const int nzp = gp.zpitch;
      const double* d = (gp.noise + 2LL * ro * gp.ny * nzp) + 2u * (uint32_t)((rb * gp.ny + iy) * nzp + (kz - gp.zoff));
      const V16<double> ga = v16_load<double>(d), gb = v16_load<double>(d + 2);     // one complex128 = one deviate pair
      const double sa = (double)fast_sigma(gp, rec, k2a), sb = (double)fast_sigma(gp, rec, k2b);
      v.c[0] = mk<float>((float)(sa * ga.c[0].x), (float)(sa * ga.c[0].y));
      v.c[1] = mk<float>((float)(sb * gb.c[0].x), (float)(sb * gb.c[0].y));
    } else {
      // float32 pairs where the one-pass replay left them: the row's entry of the row table (rf_core.h RowLoc; index iy nx + ix =
      // a lane part that is the same for all R rows of a butterfly + the uniform row offset) says where its cells start; cells kz
      // and kz + 1 are neighbours unless a segment ends between them
      RowLoc e;
      if (raw) { e.off = 0; e.seg_n = 0; }
      else e = load_rowloc((gp.rowtab + rot) + (uint32_t)(iy * gp.nx + rbt));
      cplx<float> ga, gb;
      if (raw) { ga = raw->c[0]; gb = raw->c[1]; }                                       // (loaded by preload() at the top of the kernel)
      else { ga = load_pair_global(row_pair(gp, e, kz)); gb = load_pair_global(row_pair(gp, e, kz + 1)); }
      const float sa = fast_sigma(gp, rec, k2a), sb = fast_sigma(gp, rec, k2b);
      v.c[0] = mk<float>(sa * ga.x, sa * ga.y);
      v.c[1] = mk<float>(sb * gb.x, sb * gb.y);
    }
    if (POT == 2) {
      const float ra = fast_rcp(k2a), rb2 = fast_rcp(k2b), ps = (float)gp.pscale;      // (slot kz = 0, where k2a may be 0, is replaced by fix_value())
      v.c[0] = mk<float>((v.c[0].x * ra) * ps, (v.c[0].y * ra) * ps);
      v.c[1] = mk<float>((v.c[1].x * rb2) * ps, (v.c[1].y * rb2) * ps);
    }
    if (POT == 1) {
      // row (ix, iy) of the API layout; the slot kz = 0 (Hermitian planes) is written by fix_value() instead
      // rows of gp.ppitch (even) cells: the pair (kz even, kz + 1) is one aligned 16-byte store
      const int nzp = gp.ppitch, sl = kz - gp.zoff;
      cplx<float>* row = (pot + (long long)ro * gp.ny * nzp) + (uint32_t)((rb * gp.ny + iy) * nzp);
      const float ra = fast_rcp(k2a), rb2 = fast_rcp(k2b);
      V16<float> q;
      q.c[0] = mk<float>(v.c[0].x * ra, v.c[0].y * ra);
      q.c[1] = mk<float>(v.c[1].x * rb2, v.c[1].y * rb2);
      if (FIX != 0 && kz == 0) row[sl + 1] = q.c[1];          // slot kz = 0 itself: written by fix_value()
      else v16_store<float>(row + sl, q);
    }
    return v;
  }
  // ---- sigma shared between the rows +-ix (round 5) ------------------------------------------------------------------------------
  // |k|^2 of a cell depends on kx^2 only, and the R rows j + m L of a first-pass butterfly are the mirror images (nx - ix) of the
  // rows of butterfly L - j: thread j's rows m >= R/2 need exactly the sigmas thread L - j computes for its rows R - 1 - m < R/2.
  // ColFFT::pass_first deals the butterflies to the lanes so that the two sit in the same wave 32 lanes apart (share_row), each
  // computes the sigma of its first R/2 rows only and the halves change places through the wave's cross-lane network (ds_bpermute,
  // no LDS space, no barrier): R sigma lookups (11 vector instructions and one 16-byte LDS read each) become R/2 + R/2 exchanges.
  // The values are bit for bit those of the unshared kernel: kx enters through its square.  Butterfly 0 (rows m L, mirror R - m, row
  // R/2 L its own mirror) and butterfly L/2 (its own mirror image) take no partner: they source from themselves.  XS = 2, odd phase
  // (rows of the modes 2 r + 1): the mirror of row r is N1 - 1 - r, i.e. butterfly L - 1 - j, and no butterfly is its own partner.
  // (measured on MI355X, profiles/r05_ab/r05_c_*: the whole-column kernels gain -- x pass of 1024^3 1.186 -> 1.15 ms, 1131 instead of 1198
  // vector instructions per wave -- the two-phase Col2 form, whose register budget is full with the parked half, loses 4 %: 10.87 -> 11.29 ms
  // per 2048^3; so XS = 1 only)
  static constexpr bool SIGMA_SHARE = SRC == 0 && SLAB == 0 && XS == 1;
  // butterfly (row base) of slot jl = tid / LPR when there are S slots per wave: the first S/2 slots of a wave take q = (S/2) w + s, the
  // others its partner
  RF_HD int share_row(int jl, int L, int S) const {
    const int h = S >> 1, w = jl / S, sl = jl & (S - 1), q = h * w + (sl & (h - 1));
    if (!(sl & h)) return q;
    if (XS == 2 && xp == 1) return L - 1 - q;
    return q == 0 ? (L >> 1) : L - q;
  }
  RF_HD bool share_self(int j, int L) const { return !(XS == 2 && xp == 1) && (j == 0 || j == (L >> 1)); }
  // all R rows j + m L of the lane's cell pair; `lane` = the lane's index in its wave
  template <int R> RF_HD void load_rows(long long C0, int cl, int j, int L, int lane, V16<float>* out) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const long long C = C0 + cl;
    const uint64_t seed = gp.seed;
    const int iy = (int)((unsigned)C >> nzl_shift()), kz = kz0 + (int)((unsigned)C & (unsigned)(nzl - 1));
    const uint64_t half_plane = ((uint64_t)gp.ny * (uint64_t)(gp.nz / 2)) >> 1;
    const int rbt = XS * j;
    const uint64_t ctr_l = (uint64_t)rbt * half_plane + (((uint64_t)iy * (uint64_t)(gp.nz / 2) + (uint64_t)kz) >> 1);
    const float ky = (float)fast_signed_index(iy, gp.ny) * gp.dky, ky2 = ky * ky;
    float sa[R], sb[R];
#pragma unroll
    for (int m = 0; m < R / 2; ++m) {                      // rows below nx / 2: the mode index is the row's own
      const float kx = (float)(rbt + XS * m * L + (XS == 2 ? xp : 0)) * gp.dkx;
      const float kxy = fmaf(kx, kx, ky2);
      sa[m] = fast_sigma(gp, rec, fast_k2(gp, kxy, kz));
      sb[m] = fast_sigma(gp, rec, fast_k2(gp, kxy, kz + 1));
    }
    const int src = (share_self(j, L) ? lane : lane ^ 32) << 2;
#pragma unroll
    for (int m = 0; m < R / 2; ++m) {
      sa[R - 1 - m] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sa[m])));
      sb[R - 1 - m] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(sb[m])));
    }
    if (j == 0 && !(XS == 2 && xp == 1)) {                 // rows m L: the mirror of row m is row R - m, row R/2 (mode nx / 2) is its own
      const float kx = (float)(XS * (R / 2) * L) * gp.dkx;
      const float kxy = fmaf(kx, kx, ky2);
      sa[R / 2] = fast_sigma(gp, rec, fast_k2(gp, kxy, kz));
      sb[R / 2] = fast_sigma(gp, rec, fast_k2(gp, kxy, kz + 1));
#pragma unroll
      for (int m = R / 2 + 1; m < R; ++m) { sa[m] = sa[R - m]; sb[m] = sb[R - m]; }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int ro = m * L, rot = XS * ro + (XS == 2 ? xp : 0);
      const uint64_t ctr_u = pin_uniform((uint64_t)rot * half_plane);
      const PhiloxOut o = philox_native(ctr_l + ctr_u, 0, seed);
      float g0, g1;
      BoxMuller<float>::run_scaled(o.w[0], o.w[1], sa[m], g0, g1);
      out[m].c[0] = mk<float>(g0, g1);
      BoxMuller<float>::run_scaled(o.w[2], o.w[3], sb[m], g0, g1);
      out[m].c[1] = mk<float>(g0, g1);
      if (POT != 0) {
        const int ro_s = XS * ro >= (gp.nx >> 1) ? rot - gp.nx : rot;
        const float kx = (float)(rbt + ro_s) * gp.dkx, kxy = fmaf(kx, kx, ky2);
        const float ra = fast_rcp(fast_k2(gp, kxy, kz)), rb2 = fast_rcp(fast_k2(gp, kxy, kz + 1));
        if (POT == 2) {
          const float ps = (float)gp.pscale;
          out[m].c[0] = mk<float>((out[m].c[0].x * ra) * ps, (out[m].c[0].y * ra) * ps);
          out[m].c[1] = mk<float>((out[m].c[1].x * rb2) * ps, (out[m].c[1].y * rb2) * ps);
        } else {
          const int nzp = gp.ppitch, sl = kz - gp.zoff;
          cplx<float>* row = (pot + (long long)ro * gp.ny * nzp) + (uint32_t)((j * gp.ny + iy) * nzp);
          V16<float> q;
          q.c[0] = mk<float>(out[m].c[0].x * ra, out[m].c[0].y * ra);
          q.c[1] = mk<float>(out[m].c[1].x * rb2, out[m].c[1].y * rb2);
          if (FIX != 0 && kz == 0) row[sl + 1] = q.c[1];
          else v16_store<float>(row + sl, q);
        }
      }
    }
#else
    for (int m = 0; m < R; ++m) out[m] = load(C0, cl, j, m * L);      // (the emulator has no lanes: the same values row by row)
#endif
  }
  // SRC = 2: the memory half of load() -- the row's table entry, then its two deviate pairs -- for the kernel to issue before it
  // stages any table (col_kernel): three dependent round trips (records, row table, pairs) become two that overlap the staging
  static constexpr bool HAS_PRELOAD = (SRC == 2);
  RF_HD V16<float> preload(long long C0, int cl, int rb, int ro) const {
    V16<float> v;
    const long long C = C0 + cl;
    const int iy = (int)((unsigned)C >> nzl_shift()), kz = kz0 + (int)((unsigned)C & (unsigned)(nzl - 1));
    const RowLoc e = load_rowloc((gp.rowtab + (XS * ro + (XS == 2 ? xp : 0))) + (uint32_t)(iy * gp.nx + XS * rb));
    v.c[0] = load_pair_global(row_pair(gp, e, kz));
    v.c[1] = load_pair_global(row_pair(gp, e, kz + 1));
    return v;
  }
  RF_HD V16<float> load_pre(long long C0, int cl, int rb, int ro, const V16<float>& raw) const { return load_impl(C0, cl, rb, ro, &raw); }
  RF_HD V16<float> load(long long C0, int cl, int rb, int ro) const { return load_impl(C0, cl, rb, ro, nullptr); }
  // the lane that owns slot kz = 0 replaces its provisional first cell of every row by the packed,
  // symmetrised (kz=0, kz=nz/2) pair (cold path: one lane in four of one tile in nz/16)
  static constexpr bool ROLLED_LOAD = false;
  RF_HD long long remap_tile(long long t) const { return t; }
  static constexpr bool HAS_FINISH = false;
  static constexpr int FIX_MODE = FIX;
  template <int F2> using with_fix = FastGenColIOT<F2, SLAB, POT, SRC, XS>;
  using fill_io = FastGenColIOT<1, SLAB, POT, SRC, 1>;      // the IO whose fix_value() fix_fill_kernel evaluates (mode index = row)
  const cplx<float>* fixbuf = nullptr;                          // FIX = 3: [ny][nx] repaired slots kz = 0, left by fix_fill_kernel
  RF_HD bool needs_fix(long long C) const { return FIX != 0 && kz0 + (int)((unsigned)C & (unsigned)(nzl - 1)) == 0; }
  // FIX = 3: the repaired slot of mode (XS (rb + ro) + xp, iy) from the side buffer (lane part + uniform part, as load())
  RF_HD cplx<float> fix_load(long long C, int rb, int ro) const {
    const int iy = (int)((unsigned)C >> nzl_shift());
    return load_pair_global((fixbuf + (XS * ro + (XS == 2 ? xp : 0))) + (uint32_t)(iy * gp.nx + XS * rb));
  }
  RF_HD cplx<float> fix_value(long long C, int rb, int ro) const {
    if (FIX == 3) return fix_load(C, rb, ro);
    const uint64_t seed = gp.seed;                                                      // bind_seed() ran first
    const int iy = (int)((unsigned)C >> nzl_shift());
    cplx<float> p0, pn;
    const int ixm = XS * (rb + ro) + (XS == 2 ? xp : 0);                                  // the row's mode index
    const cplx<float> packed = SRC != 0 ? fast_fix_kz0_noise<SRC == 0 ? 1 : SRC>(gp, rec, ixm, iy, p0, pn)
                                        : fast_fix_kz0(gp, rec, seed, ixm, iy, p0, pn);
    if (POT == 1) {
      cplx<float>* row = pot + ((long long)(rb + ro) * gp.ny + iy) * gp.ppitch;   // only the rank with kz0 = 0 gets here: slot 0 = plane 0
      row[0] = p0;
      row[gp.zpitch - 1] = pn;
    }
    if (POT == 2) {           // (plane kz = 0) + i (plane kz = nz/2) of the scaled potential
      const float ps = (float)gp.pscale;
      return mk<float>(p0.x * ps - pn.y * ps, p0.y * ps + pn.x * ps);
    }
    return packed;
  }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<float>& v) const {
    // x0, x1 are multiples of the last pass's row stride L (the launcher checks it) and rb < L: the test is uniform
    if (SLAB && (ro < x0 || ro >= x1)) return;
    v16_store<float>(g.at<false>(base, C0, cl, rb, ro), v);
  }
};
using FastGenColIO = FastGenColIOT<1>;   // (emulator)

// The same fast generation for float64 plans (native generator only: parity / reference-noise mode keeps the exact
// float64 chain of GenColIO).  The deviates and sigma are formed in float32 -- hardware log / sin / cos, LDS
// records -- and widened; the transform itself is float64.  One complex128 per lane (CPL = 1).
template <int FIX = 1, int SLAB = 0, int POT = 0, int XS = 1>
struct FastGenColIO64 {
  static_assert(XS == 1 || (XS == 2 && SLAB == 0 && POT != 1), "half-transform rows: no x-slab restriction, no potential store");
  int xp = 0;
  RF_HD void set_phase(int p) { xp = p; }
  cplx<double>* base;
  ColGeom g;
  FastGenParams gp;
  cplx<double>* pot = nullptr;   // POT = 1: where delta(k) / k^2 goes (API-layout rows of gp.zpitch cells, 16-byte aligned)
  int kz0, nzl;
  int x0 = 0, x1 = 1 << 30;  // replicated-generation mode: see FastGenColIOT
  RF_HD int nzl_shift() const { return 31 - __builtin_clz((unsigned)nzl); }
  const FastRec* rec;
  static constexpr int LDS_EXTRA = FAST_LDS_BINS * (int)sizeof(FastRec);
  RF_HD static void sched_fence(int = 0) {}
  RF_HD void prologue(int tid, int nthreads, void* lds_extra) {
    FastRec* l = reinterpret_cast<FastRec*>(lds_extra);
    for (int i = tid; i < gp.nbins && i < FAST_LDS_BINS; i += nthreads) l[i] = gp.rec[i];
    rec = l;
  }
  RF_HD void bind_seed() { if (gp.seed_dev) { gp.seed = pin_uniform(*gp.seed_dev); gp.seed_dev = nullptr; } }
  // the cell of row (rb, ro) in column C: its native noise index (lane part + uniform part) and |k|^2
  RF_HD void cell_of(long long C, int rb, int ro, int& iy, int& kz, uint64_t& ci_l, uint64_t& ci_u, float& k2) const {
    iy = (int)((unsigned)C >> nzl_shift());
    kz = kz0 + (int)((unsigned)C & (unsigned)(nzl - 1));                                // lane, m-invariant
    const uint64_t plane = (uint64_t)gp.ny * (uint64_t)(gp.nz / 2);                    // noise cells per unit of ix
    // mode index ix = XS (rb + ro) + xp = (lane part rbt) + (uniform part rot), as in FastGenColIOT
    const int rbt = XS * rb, rot = XS * ro + (XS == 2 ? xp : 0);
    ci_l = (uint64_t)rbt * plane + (uint64_t)iy * (uint64_t)(gp.nz / 2) + (uint64_t)kz;
    ci_u = pin_uniform((uint64_t)rot * plane);
    const int ro_s = XS * ro >= (gp.nx >> 1) ? rot - gp.nx : rot;
    const float kx = (float)(rbt + ro_s) * gp.dkx, ky = (float)fast_signed_index(iy, gp.ny) * gp.dky;
    k2 = fast_k2(gp, fmaf(kx, kx, ky * ky), kz);
  }
  // the cell's value from its two Philox words (Box-Muller, sigma) + the potential variants
  RF_HD V16<double> cell_from_words(uint32_t wa, uint32_t wb, float k2, int iy, int kz, int rb, int ro) const {
    float g0, g1;
    BoxMuller<float>::run_scaled(wa, wb, fast_sigma(gp, rec, k2), g0, g1);
    const cplx<float> c = mk<float>(g0, g1);
    V16<double> v;
    v.c[0] = mk<double>((double)c.x, (double)c.y);
    if (POT == 2) {
      const float r = fast_rcp(k2);
      v.c[0] = mk<double>((double)(c.x * r) * gp.pscale, (double)(c.y * r) * gp.pscale);
    }
    if (POT == 1 && !(FIX != 0 && kz == 0)) {          // (slot kz = 0: the two Hermitian planes, written by fix_value())
      const float r = fast_rcp(k2);
      V16<double> q;
      q.c[0] = mk<double>((double)(c.x * r), (double)(c.y * r));
      v16_store<double>((pot + (long long)ro * gp.ny * gp.ppitch) + (uint32_t)((rb * gp.ny + iy) * gp.ppitch + (kz - gp.zoff)), q);
    }
    return v;
  }
  RF_HD V16<double> load(long long C0, int cl, int rb, int ro) const {
    int iy, kz;
    uint64_t ci_l, ci_u;
    float k2;
    cell_of(C0 + cl, rb, ro, iy, kz, ci_l, ci_u, k2);
    const uint64_t ci = ci_l + ci_u;
    const PhiloxOut o = philox_native(ci >> 1, 0, gp.seed);                             // bind_seed() ran first
    const bool odd = (ci & 1u) != 0;
    return cell_from_words(odd ? o.w[2] : o.w[0], odd ? o.w[3] : o.w[1], k2, iy, kz, rb, ro);
  }
  // Two rows at once.  A Philox call serves the cell pair (kz even, kz + 1) of one row, and with one complex128 per lane that pair sits
  // in the lane pair (2l, 2l + 1): load() has both lanes run the same call and keep half of it.  Here the even lane runs row A's call
  // and the odd lane row B's; each sends the half its neighbour needs across (one quad-permute DPP move per word) -- one call per lane
  // and two rows instead of two (230 -> 109 v_mad_u64_u32 per thread in the 1024-point kernel).  Same words, same field.
  static constexpr bool HAS_LOAD_PAIR = true;
  RF_HD void load_pair(long long C0, int cl, int rb, int roA, int roB, V16<double>& a, V16<double>& b) const {
#if defined(__HIP_DEVICE_COMPILE__)
    int iy, kz, iy2, kz2;
    uint64_t ci_l, cuA, cuB, ci_l2;
    float k2A, k2B;
    cell_of(C0 + cl, rb, roA, iy, kz, ci_l, cuA, k2A);
    cell_of(C0 + cl, rb, roB, iy2, kz2, ci_l2, cuB, k2B);
    const bool odd = (kz & 1) != 0;                                                      // (the rest of the noise index is even: nz / 2 is)
    const PhiloxOut o = philox_native((ci_l + (odd ? cuB : cuA)) >> 1, 0, gp.seed);
    const uint32_t ra = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd ? o.w[0] : o.w[2]), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    const uint32_t rb2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd ? o.w[1] : o.w[3]), 0xB1, 0xF, 0xF, true);
    a = cell_from_words(odd ? ra : o.w[0], odd ? rb2 : o.w[1], k2A, iy, kz, rb, roA);
    b = cell_from_words(odd ? o.w[2] : ra, odd ? o.w[3] : rb2, k2B, iy, kz, rb, roB);
#else
    a = load(C0, cl, rb, roA);
    b = load(C0, cl, rb, roB);
#endif
  }
  static constexpr bool ROLLED_LOAD = false;
  RF_HD long long remap_tile(long long t) const { return t; }
  static constexpr bool HAS_FINISH = false;
  static constexpr int FIX_MODE = FIX;
  template <int F2> using with_fix = FastGenColIO64<F2, SLAB, POT, XS>;
  using fill_io = FastGenColIO64<1, SLAB, POT, 1>;
  const cplx<double>* fixbuf = nullptr;                         // FIX = 3: see FastGenColIOT
  RF_HD bool needs_fix(long long C) const { return FIX != 0 && kz0 + (int)((unsigned)C & (unsigned)(nzl - 1)) == 0; }
  RF_HD cplx<double> fix_load(long long C, int rb, int ro) const {
    const int iy = (int)((unsigned)C >> nzl_shift());
    return v16_load<double>((fixbuf + (XS * ro + (XS == 2 ? xp : 0))) + (uint32_t)(iy * gp.nx + XS * rb)).c[0];
  }
  RF_HD cplx<double> fix_value(long long C, int rb, int ro) const {
    if (FIX == 3) return fix_load(C, rb, ro);
    const uint64_t seed = gp.seed;                                                      // bind_seed() ran first
    cplx<float> p0, pn;
    const int iy = (int)((unsigned)C >> nzl_shift());
    const cplx<float> c = fast_fix_kz0(gp, rec, seed, XS * (rb + ro) + (XS == 2 ? xp : 0), iy, p0, pn);
    if (POT == 1) {
      cplx<double>* row = pot + ((long long)(rb + ro) * gp.ny + iy) * gp.ppitch;      // only the rank with kz0 = 0 gets here
      row[0] = mk<double>((double)p0.x, (double)p0.y);
      row[gp.zpitch - 1] = mk<double>((double)pn.x, (double)pn.y);
    }
    if (POT == 2) {
      const double ps = gp.pscale;
      return mk<double>((double)p0.x * ps - (double)pn.y * ps, (double)p0.y * ps + (double)pn.x * ps);
    }
    return mk<double>((double)c.x, (double)c.y);
  }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<double>& v) const {
    if (SLAB && (ro < x0 || ro >= x1)) return;          // uniform: see FastGenColIOT::store
    v16_store<double>(g.at<false>(base, C0, cl, rb, ro), v);
  }
};

// Does the IO split its load into an early memory part and a late arithmetic part (preload() / load_pre())?  Only the deviate-reading
// generation pass does: its loads are issued at the very top of the kernel, in front of the table staging and its barrier.
template <class IO, class = void> struct io_sigma_share { static constexpr bool value = false; };
template <class IO> struct io_sigma_share<IO, typename std::enable_if<IO::SIGMA_SHARE>::type> { static constexpr bool value = true; };
template <class IO, class = void> struct io_has_load_pair { static constexpr bool value = false; };
template <class IO> struct io_has_load_pair<IO, typename std::enable_if<IO::HAS_LOAD_PAIR>::type> { static constexpr bool value = true; };
template <class IO, class = void> struct io_has_preload { static constexpr bool value = false; };
template <class IO> struct io_has_preload<IO, typename std::enable_if<IO::HAS_PRELOAD>::type> { static constexpr bool value = true; };

// ---------------------------------------------------------------------------
// Column FFT phases.  `tw` = exp(+2 pi i q / N), q in [0, N).
// ---------------------------------------------------------------------------
template <class C, int DIR, class IO>
struct ColFFT {
  using T = typename C::T;
  using cx = cplx<T>;
  using V = V16<T>;
  static constexpr int N = C::N, CPL = C::CPL, LPR = C::LPR, BPI = C::BPI;

  struct Regs { cx v[C::IT2][cmax(C::R2, 1)]; };

  RF_HD static V* lds_at(cx* lds, int row, int lp) {
    return reinterpret_cast<V*>(lds + (long long)C::prow(row) * C::TC) + lp;
  }
  RF_HD static cx* lds_col(cx* lds, int row, int t) { return lds + (long long)C::prow(row) * C::TC + t; }
  // The swizzle row ^ ((row / R1) & 1) flips bit 0 of the row by a bit that, in every pass, depends on the THREAD (or on the unrolled
  // index m alone) but not on both: the R accesses of a butterfly are then (one of two per-thread bases) + (a compile-time multiple
  // of the row pitch), i.e. ONE or TWO address registers and immediate offsets on the ds_ instructions.  Written out, because the
  // compiler does not distribute the XOR over the sum: it spent ~100 of the generation kernel's 1300 vector instructions (lshl_add,
  // xad, or) on one full address per access.  Conditions (all shipped 3-pass configurations meet them; the generic form otherwise):
  //   rows j + m L (middle-pass reads, last-pass reads): L a multiple of 2 R1  ->  bit = (j / R1) & 1
  //   rows j R1 + m (first-pass writes):                                          bit = j & 1, row = j R1 + (m ^ bit)
  //   rows ob + m R1, ob = (j / R1) R1 R + j % R1 (middle-pass writes), R even:   bit = m & 1, row = (ob ^ bit) + m R1
  static constexpr bool FOLD_FIRST = C::NPASS >= 2;
  static constexpr bool FOLD_MIDR = C::NPASS == 3 && (N / cmax(C::R2, 1)) % (2 * C::R1) == 0;
  static constexpr bool FOLD_MIDW = C::NPASS == 3 && C::R2 % 2 == 0;
  static constexpr bool FOLD_LAST = C::NPASS >= 2 && (N / C::RL) % (2 * C::R1) == 0;

  // LDS carve: [tile][twiddles][IO tables]
  RF_HD static cx* lds_tw(cx* lds) { return lds + C::TILE_BYTES / (int)sizeof(cx); }
  RF_HD static void* lds_io(cx* lds) { return lds + (C::TILE_BYTES + C::TW_BYTES) / (int)sizeof(cx); }
  static constexpr bool HAS_PROLOGUE = (C::NPASS >= 2) || (IO::LDS_EXTRA > 0);

  // The twiddle table goes global -> registers at the very start of the kernel (tw_fetch: loads issued, not waited for) and
  // registers -> LDS after pass 1 (tw_stage), in front of the barrier that precedes its first use: its trip to L2 / HBM runs
  // under pass 1 instead of in front of it (two dependent round trips per workgroup before the first useful instruction).
  static constexpr int TWPT = (C::NPASS >= 2 ? ceil_div(N, C::NT) : 0);      // table entries per thread
  struct TwRegs { cx v[cmax(TWPT, 1)]; };
  RF_HD static void tw_fetch(int tid, const cx* tw, TwRegs& t) {
#pragma unroll
    for (int k = 0; k < TWPT; ++k) t.v[k] = tw[(tid + k * C::NT) & (N - 1)];       // (N is a power of two: no branch, no undefined slot)
  }
  RF_HD static void tw_stage(int tid, cx* lds, const TwRegs& t) {
    cx* l = lds_tw(lds);
#pragma unroll
    for (int k = 0; k < TWPT; ++k)
      if (tid + k * C::NT < N) l[tid + k * C::NT] = t.v[k];
  }
  // prologue (emulator; the kernels call the pieces): stage the twiddle table and the IO's own tables in LDS; a barrier follows
  RF_HD static void prologue(int tid, IO& io, const cx* tw, cx* lds) {
    if (C::NPASS >= 2) {
      TwRegs t;
      tw_fetch(tid, tw, t);
      tw_stage(tid, lds, t);
    }
    io.prologue(tid, C::NT, lds_io(lds));
  }

  // the early memory half of pass 1 for IOs that split their load (io_has_preload): what preload() returns, kept in registers
  static constexpr bool PRELOAD = io_has_preload<IO>::value;
  struct PreRegs { V v[PRELOAD ? C::IT1 : 1][PRELOAD ? C::R1 : 1]; };
  RF_HD static void preload(int tid, long long tile, const IO& io, PreRegs& pre) {
    if constexpr (PRELOAD) {
      constexpr int R = C::R1, L = N / R;
      const int lp = tid % LPR, jl = tid / LPR;
      const long long C0 = tile * C::TC;
#pragma unroll
      for (int it = 0; it < C::IT1; ++it) {
        const int j = it * BPI + jl;
        if (j < L) {
#pragma unroll
          for (int m = 0; m < R; ++m) pre.v[it][m] = io.preload(C0, lp * CPL, j, m * L);
        }
      }
    }
  }
  RF_HD static void pass_first(int tid, long long tile, const IO& io, cx* lds) {
    PreRegs none;
    pass_first(tid, tile, io, lds, none, false);
  }
  // pass 1: global -> R1 butterfly -> LDS (or straight back to global when N == R1)
  // (FIXOK = false: the caller knows that this tile holds no kz = 0 slot -- the second tile of a ColPair -- and the repair code is left out)
  template <bool FIXOK = true>
  RF_HD static void pass_first(int tid, long long tile, const IO& io, cx* lds, const PreRegs& pre, bool have_pre) {
    constexpr int R = C::R1, L = N / R;
    const int lp = tid % LPR, jl = tid / LPR;
    const long long C0 = tile * C::TC;                       // workgroup-uniform
    const int cl = lp * CPL;
    const long long Ccol = C0 + cl;
    // sigma shared between the rows +-ix (IO::load_rows): one iteration covers all L butterflies, whole waves, an even radix
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool SHARE = io_sigma_share<IO>::value && C::IT1 == 1 && BPI == L && R % 2 == 0 && 64 % LPR == 0 && 64 / LPR >= 2 &&
                           C::NT % 64 == 0 && CPL == 2 && !PRELOAD;
#else
    constexpr bool SHARE = false;
#endif
#pragma unroll
    for (int it = 0; it < C::IT1; ++it) {
      int j = it * BPI + jl;
      if constexpr (SHARE) j = io.share_row(jl, L, 64 / LPR);
      if (j < L) {
        cx v[CPL][R];
        constexpr bool PRE = (IO::FIX_MODE == 3) && FIXOK;
        if constexpr (SHARE) {
          V rows[R];
          io.template load_rows<R>(C0, cl, j, L, tid & 63, rows);
#pragma unroll
          for (int m = 0; m < R; ++m) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) v[c][m] = rows[m].c[c];
          }
        } else if (IO::ROLLED_LOAD && C::NPASS > 1) {
#pragma unroll 1
          for (int m = 0; m < R; ++m) *lds_at(lds, j * R + m, lp) = io.load(C0, cl, j, m * L);
#pragma unroll
          for (int m = 0; m < R; ++m) {
            V x = *lds_at(lds, j * R + m, lp);
#pragma unroll
            for (int c = 0; c < CPL; ++c) v[c][m] = x.c[c];
          }
        } else {
          if constexpr (io_has_load_pair<IO>::value && R % 2 == 0 && !PRELOAD) {
#pragma unroll
            for (int m = 0; m < R; m += 2) {                 // (IOs that generate two rows for the price of one: FastGenColIO64)
              V xa, xb;
              io.load_pair(C0, cl, j, m * L, (m + 1) * L, xa, xb);
#pragma unroll
              for (int c = 0; c < CPL; ++c) { v[c][m] = xa.c[c]; v[c][m + 1] = xb.c[c]; }
            }
          } else {
#pragma unroll
          for (int m = 0; m < R; ++m) {
            V x;
            if constexpr (PRELOAD) x = have_pre ? io.load_pre(C0, cl, j, m * L, pre.v[it][m]) : io.load(C0, cl, j, m * L);
            else x = io.load(C0, cl, j, m * L);
#pragma unroll
            for (int c = 0; c < CPL; ++c) v[c][m] = x.c[c];
            IO::sched_fence(m);
          }
          }
        }
        if (PRE) {
          // FIX_MODE == 3: the owning lane (one in LPR, of one tile in nz / 16) replaces its first cell of every row by the repaired
          // slot from the side buffer -- straight into the butterfly's registers, behind the generation: holding the eight values
          // across it costs sixteen more registers than the kernels have (28 - 256 bytes of scratch per thread when tried)
          if (io.needs_fix(Ccol)) {
#pragma unroll
            for (int m = 0; m < R; ++m) v[0][m] = io.fix_value(Ccol, j, m * L);
          }
        } else if (FIXOK && IO::FIX_MODE != 0 && io.needs_fix(Ccol)) {
          if (C::NPASS == 1) {
#pragma unroll
            for (int m = 0; m < R; ++m) v[0][m] = io.fix_value(Ccol, j, m * L);
          } else {
            // rolled loop (one copy of the body); values are parked in this thread's own, still unused
            // LDS output slots and read back with static register indices
#pragma unroll 1
            for (int m = 0; m < R; ++m) lds_at(lds, j * R + m, lp)->c[0] = io.fix_value(Ccol, j, m * L);
#pragma unroll
            for (int m = 0; m < R; ++m) v[0][m] = lds_at(lds, j * R + m, lp)->c[0];
          }
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) DFT<R, DIR>::run(v[c]);
        // (rows j R + m of the swizzled image: row j R + (m ^ (j & 1)) -- two bases, immediate offsets)
        const int sw1 = j & 1;
        V* const w_even = reinterpret_cast<V*>(lds + (long long)(j * R + sw1) * C::TC) + lp;
        V* const w_odd = reinterpret_cast<V*>(lds + (long long)(j * R - sw1) * C::TC) + lp;
#pragma unroll
        for (int m = 0; m < R; ++m) {
          V x;
#pragma unroll
          for (int c = 0; c < CPL; ++c) x.c[c] = v[c][m];
          if (C::NPASS == 1) io.store(C0, cl, j * R, m, x);
          else if (FOLD_FIRST) ((m & 1) ? w_odd : w_even)[m * LPR] = x;
          else *lds_at(lds, j * R + m, lp) = x;
        }
      }
    }
  }

  // middle pass (only when NPASS == 3): in place, so split around a barrier.  One column per
  // lane (8-byte LDS accesses): twice the threads of the 16-byte passes stay busy.
  RF_HD static void pass_mid_read(int tid, const cx* tw, cx* lds, Regs& r) {
    constexpr int R = C::R2, L = N / R, Ns = C::R1;
    const int t = tid % C::TC, jl = tid / C::TC;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int j = it * C::BPM + jl;
      if (j < L) {
        const cx* const rd = lds + (long long)(j ^ ((j / C::R1) & 1)) * C::TC + t;      // FOLD_MIDR: row (j ^ bit) + m L
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = FOLD_MIDR ? rd[(long long)m * L * C::TC] : *lds_col(lds, j + m * L, t);
          if (m > 0) x = cmul(x, tw_dir<DIR>(tw[stockham_tw_index(j, m, Ns, R, N)]));
          r.v[it][m] = x;
        }
        DFT<R, DIR>::run(r.v[it]);
      }
    }
  }
  RF_HD static void pass_mid_write(int tid, cx* lds, const Regs& r) {
    constexpr int R = C::R2, L = N / R, Ns = C::R1;
    const int t = tid % C::TC, jl = tid / C::TC;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int j = it * C::BPM + jl;
      if (j < L) {
        const int ob = stockham_out_base(j, Ns, R);
        cx* const w0 = lds + (long long)ob * C::TC + t;                               // FOLD_MIDW: row (ob ^ (m & 1)) + m Ns
        cx* const w1 = lds + (long long)(ob ^ 1) * C::TC + t;
#pragma unroll
        for (int m = 0; m < R; ++m) {
          if (FOLD_MIDW) ((m & 1) ? w1 : w0)[(long long)m * Ns * C::TC] = r.v[it][m];
          else *lds_col(lds, ob + m * Ns, t) = r.v[it][m];
        }
      }
    }
  }

  // last pass (NPASS >= 2): LDS -> RL butterfly -> global.  The butterfly of iteration `it` leaves one 16-byte
  // vector per output row m (row j + m L of this lane's CPL columns) in out[m].
  RF_HD static void last_butterfly(int j, int lp, const cx* tw, cx* lds, V* out) {
    constexpr int R = C::RL, L = N / R;  // Ns == L, out_base(j) == j, twiddle index == m*j
    cx v[CPL][R];
    const V* const rd = reinterpret_cast<const V*>(lds + (long long)(j ^ ((j / C::R1) & 1)) * C::TC) + lp;     // FOLD_LAST: row (j ^ bit) + m L
#pragma unroll
    for (int m = 0; m < R; ++m) {
      V x = FOLD_LAST ? rd[m * L * LPR] : *lds_at(lds, j + m * L, lp);
      if (m > 0) {
        const cx w = tw_dir<DIR>(tw[m * j]);
#pragma unroll
        for (int c = 0; c < CPL; ++c) x.c[c] = cmul(x.c[c], w);
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) v[c][m] = x.c[c];
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) DFT<R, DIR>::run(v[c]);
#pragma unroll
    for (int m = 0; m < R; ++m) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) out[m].c[c] = v[c][m];
    }
  }
  RF_HD static void pass_last(int tid, long long tile, const IO& io, const cx* tw, cx* lds) {
    constexpr int R = C::RL, L = N / R;
    const int lp = tid % LPR, jl = tid / LPR;
    const long long C0 = tile * C::TC;                       // workgroup-uniform
    const int cl = lp * CPL;
    const long long Ccol = C0 + cl;
#pragma unroll
    for (int it = 0; it < C::ITL; ++it) {
      const int j = it * BPI + jl;
      if (j < L) {
        V out[R];
        last_butterfly(j, lp, tw, lds, out);
#pragma unroll
        for (int m = 0; m < R; ++m) io.store(C0, cl, j, m * L, out[m]);
      }
    }
  }
};

// ---------------------------------------------------------------------------
// Col2: a strided transform of length 2 N1 as TWO transforms of length N1 per tile, one after the other, + one radix-2 step in
// registers (decimation in time):  out[x] = E[x mod N1] + w^x O[x mod N1],  E / O = the length-N1 transforms of the even / odd
// input rows, w = exp(DIR 2 pi i / 2 N1).  Phase 0 transforms the even rows and PARKS the last pass's outputs (R V16 per thread:
// 32 registers for float32) instead of storing them; phase 1 transforms the odd rows, and its last pass combines and stores rows
// x and x + N1.  The tile in LDS is the N1-point one (64 KB at N1 = 1024): two 512-thread workgroups share a CU, where the
// whole-column 2048-point tile (152 KB) allows one workgroup whose sixteen waves load / generate, transform and store in lock
// step.  The combine's twiddle w^(j + m L) = w^j * exp(DIR 2 pi i m / 2 R): one table entry per thread,
// the rest are the constant 16th roots of unity (R = 8).
// ---------------------------------------------------------------------------
// in-place pass over an array whose rows 2 r + phase feed phase `phase`: load geometry gin (row stride doubled), store geometry g
template <typename T> struct Pair2ColIO {
  cplx<T>* base;
  ColGeom gin, g;
  long long par_off;             // elements between row 2r and row 2r + 1 (the plain row stride)
  int phase = 0;
  RF_HD void set_phase(int p) { phase = p; }
  RF_HD V16<T> load(long long C0, int cl, int rb, int ro) const { return v16_load<T>(gin.at<false>(base + (long long)phase * par_off, C0, cl, rb, ro)); }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<T>& v) const { v16_store<T>(g.at<false>(base, C0, cl, rb, ro), v); }
  static constexpr int FIX_MODE = 0;
  RF_HD bool needs_fix(long long) const { return false; }
  RF_HD cplx<T> fix_value(long long, int, int) const { return cplx<T>(); }
  static constexpr int LDS_EXTRA = 0;
  RF_HD void prologue(int, int, void*) {}
  RF_HD void bind_seed() {}
  RF_HD static void sched_fence(int = 0) {}
  static constexpr bool ROLLED_LOAD = false;
  RF_HD long long remap_tile(long long t) const { return t; }
  static constexpr bool HAS_FINISH = false;
};

// ---------------------------------------------------------------------------
// The y pass of a kz-slab rank that IS the exchange ("direct" mode): it reads the rank's array [nx][ny][nzl] and
// stores every output tile straight into the receive buffer of the rank that owns the tile's x plane -- no send buffer, no copy
// kernels, no local traffic beyond what the pass moves anyway.  A tile (all ny rows of TC kz columns of ONE ix) has exactly one
// destination h = ix / nxl, and the destination's layout [source][nxl][ny][nzl] is the local one shifted by a per-destination base:
//   cell (ix, iy, kz) of rank g   local:  ((ix * ny) + iy) * nzl + kz  =  h * blk + off
//                                 remote: R_h + g * blk + off                               (blk = nxl * ny * nzl cells)
// so tab[h] = R_h + (g - h) * blk and the store geometry is the load geometry.  `tab` lives in device memory (one scalar load per
// tile); R_h is a peer-mapped pointer (hipIpcOpenMemHandle) on a real job and a plain device pointer between virtual ranks.
// bind_tile() is called once per workgroup, before the passes.
// ---------------------------------------------------------------------------
template <typename T, bool WIDE = false> struct DirectColIO : PlainColIO<T, WIDE> {
  cplx<T>* out = nullptr;
  cplx<T>* const* tab = nullptr;
  int dest_shift = 0;            // log2(x planes per destination rank)
  RF_HD void bind_tile(long long C0) { out = tab[(C0 >> this->g.inner_shift()) >> dest_shift]; }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<T>& v) const { v16_store<T>(this->g.template at<WIDE>(out, C0, cl, rb, ro), v); }
};
// ... and the same for the passes that run as two half-length transforms per tile (Col2)
template <typename T> struct Pair2DirectColIO : Pair2ColIO<T> {
  cplx<T>* out = nullptr;
  cplx<T>* const* tab = nullptr;
  int dest_shift = 0;
  RF_HD void bind_tile(long long C0) { out = tab[(C0 >> this->g.inner_shift()) >> dest_shift]; }
  RF_HD void store(long long C0, int cl, int rb, int ro, const V16<T>& v) const { v16_store<T>(this->g.template at<false>(out, C0, cl, rb, ro), v); }
};

// exp(DIR * 2 pi i m / 16), m in [0, 8)
template <int DIR, typename T> RF_HD cplx<T> w16_half(int m) {
  const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173, r = (T)0.70710678118654752440;
  T cs, sn;
  switch (m) {
    case 0: cs = 1; sn = 0; break;
    case 1: cs = c1; sn = s1; break;
    case 2: cs = r; sn = r; break;
    case 3: cs = s1; sn = c1; break;
    case 4: cs = 0; sn = 1; break;
    case 5: cs = -s1; sn = c1; break;
    case 6: cs = -r; sn = r; break;
    default: cs = -c1; sn = s1; break;
  }
  return mk<T>(cs, DIR > 0 ? sn : -sn);
}

template <class C1, int DIR, class IO>
struct Col2 {
  using F = ColFFT<C1, DIR, IO>;
  using T = typename C1::T;
  using cx = cplx<T>;
  using V = V16<T>;
  static constexpr int N1 = C1::N, R = C1::RL, L = N1 / R, CPL = C1::CPL, LPR = C1::LPR, BPI = C1::BPI;
  static_assert(C1::NPASS >= 2 && R == 8, "Col2 combines behind a radix-8 last pass through LDS");
  struct Park { V out[C1::ITL][R]; };
  // tw2 = exp(+2 pi i q / 2 N1), q in [0, 2 N1): the N1-point table is every second entry
  RF_HD static void tw_fetch(int tid, const cx* tw2, typename F::TwRegs& t) {
#pragma unroll
    for (int k = 0; k < F::TWPT; ++k) t.v[k] = tw2[2 * ((tid + k * C1::NT) & (N1 - 1))];
  }
  // phase 0, last pass: LDS -> butterfly -> registers
  RF_HD static void last_park(int tid, const cx* tw, cx* lds, Park& pk) {
    const int lp = tid % LPR, jl = tid / LPR;
#pragma unroll
    for (int it = 0; it < C1::ITL; ++it) {
      const int j = it * BPI + jl;
      if (j < L) F::last_butterfly(j, lp, tw, lds, pk.out[it]);
    }
  }
  // phase 1, last pass: LDS -> butterfly -> radix-2 step with the parked half -> rows x and x + N1
  RF_HD static void last_combine(int tid, long long tile, const IO& io, const cx* tw, const cx* tw2, cx* lds, const Park& pk) {
    const int lp = tid % LPR, jl = tid / LPR;
    const long long C0 = tile * C1::TC;
    const int cl = lp * CPL;
#pragma unroll
    for (int it = 0; it < C1::ITL; ++it) {
      const int j = it * BPI + jl;
      if (j < L) {
        V odd[R];
        F::last_butterfly(j, lp, tw, lds, odd);
        const cx wj = tw_dir<DIR>(tw2[j]);
#pragma unroll
        for (int m = 0; m < R; ++m) {
          const cx w = m == 0 ? wj : cmul(wj, w16_half<DIR, T>(m));      // w^(j + m L)
          V lo, hi;
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const cx t = cmul(odd[m].c[c], w), e = pk.out[it][m].c[c];
            lo.c[c] = e + t;
            hi.c[c] = e - t;
          }
          io.store(C0, cl, j, m * L, lo);
          io.store(C0, cl, j, m * L + N1, hi);
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_sched_barrier(0);       // one row pair at a time: hoisting all sixteen results in front of the stores spills
#endif
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// ColPair: TWO adjacent tiles per workgroup, one after the other, so that every 128-byte line of the output is written whole.  An
// 8-column float32 tile row is 64 bytes -- half a line -- and the x pass's rows are a whole x plane (4 MiB) apart: a write-only sweep of
// such half lines runs at 3.4 TB/s on MI355X where whole lines reach 5.35 TB/s, and two half-line writes to one line merge only
// when they come from the same lane back to back.  So phase 0 transforms tile 2p and PARKS the last pass's
// outputs (R V16 per thread: 32 registers) instead of storing them, phase 1 transforms tile 2p + 1, and its last pass stores, row by
// row, the parked 16 bytes of tile 2p and its own 16 bytes of tile 2p + 1 -- 64 bytes apart in the same line -- from the same lane,
// one after the other.  Same arithmetic per tile as ColFFT: the field is bit for bit the single-tile kernel's.  (Round 2 had this
// form at 128 registers + spills and dropped it; with sigma shared between the rows +-ix the generation kernel needs 82.)
// ---------------------------------------------------------------------------
template <class C, int DIR, class IO>
struct ColPair {
  using F = ColFFT<C, DIR, IO>;
  using T = typename C::T;
  using cx = cplx<T>;
  using V = V16<T>;
  static constexpr int N = C::N, R = C::RL, L = N / R, CPL = C::CPL, LPR = C::LPR, BPI = C::BPI;
  static_assert(C::NPASS >= 2, "ColPair parks the outputs of a last pass through LDS");
  struct Park { V out[C::ITL][R]; };
  RF_HD static void last_park(int tid, const cx* tw, cx* lds, Park& pk) {
    const int lp = tid % LPR, jl = tid / LPR;
#pragma unroll
    for (int it = 0; it < C::ITL; ++it) {
      const int j = it * BPI + jl;
      if (j < L) F::last_butterfly(j, lp, tw, lds, pk.out[it]);
    }
  }
  // phase 1, last pass: row by row the parked vector of tile `tile_a` and this phase's vector of tile tile_a + 1
  RF_HD static void last_store(int tid, long long tile_a, const IO& io, const cx* tw, cx* lds, const Park& pk) {
    const int lp = tid % LPR, jl = tid / LPR;
    const long long C0a = tile_a * C::TC, C0b = C0a + C::TC;
    const int cl = lp * CPL;
#pragma unroll
    for (int it = 0; it < C::ITL; ++it) {
      const int j = it * BPI + jl;
      if (j < L) {
        V b[R];
        F::last_butterfly(j, lp, tw, lds, b);
#pragma unroll
        for (int m = 0; m < R; ++m) {
          io.store(C0a, cl, j, m * L, pk.out[it][m]);
          io.store(C0b, cl, j, m * L, b[m]);
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// Row (z) pass: complex FFT of length M = nz/2 per row + Hermitian (un)tangle
// ---------------------------------------------------------------------------
template <typename T_, int M_, int R1_, int R2_, int R3_, int NRT_, int NT_>
struct RowCfg {
  using T = T_;
  static constexpr int M = M_, R1 = R1_, R2 = R2_, R3 = R3_, NRT = NRT_, NT = NT_;
  static_assert(R1_ * R2_ * R3_ == M_, "radices must multiply to M");
  static constexpr int NPASS = (R2 == 1 ? 1 : (R3 == 1 ? 2 : 3));
  static constexpr int RL = (NPASS == 1 ? R1 : (NPASS == 2 ? R2 : R3));
  static constexpr int RS = M + ((M - 1) >> 3) + 1 + 1;       // LDS row stride (complex): pad16() of the last element + 2
  static constexpr int TILE_BYTES = (NPASS == 1 ? 0 : NRT * RS * (int)sizeof(cplx<T>));
  static constexpr int TW_BYTES = 2 * M * (int)sizeof(cplx<T>);    // twiddle table exp(2 pi i q / 2M), staged behind the tile
  static constexpr int LDS_BYTES = TILE_BYTES + TW_BYTES;
  static constexpr int L1 = M / R1;                           // butterflies per row in pass 1
  static constexpr int TPR1 = cmax(1, L1 / 2);                // threads per row in pass 1 (each owns a mirror pair)
  static constexpr int IT1 = ceil_div(NRT * TPR1, NT);
  static constexpr int IT2 = (NPASS == 3 ? ceil_div(NRT * (M / R2), NT) : 1);
  static constexpr int ITL = ceil_div(NRT * (M / RL), NT);
};

// c2r row IO over the device array viewed as complex [nrows][M] on input and
// real [nrows][2M] on output (same memory).  Accumulates sum / sum of squares.
// streaming (non-temporal) access to one complex element: the z pass touches every byte exactly once, so there is
// nothing to keep in the caches (a read+write sweep with the hint ran 6 % faster than without, tools/xbench.hip)
template <typename T> RF_HD cplx<T> stream_load(const cplx<T>* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef T vt __attribute__((ext_vector_type(2)));
  const vt v = __builtin_nontemporal_load(reinterpret_cast<const vt*>(p));
  return mk<T>(v.x, v.y);
#else
  return *p;
#endif
}
template <typename T> RF_HD void stream_store(cplx<T>* p, cplx<T> z) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef T vt __attribute__((ext_vector_type(2)));
  vt v; v.x = z.x; v.y = z.y;
  __builtin_nontemporal_store(v, reinterpret_cast<vt*>(p));
#else
  *p = z;
#endif
}

// Per-thread (sum, sum of squares) of the values a thread stores in the z pass.  float32 fields: FOUR float32 accumulators (the real and
// the imaginary slot of a complex store each have their own pair: 16 values per accumulator at nz = 1024), widened once at the end --
// the float64 form cost 8 float64-rate instructions per stored complex, a fifth of the pass's vector work, and the pass is not purely
// HBM-bound (it gained 10 % from cheaper arithmetic alone).  Rounding: 16 fused adds of like-signed squares per accumulator (<= 1e-6
// relative, unbiased), then 10^7 such partial sums added in float64: the field's rms to ~1e-9.  float64 fields accumulate in float64.
template <typename T> struct MomAcc;
template <> struct MomAcc<float> {
  float a1 = 0, b1 = 0, a2 = 0, b2 = 0;
  RF_HD void add(cplx<float> z) { a1 += z.x; b1 += z.y; a2 = fmaf(z.x, z.x, a2); b2 = fmaf(z.y, z.y, b2); }
  RF_HD double sum() const { return (double)a1 + (double)b1; }
  RF_HD double sumsq() const { return (double)a2 + (double)b2; }
};
template <> struct MomAcc<double> {
  double s1 = 0, s2 = 0;
  RF_HD void add(cplx<double> z) { s1 += z.x + z.y; s2 += z.x * z.x + z.y * z.y; }
  RF_HD double sum() const { return s1; }
  RF_HD double sumsq() const { return s2; }
};

template <typename T> struct PlainRowIO {
  cplx<T>* base;
  T scale;                       // 1 / (nx ny nz)
  int M_of;                      // complex elements per row (nz / 2)
  // (tile, row of the tile, lane's element, uniform element offset): the split lets a gathering IO keep its address arithmetic
  // on the scalar unit; here it is just row = tile * NRT + rl, element = kb + ko
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const { return load(tile * NRT + rl, kb + ko); }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const { store(tile * NRT + rl, nb + no, z, mom); }
  RF_HD int gather_seg_shift() const { return -1; }
  RF_HD cplx<T> load(long long row, int k) const { return stream_load(base + row * (long long)M_of + k); }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    z.x *= scale; z.y *= scale;
    stream_store(base + row * (long long)M_of + n, z);
    mom.add(z);
  }
};

// z pass of a slab-decomposed (multi-GPU) plan: this rank owns nxl x-planes.  After the all-to-all
// the receive buffer holds P blocks [src rank g][nxl][ny][nzl]; row (x, y) is gathered from its P
// segments of nzl = nz/(2P) complex (1 KiB each at 2048^3 / 8 GPUs) -- no separate local transpose.
// Output goes to a different buffer (the send buffer, free by then): dense real [nxl][ny][nz].
template <typename T> struct GatherRowIO {
  const cplx<T>* src;
  cplx<T>* dst;
  T scale;
  int M_of;                      // nz / 2
  int nzl;                       // kz planes per source rank
  long long seg_stride;          // complex elements between two source blocks = nxl * ny * nzl
  // (tile, row of the tile, lane's element, uniform element offset): the source block and the tile's row base are workgroup
  // uniform (scalar unit); nzl is a power of two (shift / mask instead of a division per element); streaming accesses like the
  // plain z pass (every byte is touched once)
  RF_HD int nzl_shift() const { return 31 - __builtin_clz((unsigned)nzl); }
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const {
    const int sh = nzl_shift(), mask = nzl - 1;
    const cplx<T>* ub = src + (long long)(ko >> sh) * seg_stride + tile * (long long)(NRT * nzl);
    const int kl = kb + (ko & mask);                       // (< nzl whenever nzl >= the pass's L: the block index is uniform then)
    return stream_load(ub + ((long long)(kl >> sh) * seg_stride + (long long)(rl * nzl + (kl & mask))));
  }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const {
    cplx<T>* ub = dst + tile * (long long)(NRT * M_of) + no;
    z.x *= scale; z.y *= scale;
    stream_store(reinterpret_cast<cplx<T>*>((size_t)ub + (size_t)((uint32_t)(rl * M_of + nb) * (uint32_t)sizeof(cplx<T>))), z);
    mom.add(z);
  }
  RF_HD int gather_seg_shift() const { return -1; }
  RF_HD cplx<T> load(long long row, int k) const {
    const int g = k >> nzl_shift(), kk = k & (nzl - 1);
    return stream_load(src + ((long long)g * seg_stride + row * (long long)nzl + kk));
  }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    z.x *= scale; z.y *= scale;
    stream_store(dst + (row * (long long)M_of + n), z);
    mom.add(z);
  }
};

// z pass reading the blocked intermediate X [xb][kt][iy][rb][tc] (xblock_*_geom) and writing the dense rows of W.  The NRT rows
// of a workgroup are consecutive ix of one (xb, iy): local row index (of the slab the launch covers) = (xb * ny + iy) * rb + r,
// so tile T covers rows T * NRT .. + NRT of ONE (xb, iy) (rb is a multiple of NRT) and every kz tile of theirs is one contiguous
// chunk of NRT * tc cells.  SEG_SHIFT = log2(tc): pass 1 deals its threads so that a wave reads whole chunks (RowC2R::pass_first).
template <typename T> struct XGatherRowIO {
  const cplx<T>* src;            // X, at the first x block of the slab
  cplx<T>* dst;                  // W, at the first x plane of the slab
  T scale;
  int M_of;                      // nz / 2
  int seg_shift;                 // log2(tc)
  int rb_shift, ny_shift;        // log2 of the rows of x per block and of ny (both powers of two on this path)
  long long kt_stride;           // cells between two kz tiles of a block = ny * rb * tc
  long long xb_stride;           // cells between two x blocks = (M / tc) * kt_stride
  RF_HD int gather_seg_shift() const { return seg_shift; }
  // A tile's NRT rows share (xb, iy) and are consecutive r (NRT divides rb): everything but the row-in-tile, the lane's element
  // and the kz tile of the uniform offset is workgroup-uniform (scalar unit); the lane part fits 32 bits (one x block).
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const {
    const long long t0 = tile * NRT, q = t0 >> rb_shift, r0 = t0 & ((1LL << rb_shift) - 1);
    const long long xb = q >> ny_shift, iy = q & ((1LL << ny_shift) - 1);
    const int mask = (1 << seg_shift) - 1;
    const cplx<T>* ub = src + xb * xb_stride + ((((iy << rb_shift) + r0)) << seg_shift) + (long long)(ko >> seg_shift) * kt_stride;
    const int kl = kb + (ko & mask);                  // (ko is a multiple of the segment length in the product: kl == kb)
    const uint32_t lane = ((uint32_t)rl << seg_shift) + (uint32_t)(kl >> seg_shift) * (uint32_t)kt_stride + (uint32_t)(kl & mask);
    return stream_load(reinterpret_cast<const cplx<T>*>((size_t)ub + (size_t)(lane * (uint32_t)sizeof(cplx<T>))));
  }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const {
    const long long t0 = tile * NRT, q = t0 >> rb_shift, r0 = t0 & ((1LL << rb_shift) - 1);
    const long long xb = q >> ny_shift, iy = q & ((1LL << ny_shift) - 1);
    cplx<T>* ub = dst + (((((xb << rb_shift) + r0) << ny_shift) + iy)) * (long long)M_of + no;
    const uint32_t lane = (uint32_t)rl * ((uint32_t)M_of << ny_shift) + (uint32_t)nb;
    z.x *= scale; z.y *= scale;
    stream_store(reinterpret_cast<cplx<T>*>((size_t)ub + (size_t)(lane * (uint32_t)sizeof(cplx<T>))), z);
    mom.add(z);
  }
  RF_HD cplx<T> load(long long row, int k) const {
    const long long r = row & ((1LL << rb_shift) - 1), q = row >> rb_shift;      // q = xb * ny + iy
    const long long xb = q >> ny_shift, iy = q & ((1LL << ny_shift) - 1);
    return stream_load(src + xb * xb_stride + (long long)(k >> seg_shift) * kt_stride + ((((iy << rb_shift) + r)) << seg_shift) + (k & ((1 << seg_shift) - 1)));
  }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    const long long r = row & ((1LL << rb_shift) - 1), q = row >> rb_shift;
    const long long xb = q >> ny_shift, iy = q & ((1LL << ny_shift) - 1);
    z.x *= scale; z.y *= scale;
    stream_store(dst + (((((xb << rb_shift) + r) << ny_shift) + iy)) * (long long)M_of + n, z);
    mom.add(z);
  }
};

// z pass with the lognormal map in its epilogue: rho = exp(delta * Ap_z) * Bp_z with the float64 tables Ap = sqrt(log t) / sigma,
// Bp = density / sqrt(t), t = 1 + (sigma growth_z)^2, formed on the device from the y pass's Parseval sum (AccColIO,
// lognormal_tables_kernel).  The reference does the same map as four in-place numpy statements with a rounding to the array
// dtype after each (cosmotools.py:216-220, then generate.py:273); here the two divisions are folded into the tables (a float64
// division costs ~15 instructions per element and the pass has 2 x 10^9 of them): the result is within a few ulp of the
// argument of exp of the reference's chain (<= 1e-15 relative for float64 fields, 3e-7 for float32 ones; rf_lognormal is the
// rounding-exact, unfused form).  Element n of a row holds the reals z = 2n, 2n + 1; the tables are 16 KB, L1-resident.
RF_HD float exp_t(float x) { return expf(x); }
RF_HD double exp_t(double x) { return exp(x); }
// float64 plans: exp(t ln2 / 64) for an argument already in units of ln2 / 64 (the table Ap carries the factor 64 / ln2, lognormal_ap_unit):
// t = k + f, |f| <= 1/2, k = 64 e + j: 2^e * 2^(j/64) * exp(f ln2/64), the middle factor from a 64-entry table in LDS (rf_exp2_tab.h,
// correctly rounded), the last a degree-4 polynomial (|r| <= 0.0055: the first dropped term is 4e-14; round 4: degree 5).  11 float64-rate instructions
// and one ds_read_b64 per element where the library's exp takes ~22 (no table: a degree-11 polynomial, range checks); the z pass of a
// float64 plan issues 16 of them per thread.  |error| <= 4e-14 relative (the dropped term) + 1 ulp of the result + the rounding of t
// (ulp(t) ln2 / 128 <= 6e-16 at |x| = 5.5, the same size as the rounding of the product delta * Ap that both forms share).  Out-of-range arguments saturate through
// the conversion and ldexp (inf / 0), NaN propagates through r.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const double rf_exp2_tab[64] = {RF_EXP2_TAB_VALUES};
#else
static const double rf_exp2_tab[64] = {RF_EXP2_TAB_VALUES};
#endif
// the factor the float64 z pass expects in Ap: 64 / ln 2 (the unit of exp_scaled64) times the transform's 1 / (nx ny nz)
template <typename T> RF_HD double lognormal_ap_unit(double scale) { return sizeof(T) == 8 ? 0x1.71547652b82fep+6 /* 64 / ln 2 */ * scale : 1.0; }
RF_HD int exp_k_of(double kf) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)kf;                                         // (v_cvt_i32_f64 saturates)
#else
  return kf > 1e9 ? 1000000000 : (kf < -1e9 ? -1000000000 : (kf == kf ? (int)kf : 0));
#endif
}
// 2^(k >> 6) * tj * exp(c f), c = ln 2 / 64, tj = 2^((k & 63) / 64)
RF_HD double exp_finish(double f, double tj, int k) {
  // exp(c f) - 1 = f (c + f (c^2/2 + f (c^3/6 + f c^4/24))): |c f| <= ln2/128, so the first dropped term (c f)^5/120 is <= 3.9e-14 of
  // the result -- the fused map is checked against the reference's chain to 1e-12, its tolerance is 1e-11 (rounds 3 - 4 carried the
  // fifth-order term too: one more float64 fma per element, 2 x 10^9 of them per 1024^3 field)
  double q = 0x1.3b2ab6fba4e77p-31 /* c^4/24 */;
  q = __builtin_fma(f, q, 0x1.c6b08d704a0c0p-23 /* c^3/6 */);
  q = __builtin_fma(f, q, 0x1.ebfbdff82c58fp-15 /* c^2/2 */);
  q = __builtin_fma(f, q, 0x1.62e42fefa39efp-7 /* c */);
  return __builtin_ldexp(__builtin_fma(tj, f * q, tj), k >> 6);
}
template <int STRIDE = 1> RF_HD double exp_scaled64(double t, const double* tab) {
  const double kf = __builtin_rint(t);
  const double f = t - kf;                                // exact
  const int k = exp_k_of(kf);
  const double tj = tab ? tab[(k & 63) * STRIDE] : 1.0;    // (tab is never null in the product)
  return exp_finish(f, tj, k);
}
// Where the table lives: the LDS row image skips every ninth complex (pad16), so row 0 of a tile of rows of M >= 512 complex128 has 64
// unused 16-byte slots at 9 j + 8 -- entry j goes there (stage(), 64 threads, in front of the kernel's first barrier; nothing else
// ever touches those slots).  Measured at 1024^3 float64 on MI355X (z pass, plain 3.12 ms): the library's exp 3.65, the table read
// from global memory 3.62 (a 64-lane gather per element), from 512 more bytes of LDS 4.48 (the pass fills a third of the CU's LDS
// to within one allocation unit: two workgroups per CU instead of three), from the pad slots: see DESIGN.md section 3.9.
template <typename T, int SPARE = 0> struct LognormalRowIO {
  static_assert(SPARE == 0 || sizeof(T) == 8, "the exp table is the float64 plans'");
  cplx<T>* base;
  T scale;                       // 1 / (nx ny nz)
  int M_of;
  const double* Ap;              // [2 M] sqrt(log t_z) / sigma  (float64 plans: times lognormal_ap_unit)
  const double* Bp;              // [2 M] density_z / sqrt(t_z)
  const double* etab = nullptr;  // SPARE: rf_exp2_tab in the pad slots of the tile's row 0
  static constexpr bool WANTS_STAGE = SPARE != 0;
  static constexpr int ESTRIDE = SPARE ? 18 : 1;        // doubles between two entries
  template <class C> RF_HD void stage(int tid, void* lds) {
    static_assert(!SPARE || (C::NPASS >= 2 && C::RS >= 9 * 63 + 8 + 1), "64 pad slots in row 0");
    double* l = reinterpret_cast<double*>(lds) + 16;
    if (tid < 64) l[18 * tid] = rf_exp2_tab[tid];
    etab = l;
  }
  RF_HD int gather_seg_shift() const { return -1; }
  RF_HD float map(float d, int z) const {
    d = (float)((double)d * Ap[z]);
    d = exp_t(d);
    return (float)((double)d * Bp[z]);
  }
  RF_HD double map(double d, int z) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const double* tb = SPARE ? etab : rf_exp2_tab;
#else
    const double* tb = (SPARE && etab) ? etab : rf_exp2_tab;        // (the emulator has no staging step)
    if (!(SPARE && etab)) return exp_scaled64<1>(d * Ap[z], tb) * Bp[z];
#endif
    return exp_scaled64<ESTRIDE>(d * Ap[z], tb) * Bp[z];
  }
  RF_HD cplx<T> load(long long row, int k) const { return stream_load(base + row * (long long)M_of + k); }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    if (sizeof(T) == 8) {          // (float64 plans: 1 / (nx ny nz) is part of Ap too)
      z.x = map(z.x, 2 * n);
      z.y = map(z.y, 2 * n + 1);
    } else {
      z.x = map(z.x * scale, 2 * n);
      z.y = map(z.y * scale, 2 * n + 1);
    }
    stream_store(base + row * (long long)M_of + n, z);
    mom.add(z);
  }
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const { return load(tile * NRT + rl, kb + ko); }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const { store(tile * NRT + rl, nb + no, z, mom); }
  // the last pass of a multi-pass row: the 2 R entries of Ap and Bp first (RowC2R::pass_last), then all R outputs in one call --
  // for float64 in phases (arguments and table reads of all 2 R elements, then the polynomials, then the stores)
  static constexpr bool HAS_PRE = true;
  template <int R> struct Pre { double a[2 * R], b[2 * R]; };
  template <int R> RF_HD void prefetch(int j, int L, Pre<R>& p) const {
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int z = 2 * (j + m * L);
      p.a[2 * m] = Ap[z]; p.a[2 * m + 1] = Ap[z + 1];
      p.b[2 * m] = Bp[z]; p.b[2 * m + 1] = Bp[z + 1];
    }
  }
  template <int NRT, int R> RF_HD void store_row(long long tile, int rl, int j, int L, const cplx<T>* v, MomAcc<T>& mom, const Pre<R>& p) const {
    cplx<T>* const out = base + (tile * NRT + rl) * (long long)M_of + j;
    if constexpr (sizeof(T) == 8) {
#if defined(__HIP_DEVICE_COMPILE__)
      const double* tb = SPARE ? etab : rf_exp2_tab;
      constexpr int ES = ESTRIDE;
#else
      const double* tb = (SPARE && etab) ? etab : rf_exp2_tab;
      const int ES = (SPARE && etab) ? ESTRIDE : 1;
#endif
      double f[2 * R], tj[2 * R];
      int k[2 * R];
#pragma unroll
      for (int i = 0; i < 2 * R; ++i) {
        const double t = ((i & 1) ? v[i / 2].y : v[i / 2].x) * p.a[i];
        const double kf = __builtin_rint(t);
        f[i] = t - kf;
        k[i] = exp_k_of(kf);
        tj[i] = tb[(k[i] & 63) * ES];
      }
#pragma unroll
      for (int m = 0; m < R; ++m) {
        cplx<T> z;
        z.x = (T)(exp_finish(f[2 * m], tj[2 * m], k[2 * m]) * p.b[2 * m]);
        z.y = (T)(exp_finish(f[2 * m + 1], tj[2 * m + 1], k[2 * m + 1]) * p.b[2 * m + 1]);
        stream_store(out + m * L, z);
        mom.add(z);
      }
    } else {
#pragma unroll
      for (int m = 0; m < R; ++m) {
        cplx<T> z;
        z.x = (T)((double)exp_t((T)((double)(v[m].x * scale) * p.a[2 * m])) * p.b[2 * m]);
        z.y = (T)((double)exp_t((T)((double)(v[m].y * scale) * p.a[2 * m + 1])) * p.b[2 * m + 1]);
        stream_store(out + m * L, z);
        mom.add(z);
      }
    }
  }
};
// does a row IO stage something into the tile's spare LDS slots at the start of the kernel?
template <class IO, class = void> struct row_io_wants_stage { static constexpr bool value = false; };
template <class IO> struct row_io_wants_stage<IO, typename std::enable_if<IO::WANTS_STAGE>::type> { static constexpr bool value = true; };

// z pass whose store multiplies plane z by a per-z factor (float64 table, the rounding of rf_scale_z on the stored field): the
// light-cone weighting G(z) / (1 + z) of calculate_newtonian_potential (generate.py:344-347) without a sweep of its own
template <typename T> struct ScaleZRowIO {
  cplx<T>* base;
  T scale;                       // 1 / (nx ny nz)
  int M_of;
  const double* Sz;              // [2 M]
  RF_HD int gather_seg_shift() const { return -1; }
  RF_HD cplx<T> load(long long row, int k) const { return stream_load(base + row * (long long)M_of + k); }
  RF_HD void store(long long row, int n, cplx<T> z, MomAcc<T>& mom) const {
    z.x = (T)((double)(z.x * scale) * Sz[2 * n]);
    z.y = (T)((double)(z.y * scale) * Sz[2 * n + 1]);
    stream_store(base + row * (long long)M_of + n, z);
    mom.add(z);
  }
  template <int NRT> RF_HD cplx<T> load2(long long tile, int rl, int kb, int ko) const { return load(tile * NRT + rl, kb + ko); }
  template <int NRT> RF_HD void store2(long long tile, int rl, int nb, int no, cplx<T> z, MomAcc<T>& mom) const { store(tile * NRT + rl, nb + no, z, mom); }
  // (the table entries in front of the last pass, as LognormalRowIO)
  static constexpr bool HAS_PRE = true;
  template <int R> struct Pre { double s[2 * R]; };
  template <int R> RF_HD void prefetch(int j, int L, Pre<R>& p) const {
#pragma unroll
    for (int m = 0; m < R; ++m) { p.s[2 * m] = Sz[2 * (j + m * L)]; p.s[2 * m + 1] = Sz[2 * (j + m * L) + 1]; }
  }
  template <int NRT, int R> RF_HD void store_row(long long tile, int rl, int j, int L, const cplx<T>* v, MomAcc<T>& mom, const Pre<R>& p) const {
    cplx<T>* const out = base + (tile * NRT + rl) * (long long)M_of + j;
#pragma unroll
    for (int m = 0; m < R; ++m) {
      cplx<T> z;
      z.x = (T)((double)(v[m].x * scale) * p.s[2 * m]);
      z.y = (T)((double)(v[m].y * scale) * p.s[2 * m + 1]);
      stream_store(out + m * L, z);
      mom.add(z);
    }
  }
};

// does a row IO fetch table entries ahead of the last pass (IO::Pre<R>, prefetch<R>(), store_row<NRT, R>())?
struct RowNoPre {};
template <class IO, int R, class = void> struct row_io_pre { static constexpr bool value = false; using type = RowNoPre; };
template <class IO, int R> struct row_io_pre<IO, R, typename std::enable_if<IO::HAS_PRE>::type> {
  static constexpr bool value = true;
  using type = typename IO::template Pre<R>;
};

// tw = exp(+2 pi i q / (2M)), q in [0, 2M): t_k = tw[k], w_M^q = tw[2q]
template <class C, class IO>
struct RowC2R {
  using T = typename C::T;
  using cx = cplx<T>;
  static constexpr int M = C::M, NT = C::NT;
  static constexpr int DIR = +1;

  struct Regs { cx v[C::IT2][cmax(C::R2, 1)]; MomAcc<T> mom; };

  RF_HD static cx* lds_at(cx* lds, int rl, int i) { return lds + (long long)rl * C::RS + pad16(i); }
  RF_HD static cx* lds_tw(cx* lds) { return lds + C::TILE_BYTES / (int)sizeof(cx); }
  // The twiddles in LDS: NOT the plain table exp(2 pi i q / 2M) the kernel gets, but three tables cut from it, each in the order its
  // pass reads it, 2M entries in all (round 5).  Read from the plain table, the middle pass's w_(R1 R2)^(c m) sit 16 m entries apart
  // -- 128 m bytes: a 4- or 8-way bank conflict per read -- and the last pass's w_M^(m j) 2 m entries apart (2- to 8-way); on the
  // z pass of 1024^3 float32 40 % of the LDS-array cycles were conflict cycles (SQ_LDS_BANK_CONFLICT 1.08e7 of SQ_LDS_IDX_ACTIVE
  // 2.70e7 per launch, profiles/r05_a_pmc_sq_*), two thirds of them from these reads.
  //   U [k]            = t_k = tw[k], k < M                       : the untangle (consecutive lanes, consecutive k)
  //   LT[(m-1) LL + j] = tw[2 m j],   1 <= m < RL, j < LL = M / RL : the last pass (consecutive lanes, consecutive j)
  //   MT[(m-1) R1 + c] = tw[2 m c M / (R1 R2)], 1 <= m < R2, c < R1 : the middle pass (lane j reads entry c = j mod R1: broadcast)
  static constexpr int LL = M / C::RL;
  static constexpr int TW_LT = M, TW_MT = M + (C::NPASS >= 2 ? (C::RL - 1) * LL : 0);
  static constexpr int TW_END = TW_MT + (C::NPASS == 3 ? (C::R2 - 1) * C::R1 : 0);
  static_assert(TW_END <= 2 * M, "the three tables fit the space of the plain one");
  RF_HD static int tw_source(int e) {                  // entry e of the LDS image <- entry tw_source(e) of the plain table
    if (e < TW_LT) return e;
    if (e < TW_MT) { const int r = e - TW_LT; return 2 * (r / LL + 1) * (r % LL); }
    if (e < TW_END) { const int r = e - TW_MT; return 2 * (r / C::R1 + 1) * (r % C::R1) * (M / (C::R1 * cmax(C::R2, 1))); }
    return 0;
  }
  RF_HD static cx tw_last(const cx* ltw, int m, int j) { return ltw[TW_LT + (m - 1) * LL + j]; }
  RF_HD static cx tw_mid(const cx* ltw, int m, int j) {
    return ltw[TW_MT + (m - 1) * C::R1 + (j % C::R1)];
  }
  // The table (needed by pass 1 already: the untangle) goes global -> registers (tw_fetch), then the row data
  // (pass_first_load), then registers -> LDS (tw_stage) and a barrier: both trips to memory are in flight together, and the
  // older one -- the small table -- is the one that is waited for first (loads retire in order).
  static constexpr int TWPT = ceil_div(2 * M, NT);
  struct TwRegs { cx v[TWPT]; };
  RF_HD static void tw_fetch(int tid, const cx* tw, TwRegs& t) {
#pragma unroll
    for (int k = 0; k < TWPT; ++k) t.v[k] = tw[tw_source((tid + k * NT) & (2 * M - 1))];       // (M is a power of two: no branch, no undefined slot)
  }
  RF_HD static void tw_stage(int tid, cx* lds, const TwRegs& t) {
    cx* l = lds_tw(lds);
#pragma unroll
    for (int k = 0; k < TWPT; ++k)
      if (tid + k * NT < 2 * M) l[tid + k * NT] = t.v[k];
  }
  // prologue (emulator; the kernel calls the pieces): stage the twiddle table in LDS (a barrier follows)
  RF_HD static void prologue(int tid, const cx* tw, cx* lds) {
    TwRegs t;
    tw_fetch(tid, tw, t);
    tw_stage(tid, lds, t);
  }

  // pass 1 outputs: LDS (NPASS > 1) or global (NPASS == 1)
  template <int R>
  RF_HD static void emit(int rl, long long row, int idx, const cx* v, const IO& io, cx* lds, Regs& r) {
#pragma unroll
    for (int m = 0; m < R; ++m) {
      if (C::NPASS == 1) io.template store2<C::NRT>(row / C::NRT, rl, idx, m, v[m], r.mom);
      else if (R % 8 == 0) lds_at(lds, rl, idx)[m + (m >> 3)] = v[m];      // (idx is a multiple of R: the padding of idx + m splits, one base + immediates)
      else *lds_at(lds, rl, idx + m) = v[m];
    }
  }

  // which (row of the tile, butterfly pair) thread `w` of pass 1 owns
  RF_HD static void first_owner(int w, const IO& io, int& rl, int& q) {
    rl = w / C::TPR1;
    q = w % C::TPR1;
    // gathering IO: 2^sg consecutive k of a row are one segment of the source and the segments of the tile's NRT rows are
    // adjacent, so thread w takes k-in-segment = w % 2^sg, row = (w >> sg) % NRT, segment = w / (NRT 2^sg): a wave's loads
    // then cover whole chunks of NRT segments instead of one segment in each of many blocks
    const int sg = io.gather_seg_shift();
    if (sg >= 0 && C::TPR1 % (1 << sg) == 0) {
      rl = (w >> sg) % C::NRT;
      q = ((w >> sg) / C::NRT << sg) + (w & ((1 << sg) - 1));
      if (q >= C::TPR1) rl = C::NRT;         // (threads beyond NRT * TPR1: idle, as in the plain mapping)
    }
  }
  struct In { cx A[C::IT1][C::R1], B[C::IT1][C::R1]; };
  // pass 1, first half: the mirror pair's inputs global -> registers
  RF_HD static void pass_first_load(int tid, long long tile, long long nrows, const IO& io, In& in) {
    constexpr int R = C::R1, L = C::L1;
#pragma unroll
    for (int it = 0; it < C::IT1; ++it) {
      int rl, q;
      first_owner(it * NT + tid, io, rl, q);
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        const int ja = q, jb = (q == 0) ? L / 2 : L - q;
#pragma unroll
        for (int m = 0; m < R; ++m) in.A[it][m] = io.template load2<C::NRT>(tile, rl, ja, m * L);
        if (L >= 2) {
#pragma unroll
          for (int m = 0; m < R; ++m) in.B[it][m] = io.template load2<C::NRT>(tile, rl, jb, m * L);
        }
      }
    }
  }
  // pass 1, second half: untangle -> R1 butterflies of the mirror pair -> LDS
  RF_HD static void pass_first_compute(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds, Regs& r, const In& in) {
    constexpr int R = C::R1, L = C::L1;
    r.mom = MomAcc<T>();
#pragma unroll
    for (int it = 0; it < C::IT1; ++it) {
      int rl, q;
      first_owner(it * NT + tid, io, rl, q);
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        const bool self = (q == 0);
        const int ja = q;
        const int jb = self ? L / 2 : L - q;
        const bool has_b = (L >= 2);
        cx A[R], B[R], ZA[R], ZB[R];
#pragma unroll
        for (int m = 0; m < R; ++m) { A[m] = in.A[it][m]; B[m] = in.B[it][m]; }
#pragma unroll
        for (int m = 0; m < R; ++m) {
          const cx ta = tw[ja + m * L];
          if (self) {
            if (m == 0) ZA[0] = mk<T>(A[0].x + A[0].y, A[0].x - A[0].y);   // (DC + Nyq) + i (DC - Nyq)
            else ZA[m] = c2r_untangle(A[m], A[R - m], ta);
          } else {
            ZA[m] = c2r_untangle(A[m], B[R - 1 - m], ta);
          }
          if (has_b) {
            const cx tb = tw[jb + m * L];
            ZB[m] = self ? c2r_untangle(B[m], B[R - 1 - m], tb) : c2r_untangle(B[m], A[R - 1 - m], tb);
          }
        }
        DFT<R, DIR>::run(ZA);
        emit<R>(rl, row, ja * R, ZA, io, lds, r);
        if (has_b) {
          DFT<R, DIR>::run(ZB);
          emit<R>(rl, row, jb * R, ZB, io, lds, r);
        }
      }
    }
  }
  // pass 1: global -> untangle -> R1 butterflies of the mirror pair -> LDS
  RF_HD static void pass_first(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds, Regs& r) {
    In in;
    pass_first_load(tid, tile, nrows, io, in);
    pass_first_compute(tid, tile, nrows, io, tw, lds, r, in);
  }

  RF_HD static void pass_mid_read(int tid, const cx* tw, cx* lds, Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
        const cx* const rd = lds_at(lds, rl, j);                              // L % 8 == 0: pad16(j + m L) = pad16(j) + m (L + L / 8)
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = (L % 8 == 0) ? rd[m * (L + L / 8)] : *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, tw_mid(tw, m, j));
          r.v[it][m] = x;
        }
        DFT<R, DIR>::run(r.v[it]);
      }
    }
  }
  RF_HD static void pass_mid_write(int tid, cx* lds, const Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
        const int ob = stockham_out_base(j, Ns, R);
        cx* const wr = lds_at(lds, rl, ob);                                   // Ns % 8 == 0: pad16(ob + m Ns) = pad16(ob) + m (Ns + Ns / 8)
#pragma unroll
        for (int m = 0; m < R; ++m) {
          if (Ns % 8 == 0) wr[m * (Ns + Ns / 8)] = r.v[it][m];
          else *lds_at(lds, rl, ob + m * Ns) = r.v[it][m];
        }
      }
    }
  }

  RF_HD static void pass_last(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds, Regs& r) {
    constexpr int R = C::RL, L = M / R;
#pragma unroll
    for (int it = 0; it < C::ITL; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        cx v[R];
        // IOs whose store needs per-z table entries (LognormalRowIO, ScaleZRowIO) fetch the thread's 2 R entries HERE, in front of the
        // LDS reads and the butterfly, and store the R outputs in one call: written per element (load table -> map -> store) the
        // epilogue is R round trips in a row -- on gfx950 a load issued behind a store is waited for through the same counter
        // as the store (vmcnt, in order), so every element waited for the previous element's write to retire
        typename row_io_pre<IO, R>::type pre;
        if constexpr (row_io_pre<IO, R>::value) io.template prefetch<R>(j, L, pre);
        const cx* const rd = lds_at(lds, rl, j);
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = (L % 8 == 0) ? rd[m * (L + L / 8)] : *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, tw_last(tw, m, j));
          v[m] = x;
        }
        DFT<R, DIR>::run(v);
        if constexpr (row_io_pre<IO, R>::value) {
          io.template store_row<C::NRT, R>(tile, rl, j, L, v, r.mom, pre);
        } else {
#pragma unroll
          for (int m = 0; m < R; ++m) io.template store2<C::NRT>(tile, rl, j, m * L, v[m], r.mom);
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// Forward row pass: r2c along z (transform.py:199-206,270 -- np.fft.rfftn's last axis)
// ---------------------------------------------------------------------------
// The real row x[0..nz) is viewed as M = nz/2 complex z[m] = x[2m] + i x[2m+1]; Z = FFT_M(z) (forward);
// X[k] = (Z[k] + conj Z[M-k])/2 - (i/2) conj(t_k) (Z[k] - conj Z[M-k]).  The tangle needs the mirror
// pair (k, M-k) of the FFT *output*, so the LAST pass gives one thread the butterfly pair (j, L - j)
// (the mirror image of RowC2R's first pass).  Output in place: M complex per row, element 0 packs
// (X[0], X[M]) -- both are real.
template <typename T> struct PlainRowFwdIO {
  cplx<T>* base;
  int M_of;
  RF_HD cplx<T> load(long long row, int k) const { return base[row * (long long)M_of + k]; }
  RF_HD void store(long long row, int k, cplx<T> z) const { base[row * (long long)M_of + k] = z; }
};

template <class C, class IO>
struct RowR2C {
  using T = typename C::T;
  using cx = cplx<T>;
  static constexpr int M = C::M, NT = C::NT;
  static constexpr int DIR = -1;
  // the paired LAST pass needs L/2 threads per row (or 1)
  static constexpr int LL = M / C::RL;
  static constexpr int TPRL = cmax(1, LL / 2);
  static constexpr int ITF = ceil_div(C::NRT * (M / C::R1), NT);
  static constexpr int ITLP = ceil_div(C::NRT * TPRL, NT);

  struct Regs { cx v[C::IT2][cmax(C::R2, 1)]; };

  RF_HD static cx* lds_at(cx* lds, int rl, int i) { return lds + (long long)rl * C::RS + pad16(i); }
  RF_HD static cx* lds_tw(cx* lds) { return lds + C::TILE_BYTES / (int)sizeof(cx); }
  RF_HD static void prologue(int tid, const cx* tw, cx* lds) {
    cx* l = lds_tw(lds);
    for (int i = tid; i < 2 * M; i += NT) l[i] = tw[i];
  }

  // pass 1 (only when NPASS >= 2): global -> R1 butterfly -> LDS
  RF_HD static void pass_first(int tid, long long tile, long long nrows, const IO& io, cx* lds) {
    constexpr int R = C::R1, L = M / R;
#pragma unroll
    for (int it = 0; it < ITF; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        cx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = io.load(row, j + m * L);
        DFT<R, DIR>::run(v);
#pragma unroll
        for (int m = 0; m < R; ++m) *lds_at(lds, rl, j * R + m) = v[m];
      }
    }
  }

  RF_HD static void pass_mid_read(int tid, const cx* tw, cx* lds, Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, cconj(tw[2 * stockham_tw_index(j, m, Ns, R, M)]));
          r.v[it][m] = x;
        }
        DFT<R, DIR>::run(r.v[it]);
      }
    }
  }
  RF_HD static void pass_mid_write(int tid, cx* lds, const Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
        const int ob = stockham_out_base(j, Ns, R);
#pragma unroll
        for (int m = 0; m < R; ++m) *lds_at(lds, rl, ob + m * Ns) = r.v[it][m];
      }
    }
  }

  // last pass: (LDS | global when NPASS == 1) -> RL butterflies of the mirror pair -> tangle -> global
  RF_HD static void pass_last(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds) {
    constexpr int R = C::RL, L = LL;
#pragma unroll
    for (int it = 0; it < ITLP; ++it) {
      const int w = it * NT + tid;
      const int rl = w / TPRL, q = w % TPRL;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        const bool self = (q == 0);
        const int ja = q, jb = self ? L / 2 : L - q;
        const bool has_b = (L >= 2);
        cx A[R], B[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = (C::NPASS == 1) ? io.load(row, ja + m * L) : *lds_at(lds, rl, ja + m * L);
          if (C::NPASS > 1 && m > 0) x = cmul(x, cconj(tw[2 * m * ja]));
          A[m] = x;
        }
        DFT<R, DIR>::run(A);                      // A[m] = Z[ja + m L]
        if (has_b) {
#pragma unroll
          for (int m = 0; m < R; ++m) {
            cx x = (C::NPASS == 1) ? io.load(row, jb + m * L) : *lds_at(lds, rl, jb + m * L);
            if (C::NPASS > 1 && m > 0) x = cmul(x, cconj(tw[2 * m * jb]));
            B[m] = x;
          }
          DFT<R, DIR>::run(B);                    // B[m] = Z[jb + m L]
        }
        // mirror of k = ja + m L is M - k = jb + (R-1-m) L  (ja >= 1); for ja = 0: (R - m) L, and k = 0 <-> M
#pragma unroll
        for (int m = 0; m < R; ++m) {
          const int ka = ja + m * L;
          if (self) {
            if (m == 0) io.store(row, 0, mk<T>(A[0].x + A[0].y, A[0].x - A[0].y));   // (X[0], X[M]) packed
            else io.store(row, ka, r2c_tangle(A[m], A[R - m], tw[ka]));
          } else {
            io.store(row, ka, r2c_tangle(A[m], B[R - 1 - m], tw[ka]));
          }
          if (has_b) {
            const int kb = jb + m * L;
            io.store(row, kb, self ? r2c_tangle(B[m], B[R - 1 - m], tw[kb]) : r2c_tangle(B[m], A[R - 1 - m], tw[kb]));
          }
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// Plain complex row pass (unpacked c2c plans, transform.py:207-213,266-270): FFT of length M = nz along
// the contiguous axis, either direction.  tw = exp(+2 pi i q / M), q in [0, M) (conjugated for DIR = -1).
// ---------------------------------------------------------------------------
template <typename T> struct ScaledRowIO {
  cplx<T>* base;
  int M_of;                      // complex elements per row (nz)
  T scale;                       // 1 (forward) or 1 / (nx ny nz) (inverse, numpy normalisation)
  RF_HD cplx<T> load(long long row, int k) const { return base[row * (long long)M_of + k]; }
  RF_HD void store(long long row, int k, cplx<T> z) const {
    z.x *= scale; z.y *= scale;
    base[row * (long long)M_of + k] = z;
  }
};

template <class C, int DIR_, class IO>
struct RowC2C {
  using T = typename C::T;
  using cx = cplx<T>;
  static constexpr int M = C::M, NT = C::NT, DIR = DIR_;
  static constexpr int ITF = ceil_div(C::NRT * (M / C::R1), NT);

  struct Regs { cx v[C::IT2][cmax(C::R2, 1)]; };

  RF_HD static cx* lds_at(cx* lds, int rl, int i) { return lds + (long long)rl * C::RS + pad16(i); }
  RF_HD static cx* lds_tw(cx* lds) { return lds + C::TILE_BYTES / (int)sizeof(cx); }
  RF_HD static void prologue(int tid, const cx* tw, cx* lds) {
    cx* l = lds_tw(lds);
    for (int i = tid; i < M; i += NT) l[i] = tw[i];
  }

  // pass 1: global -> R1 butterfly -> LDS (or straight back to global when M == R1)
  RF_HD static void pass_first(int tid, long long tile, long long nrows, const IO& io, cx* lds) {
    constexpr int R = C::R1, L = M / R;
#pragma unroll
    for (int it = 0; it < ITF; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        cx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = io.load(row, j + m * L);
        DFT<R, DIR>::run(v);
#pragma unroll
        for (int m = 0; m < R; ++m) {
          if (C::NPASS == 1) io.store(row, j * R + m, v[m]);
          else *lds_at(lds, rl, j * R + m) = v[m];
        }
      }
    }
  }

  RF_HD static void pass_mid_read(int tid, const cx* tw, cx* lds, Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, tw_dir<DIR>(tw[stockham_tw_index(j, m, Ns, R, M)]));
          r.v[it][m] = x;
        }
        DFT<R, DIR>::run(r.v[it]);
      }
    }
  }
  RF_HD static void pass_mid_write(int tid, cx* lds, const Regs& r) {
    constexpr int R = C::R2, L = M / R, Ns = C::R1;
#pragma unroll
    for (int it = 0; it < C::IT2; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      if (rl < C::NRT) {
        const int ob = stockham_out_base(j, Ns, R);
#pragma unroll
        for (int m = 0; m < R; ++m) *lds_at(lds, rl, ob + m * Ns) = r.v[it][m];
      }
    }
  }

  // last pass (NPASS >= 2): LDS -> RL butterfly -> global
  RF_HD static void pass_last(int tid, long long tile, long long nrows, const IO& io, const cx* tw, cx* lds) {
    constexpr int R = C::RL, L = M / R;
#pragma unroll
    for (int it = 0; it < C::ITL; ++it) {
      const int w = it * NT + tid;
      const int rl = w / L, j = w % L;
      const long long row = tile * C::NRT + rl;
      if (rl < C::NRT && row < nrows) {
        cx v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
          cx x = *lds_at(lds, rl, j + m * L);
          if (m > 0) x = cmul(x, tw_dir<DIR>(tw[m * j]));
          v[m] = x;
        }
        DFT<R, DIR>::run(v);
#pragma unroll
        for (int m = 0; m < R; ++m) io.store(row, j + m * L, v[m]);
      }
    }
  }
};

}  // namespace rf
