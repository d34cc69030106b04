// rf_fft_col.h -- strided (x / y) passes: the plain IOs, ColFFT, Col2 (two half-length transforms per tile), ColPair (tile pairs), the direct-exchange IOs (part of rf_fft.h: include that)
#pragma once
#include "rf_fft.h"

namespace rf {

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

}  // namespace rf
