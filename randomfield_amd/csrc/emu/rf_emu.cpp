// rf_emu.cpp -- CPU emulator of the HIP kernels (TEST TOOLING, build container).
//
// Runs the exact phase functions of rf_fft.h / rf_core.h that the gfx950
// kernels inline, one "thread" after another with a plain array standing in
// for LDS and a loop boundary standing in for each workgroup barrier.  It
// checks index math, twiddles, Hermitian packing and the generation rules
// against the oracle without a GPU.  It is not a product path: nothing in
// randomfield_amd/ loads it.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../rf_configs.h"
#include "../rf_host.h"
#include "../rf_generic.h"

using namespace rf;

namespace {

template <class C, int DIR, class IO>
void run_col_pass(const IO& io_in, long long ncols, const cplx<typename C::T>* tw) {
  using F = ColFFT<C, DIR, IO>;
  using cx = cplx<typename C::T>;
  std::vector<cx> lds((size_t)(C::LDS_BYTES + IO::LDS_EXTRA) / sizeof(cx));   // exactly the kernel's dynamic LDS
  std::vector<typename F::Regs> regs(C::NT);
  std::vector<IO> ios(C::NT, io_in);             // every "thread" has its own copy of the kernel argument
  for (auto& io : ios) io.bind_seed();
  const long long ntiles = ncols / C::TC;
  for (long long t0 = 0; t0 < ntiles; ++t0) {
    const long long tile = io_in.remap_tile(t0);
    const cx* ltw = tw;
    if (F::HAS_PROLOGUE) {
      for (int t = 0; t < C::NT; ++t) F::prologue(t, ios[t], tw, lds.data());
      if (C::NPASS >= 2) ltw = F::lds_tw(lds.data());
    }
    for (int t = 0; t < C::NT; ++t) F::pass_first(t, tile, ios[t], lds.data());
    if (C::NPASS == 3) {
      for (int t = 0; t < C::NT; ++t) F::pass_mid_read(t, ltw, lds.data(), regs[t]);
      for (int t = 0; t < C::NT; ++t) F::pass_mid_write(t, lds.data(), regs[t]);
    }
    if (C::NPASS >= 2)
      for (int t = 0; t < C::NT; ++t) F::pass_last(t, tile, ios[t], ltw, lds.data());
  }
}

// a pass of length 2 C1::N through Col2 (col2_kernel's phases, a loop boundary for each of its barriers); tw2 = the 2 C1::N-point table
template <class C1, int DIR, class IO>
void run_col2_pass(const IO& io_in, long long ncols, const cplx<typename C1::T>* tw2) {
  using X = Col2<C1, DIR, IO>;
  using F = typename X::F;
  using cx = cplx<typename C1::T>;
  std::vector<cx> lds((size_t)(C1::LDS_BYTES + IO::LDS_EXTRA) / sizeof(cx));
  std::vector<typename F::Regs> regs(C1::NT);
  std::vector<typename F::TwRegs> twr(C1::NT);
  std::vector<typename X::Park> pk(C1::NT);
  std::vector<IO> ios(C1::NT, io_in);
  for (auto& io : ios) io.bind_seed();
  const long long ntiles = ncols / C1::TC;
  for (long long t0 = 0; t0 < ntiles; ++t0) {
    const long long tile = io_in.remap_tile(t0);
    if (IO::LDS_EXTRA > 0) for (int t = 0; t < C1::NT; ++t) ios[t].prologue(t, C1::NT, F::lds_io(lds.data()));
    for (int t = 0; t < C1::NT; ++t) X::tw_fetch(t, tw2, twr[t]);
    const cx* ltw = F::lds_tw(lds.data());
    for (int phase = 0; phase < 2; ++phase) {
      for (int t = 0; t < C1::NT; ++t) ios[t].set_phase(phase);
      for (int t = 0; t < C1::NT; ++t) F::pass_first(t, tile, ios[t], lds.data());
      if (phase == 0) for (int t = 0; t < C1::NT; ++t) F::tw_stage(t, lds.data(), twr[t]);
      if (C1::NPASS == 3) {
        for (int t = 0; t < C1::NT; ++t) F::pass_mid_read(t, ltw, lds.data(), regs[t]);
        for (int t = 0; t < C1::NT; ++t) F::pass_mid_write(t, lds.data(), regs[t]);
      }
      if (phase == 0) for (int t = 0; t < C1::NT; ++t) X::last_park(t, ltw, lds.data(), pk[t]);
      else for (int t = 0; t < C1::NT; ++t) X::last_combine(t, tile, ios[t], ltw, tw2, lds.data(), pk[t]);
    }
  }
}
// tile pairs with whole-line stores (rf_kernels.h colpair_body, rf_fft_col.h ColPair): phase 0 transforms tile 2p and PARKS the last
// pass's outputs, phase 1 transforms tile 2p + 1 (built without the kz = 0 repair: FIXOK = false) and stores both tiles row by row
template <class C, int DIR, class IO>
void run_colpair_pass(const IO& io_in, long long ncols, const cplx<typename C::T>* tw) {
  using X = ColPair<C, DIR, IO>;
  using F = typename X::F;
  using cx = cplx<typename C::T>;
  std::vector<cx> lds((size_t)(C::LDS_BYTES + IO::LDS_EXTRA) / sizeof(cx));
  std::vector<typename F::Regs> regs(C::NT);
  std::vector<typename F::TwRegs> twr(C::NT);
  std::vector<typename F::PreRegs> pre(C::NT);
  std::vector<typename X::Park> pk(C::NT);
  std::vector<IO> ios(C::NT, io_in);
  for (auto& io : ios) io.bind_seed();
  const long long npairs = ncols / (2 * C::TC);
  for (long long pair = 0; pair < npairs; ++pair) {
    if (IO::LDS_EXTRA > 0) for (int t = 0; t < C::NT; ++t) ios[t].prologue(t, C::NT, F::lds_io(lds.data()));
    for (int t = 0; t < C::NT; ++t) F::tw_fetch(t, tw, twr[t]);
    const cx* ltw = F::lds_tw(lds.data());
    for (int phase = 0; phase < 2; ++phase) {
      const long long tile = 2 * pair + phase;
      if (F::PRELOAD) for (int t = 0; t < C::NT; ++t) F::preload(t, tile, ios[t], pre[t]);
      for (int t = 0; t < C::NT; ++t) {
        if (phase == 0) F::template pass_first<true>(t, tile, ios[t], lds.data(), pre[t], F::PRELOAD);
        else F::template pass_first<false>(t, tile, ios[t], lds.data(), pre[t], F::PRELOAD);
      }
      if (phase == 0) for (int t = 0; t < C::NT; ++t) F::tw_stage(t, lds.data(), twr[t]);
      if (C::NPASS == 3) {
        for (int t = 0; t < C::NT; ++t) F::pass_mid_read(t, ltw, lds.data(), regs[t]);
        for (int t = 0; t < C::NT; ++t) F::pass_mid_write(t, lds.data(), regs[t]);
      }
      if (phase == 0) for (int t = 0; t < C::NT; ++t) X::last_park(t, ltw, lds.data(), pk[t]);
      else for (int t = 0; t < C::NT; ++t) X::last_store(t, 2 * pair, ios[t], ltw, lds.data(), pk[t]);
    }
  }
}
int g_pairs = 1;           // (the product's rule: the float32 generation pass of length 1024 runs on tile pairs)

// the product's rule (rf_k_col_plain.hip / rf_k_col_gen.hip): float32 in-place passes of length 2048 -- and, since round 4, of length
// 1024 -- run through Col2 (two half-length transforms per tile)
template <typename T, int DIR>
bool pair_pass_2048(int N, cplx<T>* base, ColGeom g, long long ncols) {
  if ((N != 2048 && N != 1024) || sizeof(T) != 4 || ncols % 8) return false;
  if constexpr (sizeof(T) == 4) {
    auto tw2 = make_twiddles<float>(N);
    Pair2ColIO<float> io; io.base = base; io.g = g; io.gin = g; io.gin.row_stride = 2 * g.row_stride; io.par_off = g.row_stride;
    if (N == 2048) run_col2_pass<GenSel<float, 1024>::type, DIR, Pair2ColIO<float>>(io, ncols, tw2.data());
    else run_col2_pass<PairSel1024::type, DIR, Pair2ColIO<float>>(io, ncols, tw2.data());
  }
  return true;
}

template <typename T, int DIR, class IO, template <typename, int> class SEL = ColSel>
int dispatch_col(int N, const IO& io, long long ncols) {
  auto tw = make_twiddles<T>(N);
  switch (N) {
#define X(NN)                                                                        \
  case NN:                                                                           \
    if (ncols % SEL<T, NN>::type::TC) return -2;                                     \
    run_col_pass<typename SEL<T, NN>::type, DIR, IO>(io, ncols, tw.data());          \
    return 0;
    RF_COL_SIZES(X)
#undef X
    default: return -1;
  }
}

// The product's hand-off of the x pass's output through the blocked intermediate X [x block][kz tile][ny][rb][tc] (rf_capi.hip
// queue_xyz / queue_yz): y pass in place on X, z pass gathering X -> W.  g_xposed mirrors RF_FLAG_TRANSPOSED_INTERMEDIATE; the
// rule for when it applies is the product's xpose_ok().
int g_xposed = 0;          // (the product's default: RF_FLAG_TRANSPOSED_INTERMEDIATE is off)
int g_rowblock = 64;
template <typename T> int tile_cols(int N, bool gen) {
  switch (N) {
#define X(NN) case NN: return gen ? GenSel<T, NN>::type::TC : ColSel<T, NN>::type::TC;
    RF_COL_SIZES(X)
#undef X
    default: return 0;
  }
}
template <typename T> int row_block(int N) {          // the product's col_gen_row_block()
  if (g_rowblock <= 0 || g_rowblock >= N || (g_rowblock & (g_rowblock - 1))) return N;
  switch (N) {
#define X(NN) case NN: { using C = typename GenSel<T, NN>::type; return (C::NPASS >= 2 && (NN / C::RL) % g_rowblock == 0) ? g_rowblock : N; }
    RF_COL_SIZES(X)
#undef X
    default: return N;
  }
}
template <typename T> bool xgather_ok(int M, int tc, int rb) {     // the product's row_c2r_xgather_ok()
  switch (M) {
#define X(MM) case MM: { using C = typename RowSel<T, MM>::type; return tc > 0 && rb > 0 && rb % C::NRT == 0 && M % tc == 0; }
    RF_ROW_SIZES(X)
#undef X
    default: return false;
  }
}
template <typename T> bool xpose_ok(int nx, int ny, long long nzl) {
  const int tcx = tile_cols<T>(nx, true), tcy = tile_cols<T>(ny, false);
  return g_xposed && tcx > 0 && tcx == tcy && nzl >= tcx && nzl % tcx == 0 && xgather_ok<T>((int)nzl, tcx, row_block<T>(nx));
}
template <typename T> ColGeom xposed_x_geom(int nx, int ny, long long nzl) {
  return xblock_x_geom(nx, ny, nzl, tile_cols<T>(nx, true), row_block<T>(nx));
}

template <class C, class IO>
void run_row_c2r(const IO& io_in, long long nrows, const cplx<typename C::T>* tw, double* s1, double* s2) {
  using F = RowC2R<C, IO>;
  IO io = io_in;
  using cx = cplx<typename C::T>;
  std::vector<cx> lds((size_t)C::LDS_BYTES / sizeof(cx));
  std::vector<typename F::Regs> regs(C::NT);
  const long long ntiles = (nrows + C::NRT - 1) / C::NRT;
  double a1 = 0, a2 = 0;
  for (long long tile = 0; tile < ntiles; ++tile) {
    for (int t = 0; t < C::NT; ++t) F::prologue(t, tw, lds.data());
    if constexpr (row_io_wants_stage<IO>::value)           // (as row_c2r_kernel: the IO's table into the tile's spare slots, once per tile)
      for (int t = 0; t < C::NT; ++t) io.template stage<C>(t, lds.data());
    const cx* ltw = F::lds_tw(lds.data());
    for (int t = 0; t < C::NT; ++t) F::pass_first(t, tile, nrows, io, ltw, lds.data(), regs[t]);
    if (C::NPASS == 3) {
      for (int t = 0; t < C::NT; ++t) F::pass_mid_read(t, ltw, lds.data(), regs[t]);
      for (int t = 0; t < C::NT; ++t) F::pass_mid_write(t, lds.data(), regs[t]);
    }
    if (C::NPASS >= 2)
      for (int t = 0; t < C::NT; ++t) F::pass_last(t, tile, nrows, io, ltw, lds.data(), regs[t]);
    for (int t = 0; t < C::NT; ++t) { a1 += regs[t].mom.sum(); a2 += regs[t].mom.sumsq(); }
  }
  *s1 = a1; *s2 = a2;
}

template <typename T>
int dispatch_row_c2r(int M, cplx<T>* base, long long nrows, double scale, double* s1, double* s2) {
  auto tw = make_twiddles<T>(2 * M);
  PlainRowIO<T> io; io.base = base; io.scale = (T)scale; io.M_of = M;
  switch (M) {
#define X(MM) case MM: run_row_c2r<typename RowSel<T, MM>::type, PlainRowIO<T>>(io, nrows, tw.data(), s1, s2); return 0;
    RF_ROW_SIZES(X)
#undef X
    default: return -1;
  }
}

// y pass in place on X, then the gathering z pass X -> W (+ moments)
template <typename T>
int xposed_yz(int nx, int ny, int nz, cplx<T>* X, cplx<T>* W, double* s1, double* s2) {
  const long long nzc = nz / 2, tc = tile_cols<T>(ny, false), rb = row_block<T>(nx);
  XposeColIO<T> io;
  io.src = X; io.gs = xblock_y_geom(nx, ny, nzc, tc, rb);
  io.base = X; io.g = io.gs;
  set_xpose_order(io, nx, nzc / tc);
  if (int rc = dispatch_col<T, +1>(ny, io, (long long)nx * nzc)) return rc;
  auto tw = make_twiddles<T>(2 * (int)nzc);
  XGatherRowIO<T> zio;
  zio.src = X; zio.dst = W; zio.scale = (T)(1.0 / ((double)nx * ny * nz)); zio.M_of = (int)nzc;
  zio.seg_shift = ilog2ll(tc); zio.rb_shift = ilog2ll(rb); zio.ny_shift = ilog2ll(ny);
  zio.kt_stride = (long long)ny * rb * tc; zio.xb_stride = (nzc / tc) * zio.kt_stride;
  switch ((int)nzc) {
#define X(MM) case MM: run_row_c2r<typename RowSel<T, MM>::type, XGatherRowIO<T>>(zio, (long long)nx * ny, tw.data(), s1, s2); return 0;
    RF_ROW_SIZES(X)
#undef X
    default: return -1;
  }
}

template <class C>
void run_row_r2c(const PlainRowFwdIO<typename C::T>& io, long long nrows, const cplx<typename C::T>* tw) {
  using F = RowR2C<C, PlainRowFwdIO<typename C::T>>;
  using cx = cplx<typename C::T>;
  std::vector<cx> lds((size_t)C::LDS_BYTES / sizeof(cx));
  std::vector<typename F::Regs> regs(C::NT);
  const long long ntiles = (nrows + C::NRT - 1) / C::NRT;
  for (long long tile = 0; tile < ntiles; ++tile) {
    for (int t = 0; t < C::NT; ++t) F::prologue(t, tw, lds.data());
    const cx* ltw = F::lds_tw(lds.data());
    if (C::NPASS >= 2) for (int t = 0; t < C::NT; ++t) F::pass_first(t, tile, nrows, io, lds.data());
    if (C::NPASS == 3) {
      for (int t = 0; t < C::NT; ++t) F::pass_mid_read(t, ltw, lds.data(), regs[t]);
      for (int t = 0; t < C::NT; ++t) F::pass_mid_write(t, lds.data(), regs[t]);
    }
    // every thread reads all its LDS inputs before any thread stores (stores go to global memory only)
    for (int t = 0; t < C::NT; ++t) F::pass_last(t, tile, nrows, io, ltw, lds.data());
  }
}

template <typename T>
int dispatch_row_r2c(int M, cplx<T>* base, long long nrows) {
  auto tw = make_twiddles<T>(2 * M);
  PlainRowFwdIO<T> io; io.base = base; io.M_of = M;
  switch (M) {
#define X(MM) case MM: run_row_r2c<typename RowSel<T, MM>::type>(io, nrows, tw.data()); return 0;
    RF_ROW_SIZES(X)
#undef X
    default: return -1;
  }
}

template <typename T>
int r2c_impl(int nx, int ny, int nz, const T* field, cplx<T>* K) {
  const long long nzc = nz / 2;
  std::vector<cplx<T>> W((size_t)nx * ny * nzc);
  memcpy(W.data(), field, W.size() * sizeof(cplx<T>));
  int rc = dispatch_row_r2c<T>((int)nzc, W.data(), (long long)nx * ny);
  if (rc) return rc;
  PlainColIO<T> yio; yio.base = W.data(); yio.g = ColGeom{nzc, (long long)ny * nzc, nzc};
  rc = dispatch_col<T, -1>(ny, yio, (long long)nx * nzc);
  if (rc) return rc;
  PlainColIO<T> xio; xio.base = W.data(); xio.g = ColGeom{(long long)ny * nzc, 0, (long long)ny * nzc};
  rc = dispatch_col<T, -1>(nx, xio, (long long)ny * nzc);
  if (rc) return rc;
  const int nzh = (int)nzc + 1;
  for (int ix = 0; ix < nx; ++ix)
    for (int iy = 0; iy < ny; ++iy) {
      const long long col = (long long)ix * ny + iy, mcol = (long long)((nx - ix) % nx) * ny + (ny - iy) % ny;
      for (int iz = 1; iz < nzc; ++iz) K[col * nzh + iz] = W[col * nzc + iz];
      const cplx<T> a = W[col * nzc], b = W[mcol * nzc];
      K[col * nzh] = mk<T>((T)0.5 * (a.x + b.x), (T)0.5 * (a.y - b.y));
      K[col * nzh + nzc] = mk<T>((T)0.5 * (a.y + b.y), (T)0.5 * (b.x - a.x));
    }
  return 0;
}

template <class C, int DIR>
void run_row_c2c(const ScaledRowIO<typename C::T>& io, long long nrows, const cplx<typename C::T>* tw) {
  using F = RowC2C<C, DIR, ScaledRowIO<typename C::T>>;
  using cx = cplx<typename C::T>;
  std::vector<cx> lds((size_t)C::LDS_BYTES / sizeof(cx) + 1);
  std::vector<typename F::Regs> regs(C::NT);
  const long long ntiles = (nrows + C::NRT - 1) / C::NRT;
  for (long long tile = 0; tile < ntiles; ++tile) {
    for (int t = 0; t < C::NT; ++t) F::prologue(t, tw, lds.data());
    const cx* ltw = F::lds_tw(lds.data());
    for (int t = 0; t < C::NT; ++t) F::pass_first(t, tile, nrows, io, lds.data());
    if (C::NPASS == 3) {
      for (int t = 0; t < C::NT; ++t) F::pass_mid_read(t, ltw, lds.data(), regs[t]);
      for (int t = 0; t < C::NT; ++t) F::pass_mid_write(t, lds.data(), regs[t]);
    }
    if (C::NPASS >= 2)
      for (int t = 0; t < C::NT; ++t) F::pass_last(t, tile, nrows, io, ltw, lds.data());
  }
}

template <typename T, int DIR>
int dispatch_row_c2c(int M, cplx<T>* base, long long nrows, double scale) {
  auto tw = make_twiddles<T>(M);
  ScaledRowIO<T> io; io.base = base; io.M_of = M; io.scale = (T)scale;
  switch (M) {
#define X(MM) case MM: run_row_c2c<typename RowSel<T, MM>::type, DIR>(io, nrows, tw.data()); return 0;
    RF_ROW_SIZES(X)
#undef X
    default: return -1;
  }
}

// unpacked c2c in place on data[nx][ny][nz] (same pass order as rf_execute_c2c)
template <typename T, int DIR>
int c2c_impl(int nx, int ny, int nz, cplx<T>* W) {
  PlainColIO<T> xio; xio.base = W; xio.g = ColGeom{(long long)ny * nz, 0, (long long)ny * nz};
  int rc = dispatch_col<T, DIR>(nx, xio, (long long)ny * nz);
  if (rc) return rc;
  PlainColIO<T> yio; yio.base = W; yio.g = ColGeom{nz, (long long)ny * nz, nz};
  rc = dispatch_col<T, DIR>(ny, yio, (long long)nx * nz);
  if (rc) return rc;
  return dispatch_row_c2c<T, DIR>(nz, W, (long long)nx * ny, DIR > 0 ? 1.0 / ((double)nx * ny * nz) : 1.0);
}

struct GenHost {
  SigmaTableHost tab;
  GenParams gp;
};

void fill_gen(GenHost& h, int nx, int ny, int nz, const double* kx2, const double* ky2, const double* kz2,
              const double* log10k, const double* sigma, int nt, int noise_mode, uint64_t seed, const double* noise) {
  build_sigma_table(log10k, sigma, nt, h.tab);
  GenParams& g = h.gp;
  g.nx = nx; g.ny = ny; g.nz = nz; g.kx2 = kx2; g.ky2 = ky2; g.kz2 = kz2;
  g.xt = h.tab.xt.data(); g.st = h.tab.st.data(); g.sl = h.tab.sl.data(); g.bin = h.tab.bin.data();
  g.nt = nt; g.nbins = (int)h.tab.bin.size(); g.x0 = h.tab.x0; g.inv_dx = h.tab.inv_dx;
  g.noise_mode = noise_mode; g.seed = seed; g.seed_dev = nullptr; g.noise = noise;
  g.zpitch = nz / 2 + 1; g.zoff = 0;
}

template <typename T>
int c2r_impl(int nx, int ny, int nz, const GenHost* gen, const cplx<T>* kspace, cplx<T>* W, double* s1, double* s2) {
  const long long nzc = nz / 2;
  // x pass (generation or API-layout k-space fused into the load)
  GenColIO<T> gio;
  gio.base = W; gio.g = ColGeom{(long long)ny * nzc, 0, (long long)ny * nzc};
  if (gen) gio.gp = gen->gp; else { memset(&gio.gp, 0, sizeof(gio.gp)); gio.gp.nx = nx; gio.gp.ny = ny; gio.gp.nz = nz; gio.gp.zpitch = nz / 2 + 1; }
  gio.kspace = kspace; gio.kz0 = 0; gio.nzl = (int)nzc;
  int rc;
  if (xpose_ok<T>(nx, ny, nzc)) {                  // x pass -> transposed intermediate, y pass X -> W
    std::vector<cplx<T>> X((size_t)nx * ny * nzc);
    gio.base = X.data(); gio.g = xposed_x_geom<T>(nx, ny, nzc);
    rc = dispatch_col<T, +1, GenColIO<T>, GenSel>(nx, gio, (long long)ny * nzc);
    if (rc) return rc;
    return xposed_yz<T>(nx, ny, nz, X.data(), W, s1, s2);
  }
  rc = dispatch_col<T, +1, GenColIO<T>, GenSel>(nx, gio, (long long)ny * nzc);
  if (rc) return rc;
  // y pass, in place
  PlainColIO<T> pio; pio.base = W; pio.g = ColGeom{nzc, (long long)ny * nzc, nzc};
  rc = dispatch_col<T, +1>(ny, pio, (long long)nx * nzc);
  if (rc) return rc;
  // z pass c2r, in place
  return dispatch_row_c2r<T>((int)nzc, W, (long long)nx * ny, 1.0 / ((double)nx * ny * nz), s1, s2);
}

// fix_fill_kernel (rf_kernels.h): the repaired kz = 0 slot of every mode (ix, iy) into the side buffer the FIX = 3 launch reads
template <class IOF, class CT>
void fill_fix_buffer(IOF iof, std::vector<CT>& buf) {
  const int nx = iof.gp.nx, ny = iof.gp.ny;
  buf.resize((size_t)nx * ny);
  std::vector<char> lds(IOF::LDS_EXTRA + 16);
  iof.bind_seed();
  iof.prologue(0, 1, lds.data());
  for (int iy = 0; iy < ny; ++iy)
    for (int ix = 0; ix < nx; ++ix) buf[(size_t)iy * nx + ix] = iof.fix_value((long long)iy * iof.nzl, ix, 0);
}

template <typename T, class IO>
int realise_fast_impl(int nx, int ny, int nz, const GenHost& h, uint64_t seed, double xlo, double xhi, double dkx,
                      cplx<T>* W, double* s1, double* s2) {
  const long long nzc = nz / 2;
  std::vector<FastRec> rec;
  IO io;
  io.base = W; io.g = ColGeom{(long long)ny * nzc, 0, (long long)ny * nzc}; io.kz0 = 0; io.nzl = (int)nzc; io.rec = nullptr;
  FastGenParams& f = io.gp;
  double x0, dx;
  if (!build_fast_records(h.tab, xlo, xhi, rec, x0, dx)) return -3;
  f.nx = nx; f.ny = ny; f.nz = nz; f.dkx = (float)dkx; f.dky = (float)std::sqrt(h.gp.ky2[1]); f.dkz = (float)std::sqrt(h.gp.kz2[1]);
  f.rec = rec.data(); f.nbins = (int)rec.size();
  f.u_scale = (float)(0.5 * std::log10(2.0) / dx); f.u_off = (float)(-x0 / dx);
  f.seed = seed; f.seed_dev = nullptr; f.noise = nullptr; f.noise32 = nullptr; f.rowtab = nullptr; f.seg_cap = 0; f.zpitch = nz / 2 + 1; f.zoff = 0; f.ppitch = nz / 2 + 2;
  int rc;
  if (xpose_ok<T>(nx, ny, nzc)) {
    std::vector<cplx<T>> X((size_t)nx * ny * nzc);
    io.base = X.data(); io.g = xposed_x_geom<T>(nx, ny, nzc);
    rc = dispatch_col<T, +1, IO, GenSel>(nx, io, (long long)ny * nzc);
    if (rc) return rc;
    return xposed_yz<T>(nx, ny, nz, X.data(), W, s1, s2);
  }
  bool x_done = false;
  // (where the library splits the pass -- kz runs of more than one whole tile -- its repair launch is the FIX = 3 kernel behind
  // fix_fill_kernel; the emulator runs that kernel over every tile: tiles without kz = 0 take nothing from the side buffer, as
  // the FIX = 0 launch does)
  std::vector<cplx<T>> fixbuf;
  if constexpr (sizeof(T) == 4) {
    if (nx == 2048 && ((long long)ny * nzc) % 8 == 0) {      // the library's x pass at this length: Col2 over the 1024-point configuration
      using C1 = GenSel<float, 1024>::type;
      using IO2 = FastGenColIOT<3, 0, 0, 0, 2>;
      IO2 io2;
      io2.base = W; io2.g = io.g; io2.kz0 = 0; io2.nzl = (int)nzc; io2.rec = nullptr; io2.gp = io.gp; io2.pot = nullptr;
      typename IO2::fill_io iof;
      iof.base = W; iof.g = io.g; iof.kz0 = 0; iof.nzl = (int)nzc; iof.rec = nullptr; iof.gp = io.gp; iof.pot = nullptr;
      fill_fix_buffer(iof, fixbuf);
      io2.fixbuf = fixbuf.data();
      auto tw2 = make_twiddles<float>(2048);
      run_col2_pass<C1, +1, IO2>(io2, (long long)ny * nzc, tw2.data());
      x_done = true;
    }
  }
  if constexpr (sizeof(T) == 8) {
    if (nx == 1024 && ((long long)ny * nzc) % 8 == 0) {      // float64, length 1024: Col2 over the 512-point configuration
      using C1 = GenSel<double, 512>::type;
      using IO2 = FastGenColIO64<3, 0, 0, 2>;
      IO2 io2;
      io2.base = W; io2.g = io.g; io2.kz0 = 0; io2.nzl = (int)nzc; io2.rec = nullptr; io2.gp = io.gp; io2.pot = nullptr;
      typename IO2::fill_io iof;
      iof.base = W; iof.g = io.g; iof.kz0 = 0; iof.nzl = (int)nzc; iof.rec = nullptr; iof.gp = io.gp; iof.pot = nullptr;
      fill_fix_buffer(iof, fixbuf);
      io2.fixbuf = fixbuf.data();
      auto tw2 = make_twiddles<double>(1024);
      run_col2_pass<C1, +1, IO2>(io2, (long long)ny * nzc, tw2.data());
      x_done = true;
    }
  }
  if constexpr (sizeof(T) == 4) {
    using C = GenSel<float, 1024>::type;
    if (!x_done && g_pairs && nx == 1024 && nzc > C::TC && nzc % (2 * C::TC) == 0) {      // float32, length 1024: tile pairs (ColPair), repair from the side buffer
      using IOC = typename IO::template with_fix<3>;
      IOC ioc;
      ioc.base = W; ioc.g = io.g; ioc.kz0 = 0; ioc.nzl = (int)nzc; ioc.rec = nullptr; ioc.gp = io.gp; ioc.pot = nullptr;
      typename IOC::fill_io iof;
      iof.base = W; iof.g = io.g; iof.kz0 = 0; iof.nzl = (int)nzc; iof.rec = nullptr; iof.gp = io.gp; iof.pot = nullptr;
      fill_fix_buffer(iof, fixbuf);
      ioc.fixbuf = fixbuf.data();
      auto tw = make_twiddles<float>(1024);
      run_colpair_pass<C, +1, IOC>(ioc, (long long)ny * nzc, tw.data());
      x_done = true;
    }
  }
  if (!x_done && (nx == 512 || nx == 1024)) {               // the long whole-column passes: FIX = 3 where the library splits
    using IOC = typename IO::template with_fix<3>;
    const int tc = tile_cols<T>(nx, true);
    if (nzc > tc && nzc % tc == 0) {
      IOC ioc;
      ioc.base = W; ioc.g = io.g; ioc.kz0 = 0; ioc.nzl = (int)nzc; ioc.rec = nullptr; ioc.gp = io.gp; ioc.pot = nullptr;
      typename IOC::fill_io iof;
      iof.base = W; iof.g = io.g; iof.kz0 = 0; iof.nzl = (int)nzc; iof.rec = nullptr; iof.gp = io.gp; iof.pot = nullptr;
      fill_fix_buffer(iof, fixbuf);
      ioc.fixbuf = fixbuf.data();
      rc = dispatch_col<T, +1, IOC, GenSel>(nx, ioc, (long long)ny * nzc);
      if (rc) return rc;
      x_done = true;
    }
  }
  if (!x_done) {
    rc = dispatch_col<T, +1, IO, GenSel>(nx, io, (long long)ny * nzc);
    if (rc) return rc;
  }
  const ColGeom gy{nzc, (long long)ny * nzc, nzc};
  if (!pair_pass_2048<T, +1>(ny, W, gy, (long long)nx * nzc)) {
    PlainColIO<T> pio; pio.base = W; pio.g = gy;
    rc = dispatch_col<T, +1>(ny, pio, (long long)nx * nzc);
    if (rc) return rc;
  }
  return dispatch_row_c2r<T>((int)nzc, W, (long long)nx * ny, 1.0 / ((double)nx * ny * nz), s1, s2);
}


// ---- non-power-of-two path (rf_generic.h): the same block functions the kernels run, one "thread" per block ----
struct NoSync { void operator()() const {} };

// The library's sequences (rf_generic.h generic_*_seq) with every launch replaced by a loop over its blocks, one "thread" each.
// g_generic_cap: the longest line kept "in LDS" -- the library's cap is 8192 / 4096; tests lower it so that small grids take the
// four-step form of the long axes.
int g_generic_cap = 0;                  // 0: generic_max_axis(dtype)
int g_generic_tile = 3;                 // lines per block of the strided passes: 3 = deliberately not dividing the line counts (the walk by
                                        // index); 1 = the kernels' walk, a thread staying on one line (GenericWalk::fixed)
template <typename T> struct EmuGenericOps {
  const cplx<T>*rx, *ry, *rz;
  int nx, ny, nz, M;                    // M = row length of the contiguous complex transform's root table / 2 (packed) -- see callers
  long long rows;
  std::vector<cplx<T>> lds;
  double s1 = 0, s2 = 0;
  const cplx<T>* root(int which) const { return which == 0 ? rx : (which == 1 ? ry : rz); }
  int axis(const void* src, void* dst, const GenericAxis& ax, long long stride, long long inner, long long outer, long long nlines, int which, int sign, double scale) {
    const int TC = g_generic_tile;
    lds.resize(2 * (size_t)ax.n * TC + 2 * ax.n + 4);
    for (long long b = 0; b * TC < nlines; ++b)
      generic_axis_block<T>((const cplx<T>*)src, (cplx<T>*)dst, ax, stride, inner, outer, nlines, TC, root(which), sign, (T)scale, lds.data(), b, 0, 1, NoSync(), 1);
    return 0;
  }
  int lines(const void* src, void* dst, const GenericLines& L, int which) {
    const int TC = g_generic_tile;
    lds.resize(2 * (size_t)L.ax.n * TC + 2 * L.ax.n + 4);
    for (long long b = 0; b * TC < L.nlines(); ++b)
      generic_lines_block<T>((const cplx<T>*)src, (cplx<T>*)dst, L, TC, root(which), lds.data(), b, 0, 1, NoSync(), 1);
    return 0;
  }
  GenericAxis az;
  int row_c2r(const void* G, void* W, double scale) {
    const int TR = 2;
    lds.resize(2 * (size_t)az.n * generic_row_pitch(TR) + 2 * az.n + 4);
    for (long long b = 0; b * TR < rows; ++b)
      generic_row_c2r_block<T>((const cplx<T>*)G, (T*)W, az, rows, TR, rz, (T)scale, lds.data(), b, 0, 1, NoSync(), s1, s2, 1);
    return 0;
  }
  int row_r2c(const void* W, void* G) {
    const int TR = 2;
    lds.resize(2 * (size_t)az.n * generic_row_pitch(TR) + 2 * az.n + 4);
    for (long long b = 0; b * TR < rows; ++b) generic_row_r2c_block<T>((const T*)W, (cplx<T>*)G, az, rows, TR, rz, lds.data(), b, 0, 1, NoSync(), 1);
    return 0;
  }
  int untangle(const void* G, void* Z) { for (long long i = 0; i < rows * M; ++i) generic_untangle_at<T>((const cplx<T>*)G, (cplx<T>*)Z, M, rz, i); return 0; }
  int tangle(const void* Z, void* G) { for (long long i = 0; i < rows * (M + 1); ++i) generic_tangle_at<T>((const cplx<T>*)Z, (cplx<T>*)G, M, rz, i); return 0; }
  int moments(const void* W) {
    const T* w = (const T*)W;
    for (long long i = 0; i < rows * 2 * M; ++i) { s1 += (double)w[i]; s2 += (double)w[i] * (double)w[i]; }
    return 0;
  }
  int copy(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
};
// one axis: short (its line fits the cap) or split in two factors that do; false: neither
inline bool emu_axis(long long n, int cap, GenericAxis& ax, GenericLong& lg) {
  lg = GenericLong();
  if (n <= cap) return generic_factor((int)n, ax);
  return generic_split(n, cap, lg);
}
template <typename T> bool emu_dims(int nx, int ny, int nz, bool packed, GenericDims& d) {
  const int cap = g_generic_cap > 0 ? g_generic_cap : generic_max_axis(sizeof(T) == 8);
  d.nx = nx; d.ny = ny; d.nz = nz; d.csize = sizeof(cplx<T>);
  if (packed && (nz & 1)) return false;
  return emu_axis(nx, cap, d.ax, d.lx) && emu_axis(ny, cap, d.ay, d.ly) && emu_axis(packed ? nz / 2 : nz, cap, d.az, d.lz);
}

template <typename T>
int generic_c2r_impl(int nx, int ny, int nz, const cplx<T>* K, T* W, double* s1, double* s2) {
  GenericDims d;
  if (!emu_dims<T>(nx, ny, nz, true, d)) return -1;
  const long long nzh = nz / 2 + 1;
  auto rx = make_twiddles<T>(nx), ry = make_twiddles<T>(ny), rz = make_twiddles<T>(nz);
  std::vector<cplx<T>> G((size_t)nx * ny * nzh), G2((size_t)nx * ny * nzh);
  EmuGenericOps<T> ops{rx.data(), ry.data(), rz.data(), nx, ny, nz, nz / 2, (long long)nx * ny};
  ops.az = d.az;
  const int rc = generic_c2r_seq(ops, d, K, G.data(), G2.data(), W, 1.0 / ((double)nx * ny * nz));
  if (s1) *s1 = ops.s1;
  if (s2) *s2 = ops.s2;
  return rc;
}

template <typename T>
int generic_r2c_impl(int nx, int ny, int nz, const T* W, cplx<T>* K) {
  GenericDims d;
  if (!emu_dims<T>(nx, ny, nz, true, d)) return -1;
  const long long nzh = nz / 2 + 1;
  auto rx = make_twiddles<T>(nx), ry = make_twiddles<T>(ny), rz = make_twiddles<T>(nz);
  std::vector<cplx<T>> G((size_t)nx * ny * nzh), G2((size_t)nx * ny * nzh);
  EmuGenericOps<T> ops{rx.data(), ry.data(), rz.data(), nx, ny, nz, nz / 2, (long long)nx * ny};
  ops.az = d.az;
  return generic_r2c_seq(ops, d, W, K, G.data(), G2.data());
}

template <typename T>
int generic_c2c_impl(int nx, int ny, int nz, int dir, cplx<T>* D) {
  GenericDims d;
  if (!emu_dims<T>(nx, ny, nz, false, d)) return -1;
  auto rx = make_twiddles<T>(nx), ry = make_twiddles<T>(ny), rz = make_twiddles<T>(nz);
  std::vector<cplx<T>> G((size_t)nx * ny * nz);
  EmuGenericOps<T> ops{rx.data(), ry.data(), rz.data(), nx, ny, nz, nz, (long long)nx * ny};
  ops.az = d.az;
  return generic_c2c_seq(ops, d, D, G.data(), dir, dir > 0 ? 1.0 / ((double)nx * ny * nz) : 1.0);
}

// rf_realise_lognormal on an uploaded k-space array: x pass, the accumulating y pass (AccColIO: one Parseval partial per tile),
// sigma and the tables as lognormal_tables_kernel forms them, the z pass with the map in its epilogue (LognormalRowIO)
template <typename T>
int c2r_lognormal_impl(int nx, int ny, int nz, const cplx<T>* kspace, const double* growth, const double* density, cplx<T>* W,
                       double* sigma_out, double* s1, double* s2) {
  const long long nzc = nz / 2;
  GenColIO<T> gio;
  gio.base = W; gio.g = ColGeom{(long long)ny * nzc, 0, (long long)ny * nzc};
  memset(&gio.gp, 0, sizeof(gio.gp)); gio.gp.nx = nx; gio.gp.ny = ny; gio.gp.nz = nz; gio.gp.zpitch = nz / 2 + 1;
  gio.kspace = kspace; gio.kz0 = 0; gio.nzl = (int)nzc;
  int rc = dispatch_col<T, +1, GenColIO<T>, GenSel>(nx, gio, (long long)ny * nzc);
  if (rc) return rc;
  // y pass with the accumulator: run tile by tile so that every "thread"'s acc is collected (the kernel's finish())
  const int tc = tile_cols<T>(ny, false);
  double S = 0;
  {
    auto tw = make_twiddles<T>(ny);
    const long long ncols = (long long)nx * nzc;
    if (ncols % tc) return -2;
    switch (ny) {
#define X(NN) case NN: { using C = typename ColSel<T, NN>::type; using IO = AccColIO<T>; using F = ColFFT<C, +1, IO>;                 \
      std::vector<cplx<T>> lds((size_t)(C::LDS_BYTES + IO::LDS_EXTRA) / sizeof(cplx<T>) + 1);                                       \
      std::vector<typename F::Regs> regs(C::NT);                                                                                    \
      for (long long tile = 0; tile < ncols / C::TC; ++tile) {                                                                      \
        IO io0; io0.base = W; io0.g = ColGeom{nzc, (long long)ny * nzc, nzc}; io0.partials = nullptr; io0.kz0 = 0; io0.nzl = (int)nzc; \
        std::vector<IO> ios(C::NT, io0);                                                                                            \
        const cplx<T>* ltw = tw.data();                                                                                             \
        if (F::HAS_PROLOGUE) { for (int t = 0; t < C::NT; ++t) F::prologue(t, ios[t], tw.data(), lds.data()); if (C::NPASS >= 2) ltw = F::lds_tw(lds.data()); } \
        for (int t = 0; t < C::NT; ++t) F::pass_first(t, tile, ios[t], lds.data());                                                 \
        if (C::NPASS == 3) { for (int t = 0; t < C::NT; ++t) F::pass_mid_read(t, ltw, lds.data(), regs[t]);                        \
                             for (int t = 0; t < C::NT; ++t) F::pass_mid_write(t, lds.data(), regs[t]); }                          \
        if (C::NPASS >= 2) for (int t = 0; t < C::NT; ++t) F::pass_last(t, tile, ios[t], ltw, lds.data());                          \
        double a = 0; for (int t = 0; t < C::NT; ++t) a += ios[t].weighted_sum();                                                              \
        S += a;                                                                                                                     \
      } break; }
      RF_COL_SIZES(X)
#undef X
      default: return -1;
    }
  }
  const double n3 = (double)nx * ny * nz;
  double sigma = std::sqrt(S / ((double)nx * ny * n3 * n3));
  if (sizeof(T) == 4) sigma = (double)(float)sigma;
  std::vector<double> Ap(nz), Bp(nz);
  for (int z = 0; z < nz; ++z) {
    const double g = sigma * growth[z], t = g * g + 1.0;
    Ap[z] = std::sqrt(std::log(t)) / sigma * lognormal_ap_unit<T>(1.0 / n3);
    Bp[z] = (density ? density[z] : 1.0) / std::sqrt(t);
  }
  *sigma_out = sigma;
  auto twz = make_twiddles<T>(2 * (int)nzc);
  switch ((int)nzc) {
#define X(MM) case MM: { constexpr int SP = (sizeof(T) == 8 && MM >= 512) ? 1 : 0;              /* as rf_k_row.hip launch_lognormal_t */  \
    LognormalRowIO<T, SP> zio; zio.base = W; zio.scale = (T)(1.0 / n3); zio.M_of = (int)nzc; zio.Ap = Ap.data(); zio.Bp = Bp.data();       \
    run_row_c2r<typename RowSel<T, MM>::type, LognormalRowIO<T, SP>>(zio, (long long)nx * ny, twz.data(), s1, s2); return 0; }
    RF_ROW_SIZES(X)
#undef X
    default: return -1;
  }
}
}  // namespace

extern "C" {

// The row table of resident float32 deviates (rf_core.h make_rowloc / row_pair; mt_rowtab_kernel builds it on the device): where
// the generation pass finds the pair of cell kz of the row whose first stream cell is rows[i].  first = exclusive scan of the
// per-segment counts, nseg + 1 entries.  out[i * nzh + kz] = slot index in the runs; returns the "row spans more than two
// segments" flag.
int emu_row_lookup(const unsigned long long* first, int nseg, unsigned long long cap, unsigned nzh, const unsigned long long* rows, int n,
                   unsigned long long* out) {
  FastGenParams g;
  memset(&g, 0, sizeof g);
  std::vector<cplx<float>> dummy(1);
  g.noise32 = dummy.data(); g.seg_cap = cap;
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const RowLoc e = make_rowloc(first, nseg, rows[i], nzh, &bad);
    for (unsigned kz = 0; kz < nzh; ++kz) out[(size_t)i * nzh + kz] = (unsigned long long)(row_pair(g, e, (int)kz) - g.noise32);
  }
  return bad;
}

// the generic (any even shape) transforms: API-layout half spectrum <-> dense real field; c2c in place
int emu_generic_c2r(int f64, int nx, int ny, int nz, const void* K, void* W, double* s1, double* s2) {
  return f64 ? generic_c2r_impl<double>(nx, ny, nz, (const cplx<double>*)K, (double*)W, s1, s2)
             : generic_c2r_impl<float>(nx, ny, nz, (const cplx<float>*)K, (float*)W, s1, s2);
}
int emu_generic_r2c(int f64, int nx, int ny, int nz, const void* W, void* K) {
  return f64 ? generic_r2c_impl<double>(nx, ny, nz, (const double*)W, (cplx<double>*)K)
             : generic_r2c_impl<float>(nx, ny, nz, (const float*)W, (cplx<float>*)K);
}
int emu_generic_c2c(int f64, int nx, int ny, int nz, int dir, void* D) {
  return f64 ? generic_c2c_impl<double>(nx, ny, nz, dir, (cplx<double>*)D) : generic_c2c_impl<float>(nx, ny, nz, dir, (cplx<float>*)D);
}

int emu_c2r_lognormal(int f64, int nx, int ny, int nz, const void* kspace, const double* growth, const double* density, void* W,
                      double* sigma_out, double* s1, double* s2) {
  return f64 ? c2r_lognormal_impl<double>(nx, ny, nz, (const cplx<double>*)kspace, growth, density, (cplx<double>*)W, sigma_out, s1, s2)
             : c2r_lognormal_impl<float>(nx, ny, nz, (const cplx<float>*)kspace, growth, density, (cplx<float>*)W, sigma_out, s1, s2);
}

// fused realisation with the fast native generation (float32 arithmetic; f64 != 0: float64 plan, values widened);
// [xlo, xhi] = log10 k range of the grid (padded)
int emu_realise_fast(int f64, int nx, int ny, int nz, const double* kx2, const double* ky2, const double* kz2,
                     const double* log10k, const double* sigma, int nt, uint64_t seed, double xlo, double xhi,
                     double dkx, void* W, double* s1, double* s2) {
  GenHost h;
  fill_gen(h, nx, ny, nz, kx2, ky2, kz2, log10k, sigma, nt, 0, seed, nullptr);
  if (f64) return realise_fast_impl<double, FastGenColIO64<1>>(nx, ny, nz, h, seed, xlo, xhi, dkx, (cplx<double>*)W, s1, s2);
  return realise_fast_impl<float, FastGenColIO>(nx, ny, nz, h, seed, xlo, xhi, dkx, (cplx<float>*)W, s1, s2);
}

// Row T alone: the fast float32 sigma(|k|^2) lookup (per-bin records, rf_core.h fast_sigma) next to the exact float64
// interpolation of the same table, for n values of |k|^2 -- bounds the table-lookup error separately from the
// transcendental error of the deviates.  [xlo, xhi] = padded log10 k range of the grid, as the library passes it.
int emu_fast_sigma(const double* log10k, const double* sigma, int nt, double xlo, double xhi, const float* k2, int n,
                   float* out_fast, double* out_exact) {
  SigmaTableHost tab;
  build_sigma_table(log10k, sigma, nt, tab);
  std::vector<FastRec> rec;
  double x0, dx;
  if (!build_fast_records(tab, xlo, xhi, rec, x0, dx)) return -3;
  FastGenParams f;
  memset(&f, 0, sizeof f);
  f.rec = rec.data(); f.nbins = (int)rec.size();
  f.u_scale = (float)(0.5 * std::log10(2.0) / dx); f.u_off = (float)(-x0 / dx);
  GenParams g;
  memset(&g, 0, sizeof g);
  g.xt = tab.xt.data(); g.st = tab.st.data(); g.sl = tab.sl.data(); g.bin = tab.bin.data();
  g.nt = nt; g.nbins = (int)tab.bin.size(); g.x0 = tab.x0; g.inv_dx = tab.inv_dx;
  for (int i = 0; i < n; ++i) {
    out_fast[i] = fast_sigma(f, rec.data(), k2[i]);
    out_exact[i] = sigma_lookup(g, 0.5 * std::log10((double)k2[i]));
  }
  return f.nbins;
}

// k-space after symmetrise in the API layout [nx][ny][nz/2+1] (rows K,T,R,S)
int emu_generate_kspace(int f64, int nx, int ny, int nz, const double* kx2, const double* ky2, const double* kz2,
                        const double* log10k, const double* sigma, int nt, int noise_mode, uint64_t seed,
                        const double* noise, void* out) {
  GenHost h;
  fill_gen(h, nx, ny, nz, kx2, ky2, kz2, log10k, sigma, nt, noise_mode, seed, noise);
  const int nzh = nz / 2 + 1;
  for (int ix = 0; ix < nx; ++ix)
    for (int iy = 0; iy < ny; ++iy)
      for (int iz = 0; iz < nzh; ++iz) {
        const size_t c = ((size_t)ix * ny + iy) * nzh + iz;
        if (f64) ((cplx<double>*)out)[c] = gen_cell<double>(h.gp, seed, ix, iy, iz);
        else     ((cplx<float>*)out)[c] = gen_cell<float>(h.gp, seed, ix, iy, iz);
      }
  return 0;
}

// the gathering z pass alone: X = blocked intermediate [nx / rb][M / tc][ny][rb][tc] -> W dense real rows [nx][ny][2 M]
int emu_row_c2r_xgather(int f64, int M, int nx, int ny, int tc, int rb, const void* X, void* W, double scale, double* s1, double* s2) {
  auto run = [&](auto tag) -> int {
    using T = decltype(tag);
    if (!xgather_ok<T>(M, tc, rb) || nx % rb) return -2;
    auto tw = make_twiddles<T>(2 * M);
    XGatherRowIO<T> zio;
    zio.src = (const cplx<T>*)X; zio.dst = (cplx<T>*)W; zio.scale = (T)scale; zio.M_of = M;
    zio.seg_shift = ilog2ll(tc); zio.rb_shift = ilog2ll(rb); zio.ny_shift = ilog2ll(ny);
    zio.kt_stride = (long long)ny * rb * tc; zio.xb_stride = (long long)(M / tc) * zio.kt_stride;
    switch (M) {
#define X(MM) case MM: run_row_c2r<typename RowSel<T, MM>::type, XGatherRowIO<T>>(zio, (long long)nx * ny, tw.data(), s1, s2); return 0;
      RF_ROW_SIZES(X)
#undef X
      default: return -1;
    }
  };
  return f64 ? run(double()) : run(float());
}

// 1 (default): c2r transforms hand x -> y through the transposed intermediate where the product would; 0: in place
int emu_set_xposed(int on) { const int old = g_xposed; g_xposed = on; return old; }
int emu_set_rowblock(int rb) { const int old = g_rowblock; g_rowblock = rb; return old; }
// the longest line the generic path keeps whole (0 = the library's cap, 8192 complex64 / 4096 complex128): longer axes take the four-step
// form -- lowered by tests so that small grids exercise it
int emu_set_generic_cap(int cap) { const int old = g_generic_cap; g_generic_cap = cap; return old; }
int emu_set_generic_tile(int tc) { const int old = g_generic_tile; g_generic_tile = tc > 0 ? tc : 3; return old; }
// 1 (default): the float32 generation pass of length 1024 on tile pairs (ColPair), as the product; 0: one tile per workgroup (ColFFT)
int emu_set_pairs(int on) { const int old = g_pairs; g_pairs = on; return old; }
// FastGenColIOT::share_row (rf_fft_gen.h): the butterfly row of slot jl when L butterflies are dealt to slots of S per wave so that
// the rows +-ix sit 32 lanes apart (the sigma exchange of the x pass); a host-side table for the bijection test
// (xs2_phase < 0: the whole-column form XS = 1; 0 / 1: the even / odd phase of the two-half-transform form XS = 2)
int emu_share_row(int jl, int L, int S, int xs2_phase) {
  if (xs2_phase < 0) { FastGenColIOT<0> io; return io.share_row(jl, L, S); }
  FastGenColIOT<0, 0, 0, 0, 2> io;
  io.set_phase(xs2_phase);
  return io.share_row(jl, L, S);
}
// generic_pos (rf_generic.h): where element e of a line of n goes in the LDS image of the in-place transform; out[e] = position,
// returns 1 when the axis is smooth (radices 2..5), 0 when the positions are the identity (two-buffer form), -1 when n does not factor
int emu_generic_positions(int n, int* out) {
  GenericAxis ax;
  if (!generic_factor(n, ax)) return -1;
  for (int e = 0; e < n; ++e) out[e] = generic_pos(ax, e);
  return generic_smooth(ax) ? 1 : 0;
}
// FastDiv (rf_generic.h): the first a < amax at which a / d or a % d by multiply-high differs from the machine's division; -1 if none
long long emu_fastdiv_first_error(unsigned d, unsigned amax) {
  const FastDiv fd(d, generic_magic(d));
  const FastDiv fd2(d);
  for (unsigned a = 0; a < amax; ++a) {
    uint32_t q, r;
    fd.divmod(a, q, r);
    if (q != a / d || r != a % d || fd2.div(a) != a / d) return (long long)a;
  }
  return -1;
}
int emu_xpose_applies(int f64, int nx, int ny, int nz) { return f64 ? xpose_ok<double>(nx, ny, nz / 2) : xpose_ok<float>(nx, ny, nz / 2); }

// full fused realisation: generation + x, y, z passes -> W (real [nx][ny][nz]) and (sum, sumsq)
int emu_realise(int f64, int nx, int ny, int nz, const double* kx2, const double* ky2, const double* kz2,
                const double* log10k, const double* sigma, int nt, int noise_mode, uint64_t seed,
                const double* noise, void* W, double* s1, double* s2) {
  GenHost h;
  fill_gen(h, nx, ny, nz, kx2, ky2, kz2, log10k, sigma, nt, noise_mode, seed, noise);
  return f64 ? c2r_impl<double>(nx, ny, nz, &h, nullptr, (cplx<double>*)W, s1, s2)
             : c2r_impl<float>(nx, ny, nz, &h, nullptr, (cplx<float>*)W, s1, s2);
}

// unfused c2r of an API-layout k-space array
int emu_c2r(int f64, int nx, int ny, int nz, const void* kspace, void* W, double* s1, double* s2) {
  return f64 ? c2r_impl<double>(nx, ny, nz, nullptr, (const cplx<double>*)kspace, (cplx<double>*)W, s1, s2)
             : c2r_impl<float>(nx, ny, nz, nullptr, (const cplx<float>*)kspace, (cplx<float>*)W, s1, s2);
}

// forward r2c of a dense real field [nx][ny][nz] into the API k layout [nx][ny][nz/2+1]
int emu_r2c(int f64, int nx, int ny, int nz, const void* field, void* K) {
  return f64 ? r2c_impl<double>(nx, ny, nz, (const double*)field, (cplx<double>*)K)
             : r2c_impl<float>(nx, ny, nz, (const float*)field, (cplx<float>*)K);
}

// unpacked complex-to-complex transform in place: dir = +1 inverse (1/N), -1 forward
int emu_c2c(int f64, int nx, int ny, int nz, int dir, void* data) {
  if (f64) return dir > 0 ? c2c_impl<double, +1>(nx, ny, nz, (cplx<double>*)data) : c2c_impl<double, -1>(nx, ny, nz, (cplx<double>*)data);
  return dir > 0 ? c2c_impl<float, +1>(nx, ny, nz, (cplx<float>*)data) : c2c_impl<float, -1>(nx, ny, nz, (cplx<float>*)data);
}

// one strided FFT pass over data[(C / inner) * outer_stride + C % inner + row * row_stride]
int emu_col_fft(int f64, int N, int dir, void* data, long long ncols, long long inner, long long outer_stride,
                long long row_stride) {
  ColGeom g{inner, outer_stride, row_stride};
  if (f64) {
    PlainColIO<double> io; io.base = (cplx<double>*)data; io.g = g;
    return dir > 0 ? dispatch_col<double, +1>(N, io, ncols) : dispatch_col<double, -1>(N, io, ncols);
  }
  if ((N == 2048 || N == 1024) && ncols % 8 == 0) {     // as the library does at these lengths: two half-length transforms per tile
    if (dir > 0) pair_pass_2048<float, +1>(N, (cplx<float>*)data, g, ncols); else pair_pass_2048<float, -1>(N, (cplx<float>*)data, g, ncols);
    return 0;
  }
  PlainColIO<float> io; io.base = (cplx<float>*)data; io.g = g;
  return dir > 0 ? dispatch_col<float, +1>(N, io, ncols) : dispatch_col<float, -1>(N, io, ncols);
}

}  // extern "C"
