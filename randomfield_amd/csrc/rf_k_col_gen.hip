// x pass of the c2r transform with generation (rows K,T,R,S) fused into its load: float32 kernels and the exact-chain kernels
// (the float64 fast-generation kernels live in rf_k_col_gen64.hip: see rf_col_gen_launch.h)
#include "rf_col_gen_launch.h"

namespace rf {
namespace {
template <typename T>
hipError_t launch_t(int N, cplx<T>* W, ColGeom g, long long ncols, const GenParams& gp, const cplx<T>* kspace,
                    int kz0, int nzl, const cplx<T>* tw, hipStream_t s, bool po) {
  GenColIO<T> io; io.base = W; io.g = g; io.gp = gp; io.kspace = kspace; io.kz0 = kz0; io.nzl = nzl;
  switch (N) {
#define X(NN)                                                                                                    \
  case NN: {                                                                                                     \
    using C = typename GenSel<T, NN>::type;                                                                      \
    if constexpr (NN == 2048 && sizeof(T) == 8) {               /* 64-bit lane offsets: float64, length 2048 */ \
      if (po || g.needs_wide(C::LMAX, C::TC, (int)sizeof(cplx<T>))) {                                            \
        GenColIO<T, true> iow; iow.base = W; iow.g = g; iow.gp = gp; iow.kspace = kspace; iow.kz0 = kz0; iow.nzl = nzl; \
        hipError_t e = launch_one<C, GenColIO<T, true>>(iow, ncols, tw, s, po);                                  \
        if (!po || e != hipSuccess) return e;                                                                    \
      }                                                                                                          \
    } else if (g.needs_wide(C::LMAX, C::TC, (int)sizeof(cplx<T>))) {                                             \
      return hipErrorInvalidValue;                                                                               \
    }                                                                                                            \
    return launch_one<C, GenColIO<T>>(io, ncols, tw, s, po);                                                     \
  }
    RF_COL_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t launch_col_fastgen(int f64, int N, void* W, ColGeom g, long long ncols, const FastGenParams& gp, int kz0, int nzl,
                              const void* tw, hipStream_t s, bool po, hipEvent_t after_repair, int x0, int x1, void* pot, void* fixbuf) {
  if (!col_fastgen_supported(f64, N)) return po ? hipSuccess : hipErrorInvalidValue;    // the caller keeps the exact kernel
  if (f64) return launch_col_fastgen64(N, W, g, ncols, gp, kz0, nzl, tw, s, po, after_repair, x0, x1, pot, fixbuf);
  const bool slab = x0 > 0 || x1 < N;          // replicated-generation mode: the SLAB instantiations guard their stores
  if (gp.emit_potential || po) {               // POT = 2: the pass transforms pscale * delta(k) / k^2 (rf_realise_scaled_potential)
    if (gp.emit_potential && (slab || pot || gp.noise)) return hipErrorInvalidValue;
    if (gp.noise32 || po)                      // ... of the replayed deviates (float32 pairs in the replay's runs)
      switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 2, 2>, FastGenColIOT<1, 0, 2, 2>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)fixbuf); if (!po || e != hipSuccess) return e; break; }
        RF_COL_SIZES(X)
#undef X
        default: return hipErrorInvalidValue;
      }
    if (RF_COL2_2048 && N == 2048) {
      using C1 = GenSel<float, 1024>::type;
      hipError_t e = launch_fast_one2<C1, FastGenColIOT<0, 0, 2, 0, 2>, FastGenColIOT<1, 0, 2, 0, 2>>(
          gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, (cplx<float>*)fixbuf);
      if (!po || e != hipSuccess) return e;
    }
    switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 2>, FastGenColIOT<1, 0, 2>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)fixbuf); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  if (gp.noise || po) {                        // resident float64 deviates (rng='reference') through the fast float32 sigma path
    if (gp.noise && (slab || pot)) return hipErrorInvalidValue;
    switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 0, 1>, FastGenColIOT<1, 0, 0, 1>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)fixbuf); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  if (gp.noise32 || po) {                      // resident float32 deviates, without / with the potential store
    if (gp.noise32 && slab) return hipErrorInvalidValue;
    // (the same pass as two 512-point transforms per tile, as the in-place y pass runs: measured 2.69 against 2.47 ms per 1024^3, not kept)
    if (!pot || po)
      switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 0, 2>, FastGenColIOT<1, 0, 0, 2>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)fixbuf); if (!po || e != hipSuccess) return e; break; }
        RF_COL_SIZES(X)
#undef X
        default: return hipErrorInvalidValue;
      }
    if (pot || po)
      switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 1, 2>, FastGenColIOT<1, 0, 1, 2>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)fixbuf, (cplx<float>*)pot); if (!po || e != hipSuccess) return e; break; }
        RF_COL_SIZES(X)
#undef X
        default: return hipErrorInvalidValue;
      }
  }
  if (pot || po) {                             // generation + potential store (save_potential=True), whole grid
    if (slab && pot) return hipErrorInvalidValue;
    switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 1>, FastGenColIOT<1, 0, 1>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)fixbuf, (cplx<float>*)pot); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  if (slab || po) {
    switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 1>, FastGenColIOT<1, 1>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)fixbuf); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  if (RF_COL2_2048 && N == 2048 && (!slab || po)) {
    using C1 = GenSel<float, 1024>::type;
    hipError_t e = launch_fast_one2<C1, FastGenColIOT<0, 0, 0, 0, 2>, FastGenColIOT<1, 0, 0, 0, 2>>(
        gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, (cplx<float>*)fixbuf);
    if (!po || e != hipSuccess) return e;
  }
  switch (N) {
#define X(NN) case NN: return launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0>, FastGenColIOT<1, 0>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)fixbuf);
    RF_COL_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

int col_gen_tile_cols(int f64, int N) {
  switch (N) {
#define X(NN) case NN: return f64 ? GenSel<double, NN>::type::TC : GenSel<float, NN>::type::TC;
    RF_COL_SIZES(X)
#undef X
    default: return 0;
  }
}

// rows of x per block of the transposed intermediate (rf_fft.h xpose_store_geom): `want` if the last pass's uniform row
// offsets (multiples of N / RL) are whole blocks, else N (no blocking)
int col_gen_row_block(int f64, int N, int want) {
  if (want <= 0 || want >= N || (want & (want - 1))) return N;
  switch (N) {
#define X(NN) case NN: { using C = GenSel<float, NN>::type; using D = GenSel<double, NN>::type;                              \
    const int np = f64 ? D::NPASS : C::NPASS, L = f64 ? NN / D::RL : NN / C::RL; return (np >= 2 && L % want == 0) ? want : N; }
    RF_COL_SIZES(X)
#undef X
    default: return N;
  }
}

// does the fast-generation x pass of this length fit a CU's LDS (tile + twiddles + the generation tables)?
bool col_fastgen_supported(int f64, int N) {
  switch (N) {
#define X(NN) case NN: return f64 ? GenSel<double, NN>::type::LDS_BYTES + FastGenColIO64<1, 0>::LDS_EXTRA <= 160 * 1024 \
                                  : GenSel<float, NN>::type::LDS_BYTES + FastGenColIOT<1, 0>::LDS_EXTRA <= 160 * 1024;
    RF_COL_SIZES(X)
#undef X
    default: return false;
  }
}

// replicated-generation mode stores rows [rank N/P, (rank+1) N/P) only and tests the row offset m * L of the last pass:
// available when the x pass has an LDS stage and N/P is a multiple of L = N / (radix of the last pass)
bool col_replicate_supported(int f64, int N, int nranks) {
  if (!col_fastgen_supported(f64, N) || nranks < 1 || N % nranks) return false;
  switch (N) {
#define X(NN) case NN: { using C = GenSel<float, NN>::type; return C::NPASS >= 2 && (NN / nranks) % (NN / C::RL) == 0; }
    RF_COL_SIZES(X)
#undef X
    default: return false;
  }
}

hipError_t launch_col_gen(int f64, int N, void* W, ColGeom g, long long ncols, const GenParams& gp,
                          const void* kspace, int kz0, int nzl, const void* tw, hipStream_t s, bool po) {
  if (f64) return launch_t<double>(N, (cplx<double>*)W, g, ncols, gp, (const cplx<double>*)kspace, kz0, nzl, (const cplx<double>*)tw, s, po);
  return launch_t<float>(N, (cplx<float>*)W, g, ncols, gp, (const cplx<float>*)kspace, kz0, nzl, (const cplx<float>*)tw, s, po);
}
}  // namespace rf
