// x pass of the c2r transform with generation (rows K,T,R,S) fused into its load
#include "rf_kernels.h"
#include "rf_launch.h"


#ifndef RF_COL2_2048
#define RF_COL2_2048 1                 // length-2048 float32 passes as two 1024-point transforms per tile (Col2); 0 = the whole-column kernels
#endif
#ifndef RF_COL2_F64_1024
#define RF_COL2_F64_1024 1             // the float64 generation pass of length 1024 as two 512-point transforms per tile (Col2): two workgroups per CU
#endif
namespace rf {
namespace {
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

// fast float32 generation + x pass of length 2 C1::N through Col2 (native generator, whole grid or kz slab; no potential store,
// no resident deviates, no x-slab restriction: those keep the whole-column kernel)
template <class C1, class IO0, class IO1>
hipError_t launch_fast_one2(const FastGenParams& gp, cplx<typename C1::T>* W, ColGeom g, long long ncols, int kz0, int nzl,
                            const cplx<typename C1::T>* tw2, hipStream_t s, bool po, hipEvent_t after_repair) {
  if (nzl <= 0 || (nzl & (nzl - 1)) || ncols >= (1LL << 31) || g.needs_wide(C1::LMAX, C1::TC, (int)sizeof(cplx<typename C1::T>)) || !g.rows_ok(C1::N / C1::RL, C1::NPASS))
    return hipErrorInvalidValue;
  IO0 io0; io0.base = W; io0.g = g; io0.gp = gp; io0.kz0 = kz0; io0.nzl = nzl; io0.rec = nullptr; io0.pot = nullptr;
  IO1 io1; io1.base = W; io1.g = g; io1.gp = gp; io1.kz0 = kz0; io1.nzl = nzl; io1.rec = nullptr; io1.pot = nullptr;
  const bool split = nzl > C1::TC && nzl % C1::TC == 0;
  const long long tiles_per_iy = nzl / C1::TC, ntiles = ncols / C1::TC;
  // the split launch's repair kernel computes the repair values with all lanes (FIX = 2, ColFFT::fix_prepare)
  using IOC = typename IO1::template with_fix<2>;
  IOC ioc; ioc.base = W; ioc.g = g; ioc.gp = gp; ioc.kz0 = kz0; ioc.nzl = nzl; ioc.rec = nullptr; ioc.pot = nullptr;
  if (po) {
    hipError_t e = launch_one2<C1, IO0>(io0, ncols, tw2, s, true);
    if (e == hipSuccess) e = launch_one2<C1, IOC>(ioc, ncols, tw2, s, true);
    return e != hipSuccess ? e : launch_one2<C1, IO1>(io1, ncols, tw2, s, true);
  }
  if (!split) return launch_one2<C1, IO1>(io1, ncols, tw2, s, false);
  if (kz0 != 0) return launch_one2<C1, IO0>(io0, ncols, tw2, s, false);
  hipError_t e = launch_one2<C1, IOC>(ioc, ncols, tw2, s, false, ncols / nzl, tiles_per_iy, 0);
  if (e != hipSuccess || tiles_per_iy >= (1LL << 30) || ntiles >= (1LL << 31)) return e != hipSuccess ? e : hipErrorInvalidValue;
  if (after_repair && (e = hipEventRecord(after_repair, s)) != hipSuccess) return e;
  return launch_one2<C1, IO0>(io0, ncols, tw2, s, false, ntiles - ntiles / tiles_per_iy, 1, 0, (int)tiles_per_iy);
}

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

// Fast float32 generation.  The tiles that contain the kz = 0 slot (one per iy when a tile is narrower than a
// kz row) are run by the kernel WITH the Hermitian repair, which carries the extra register pressure only where
// it is needed; every other tile by the kernel WITHOUT it (skip_period = tiles per iy).
template <class C, class IO0, class IO1, class CT>
hipError_t launch_fast_one(const FastGenParams& gp, CT* W, ColGeom g, long long ncols, int kz0, int nzl,
                           const CT* tw, hipStream_t s, bool po, hipEvent_t after_repair, int x0, int x1, CT* pot = nullptr) {
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
  // the split launch's repair kernel of the long passes computes the repair values with all lanes (FIX = 2, ColFFT::fix_prepare)
  constexpr bool coop = C::N >= 512 && C::NPASS >= 2;
  using IOC = typename IO1::template with_fix<coop ? 2 : 1>;
  IOC ioc; ioc.base = base; ioc.g = g; ioc.gp = gp; ioc.kz0 = kz0; ioc.nzl = nzl; ioc.rec = nullptr; ioc.x0 = x0; ioc.x1 = x1; ioc.pot = pot;
  if (po) {
    hipError_t e = launch_one<C, IO0>(io0, ncols, tw, s, true);
    if (e == hipSuccess && coop) e = launch_one<C, IOC>(ioc, ncols, tw, s, true);
    return e != hipSuccess ? e : launch_one<C, IO1>(io1, ncols, tw, s, true);
  }
  if (!split) return launch_one<C, IO1>(io1, ncols, tw, s, false);
  if (kz0 != 0) return launch_one<C, IO0>(io0, ncols, tw, s, false);   // only the slab that owns kz = 0 needs the repair
  // first the (few) tiles that hold slot kz = 0, with the repair; then every other tile without it
  hipError_t e = launch_one<C, IOC>(ioc, ncols, tw, s, false, ncols / nzl, tiles_per_iy, 0);
  if (e != hipSuccess || tiles_per_iy >= (1LL << 30) || ntiles >= (1LL << 31)) return e != hipSuccess ? e : hipErrorInvalidValue;
  if (after_repair && (e = hipEventRecord(after_repair, s)) != hipSuccess) return e;
  return launch_one<C, IO0>(io0, ncols, tw, s, false, ntiles - ntiles / tiles_per_iy, 1, 0, (int)tiles_per_iy);
}

hipError_t launch_col_fastgen(int f64, int N, void* W, ColGeom g, long long ncols, const FastGenParams& gp, int kz0, int nzl,
                              const void* tw, hipStream_t s, bool po, hipEvent_t after_repair, int x0, int x1, void* pot) {
  if (!col_fastgen_supported(f64, N)) return po ? hipSuccess : hipErrorInvalidValue;    // the caller keeps the exact kernel
  const bool slab = x0 > 0 || x1 < N;          // replicated-generation mode: the SLAB instantiations guard their stores
  if (gp.emit_potential || po) {               // POT = 2: the pass transforms pscale * delta(k) / k^2 (rf_realise_scaled_potential)
    if (gp.emit_potential && (slab || pot || gp.noise)) return hipErrorInvalidValue;
    if (!f64) {
      if (gp.noise32 || po)                    // ... of the replayed deviates (float32 pairs in the replay's runs)
        switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 0, 2, 2>, FastGenColIOT<0, 1, 0, 2, 2>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1); if (!po || e != hipSuccess) return e; break; }
          RF_COL_SIZES(X)
#undef X
          default: return hipErrorInvalidValue;
        }
      if (RF_COL2_2048 && N == 2048) {
        using C1 = GenSel<float, 1024>::type;
        hipError_t e = launch_fast_one2<C1, FastGenColIOT<0, 0, 0, 2, 0, 2>, FastGenColIOT<0, 1, 0, 2, 0, 2>>(
            gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair);
        if (!po || e != hipSuccess) return e;
      }
      switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 0, 2>, FastGenColIOT<0, 1, 0, 2>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1); if (!po || e != hipSuccess) return e; break; }
        RF_COL_SIZES(X)
#undef X
        default: return hipErrorInvalidValue;
      }
    } else {
      if (gp.noise32 && !po) return hipErrorInvalidValue;
      if (RF_COL2_F64_1024 && N == 1024) {
        using C1 = GenSel<double, 512>::type;
        hipError_t e = launch_fast_one2<C1, FastGenColIO64<0, 0, 2, 2>, FastGenColIO64<1, 0, 2, 2>>(
            gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair);
        if (!po || e != hipSuccess) return e;
      }
      switch (N) {
#define X(NN) case NN: { if (!col_fastgen_supported(1, NN)) { if (po) break; return hipErrorInvalidValue; } hipError_t e = launch_fast_one<typename GenSel<double, NN>::type, FastGenColIO64<0, 0, 2>, FastGenColIO64<1, 0, 2>, cplx<double>>(gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, x0, x1); if (!po || e != hipSuccess) return e; break; }
        RF_COL_SIZES(X)
#undef X
        default: return hipErrorInvalidValue;
      }
    }
  }
  if ((gp.noise || po) && !f64) {              // resident float64 deviates (rng='reference') through the fast float32 sigma path
    if (gp.noise && (slab || pot)) return hipErrorInvalidValue;
    switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 0, 0, 1>, FastGenColIOT<0, 1, 0, 0, 1>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  } else if (gp.noise) {
    return hipErrorInvalidValue;
  }
  if ((gp.noise32 || po) && !f64) {            // resident float32 deviates, without / with the potential store
    if (gp.noise32 && slab) return hipErrorInvalidValue;
    if (!pot || po)
      switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 0, 0, 2>, FastGenColIOT<0, 1, 0, 0, 2>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1); if (!po || e != hipSuccess) return e; break; }
        RF_COL_SIZES(X)
#undef X
        default: return hipErrorInvalidValue;
      }
    if (pot || po)
      switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 0, 1, 2>, FastGenColIOT<0, 1, 0, 1, 2>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)pot); if (!po || e != hipSuccess) return e; break; }
        RF_COL_SIZES(X)
#undef X
        default: return hipErrorInvalidValue;
      }
  } else if (gp.noise32) {
    return hipErrorInvalidValue;
  }
  if ((pot || po) && !f64) {                   // generation + potential store (save_potential=True), float32, whole grid
    if (slab && pot) return hipErrorInvalidValue;
    switch (N) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 0, 1>, FastGenColIOT<0, 1, 0, 1>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1, (cplx<float>*)pot); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  } else if (pot || po) {                      // float64 plans: the same second store stream, values widened
    if (slab && pot) return hipErrorInvalidValue;
    switch (N) {
#define X(NN) case NN: { if (!col_fastgen_supported(1, NN)) { if (po) break; return hipErrorInvalidValue; } hipError_t e = launch_fast_one<typename GenSel<double, NN>::type, FastGenColIO64<0, 0, 1>, FastGenColIO64<1, 0, 1>, cplx<double>>(gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, x0, x1, (cplx<double>*)pot); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
#define RF_FAST(T, IO0, IO1)                                                                                              \
  switch (N) {                                                                                                           \
    RF_COL_SIZES(X)                                                                                                       \
    default: return hipErrorInvalidValue;                                                                                \
  }
  if (RF_COL2_F64_1024 && f64 && N == 1024 && (!slab || po)) {
    using C1 = GenSel<double, 512>::type;
    hipError_t e = launch_fast_one2<C1, FastGenColIO64<0, 0, 0, 2>, FastGenColIO64<1, 0, 0, 2>>(
        gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair);
    if (!po || e != hipSuccess) return e;
  }
  if (f64) {
    if (slab || po) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<double, NN>::type, FastGenColIO64<0, 1>, FastGenColIO64<1, 1>, cplx<double>>(gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, x0, x1); if (!po || e != hipSuccess) return e; break; }
      RF_FAST(double, 0, 0)
#undef X
    }
#define X(NN) case NN: return launch_fast_one<typename GenSel<double, NN>::type, FastGenColIO64<0, 0>, FastGenColIO64<1, 0>, cplx<double>>(gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, x0, x1);
    RF_FAST(double, 0, 0)
#undef X
  }
  if (slab || po) {
#define X(NN) case NN: { hipError_t e = launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<0, 0, 1>, FastGenColIOT<0, 1, 1>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1); if (!po || e != hipSuccess) return e; break; }
    RF_FAST(float, 0, 0)
#undef X
  }
#ifndef RF_GEN_AB
#define RF_GEN_AB 0                    // ablation mask of the benchmarked kernel (rf_core.h fast_gen_pair_at); 0 in the product
#endif
  if (RF_COL2_2048 && N == 2048 && !f64 && (!slab || po)) {
    using C1 = GenSel<float, 1024>::type;
    hipError_t e = launch_fast_one2<C1, FastGenColIOT<RF_GEN_AB, 0, 0, 0, 0, 2>, FastGenColIOT<RF_GEN_AB, 1, 0, 0, 0, 2>>(
        gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair);
    if (!po || e != hipSuccess) return e;
  }
#define X(NN) case NN: return launch_fast_one<typename GenSel<float, NN>::type, FastGenColIOT<RF_GEN_AB, 0, 0>, FastGenColIOT<RF_GEN_AB, 1, 0>, cplx<float>>(gp, (cplx<float>*)W, g, ncols, kz0, nzl, (const cplx<float>*)tw, s, po, after_repair, x0, x1);
  RF_FAST(float, 0, 0)
#undef X
#undef RF_FAST
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
                                  : GenSel<float, NN>::type::LDS_BYTES + FastGenColIOT<0, 1, 0>::LDS_EXTRA <= 160 * 1024;
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
