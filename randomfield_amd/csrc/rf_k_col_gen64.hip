// the float64 fast-generation x pass (values drawn in float32 arithmetic and widened): its own translation unit, compiled with the
// compiler's default scheduling strategy (rf_col_gen_launch.h)
#include "rf_col_gen_launch.h"

namespace rf {

hipError_t launch_col_fastgen64(int N, void* W, ColGeom g, long long ncols, const FastGenParams& gp, int kz0, int nzl,
                                const void* tw, hipStream_t s, bool po, hipEvent_t after_repair, int x0, int x1, void* pot, void* fixbuf) {
  if (!col_fastgen_supported(1, N)) return po ? hipSuccess : hipErrorInvalidValue;
  const bool slab = x0 > 0 || x1 < N;          // replicated-generation mode: the SLAB instantiations guard their stores
  if (gp.noise || (gp.noise32 && !po)) return hipErrorInvalidValue;     // resident deviates: the exact kernel serves float64 plans
  if (gp.emit_potential || po) {               // POT = 2: the pass transforms pscale * delta(k) / k^2 (rf_realise_scaled_potential)
    if (gp.emit_potential && (slab || pot)) return hipErrorInvalidValue;
    if (N == 1024) {                   // (as two 512-point transforms per tile, Col2: two workgroups per CU -- DESIGN.md 3.10)
      using C1 = GenSel<double, 512>::type;
      hipError_t e = launch_fast_one2<C1, FastGenColIO64<0, 0, 2, 2>, FastGenColIO64<1, 0, 2, 2>>(
          gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, (cplx<double>*)fixbuf);
      if (!po || e != hipSuccess) return e;
    }
    switch (N) {
#define X(NN) case NN: { if (!col_fastgen_supported(1, NN)) { if (po) break; return hipErrorInvalidValue; } hipError_t e = launch_fast_one<typename GenSel<double, NN>::type, FastGenColIO64<0, 0, 2>, FastGenColIO64<1, 0, 2>, cplx<double>>(gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, x0, x1, (cplx<double>*)fixbuf); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  if (pot || po) {                             // the second store stream: delta(k) / k^2, values widened
    if (slab && pot) return hipErrorInvalidValue;
    switch (N) {
#define X(NN) case NN: { if (!col_fastgen_supported(1, NN)) { if (po) break; return hipErrorInvalidValue; } hipError_t e = launch_fast_one<typename GenSel<double, NN>::type, FastGenColIO64<0, 0, 1>, FastGenColIO64<1, 0, 1>, cplx<double>>(gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, x0, x1, (cplx<double>*)fixbuf, (cplx<double>*)pot); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  if (N == 1024 && (!slab || po)) {
    using C1 = GenSel<double, 512>::type;
    hipError_t e = launch_fast_one2<C1, FastGenColIO64<0, 0, 0, 2>, FastGenColIO64<1, 0, 0, 2>>(
        gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, (cplx<double>*)fixbuf);
    if (!po || e != hipSuccess) return e;
  }
  if (slab || po) {
    switch (N) {
#define X(NN) case NN: { if (!col_fastgen_supported(1, NN)) { if (po) break; return hipErrorInvalidValue; } hipError_t e = launch_fast_one<typename GenSel<double, NN>::type, FastGenColIO64<0, 1>, FastGenColIO64<1, 1>, cplx<double>>(gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, x0, x1, (cplx<double>*)fixbuf); if (!po || e != hipSuccess) return e; break; }
      RF_COL_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  switch (N) {
#define X(NN) case NN: { if (!col_fastgen_supported(1, NN)) return po ? hipSuccess : hipErrorInvalidValue; return launch_fast_one<typename GenSel<double, NN>::type, FastGenColIO64<0, 0>, FastGenColIO64<1, 0>, cplx<double>>(gp, (cplx<double>*)W, g, ncols, kz0, nzl, (const cplx<double>*)tw, s, po, after_repair, x0, x1, (cplx<double>*)fixbuf); }
    RF_COL_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace rf
