// persistent fused x (generation) + y pass: see rf_fused.h
#include "rf_fused.h"
#include "rf_launch.h"
#include <cstdio>
#include <cstdlib>

namespace rf {
namespace {
template <int N>
hipError_t launch_fused_t(cplx<float>* W, ColGeom gx, ColGeom gy, const FastGenParams& gp, int kz0, int nzl, int nx, int ny,
                          const cplx<float>* tw, unsigned* ctrl, unsigned* abort_flag, int skip_kz0_tile, hipStream_t s, bool prepare_only) {
  using CX = typename GenSel<float, N>::type;
  using CY = typename ColSel<float, N>::type;
  using IOX = FastGenColIOT<0, 0, 0>;
  using IOY = PlainColIO<float>;
  auto k = xy_fused_kernel<CX, CY, IOX, IOY>;
  constexpr int lds_bytes = CX::LDS_BYTES + IOX::LDS_EXTRA;
  static int wgs_per_cu = 0, ncu = 0;
  if (wgs_per_cu == 0) {
    if (lds_bytes > 65536) {
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      if (e != hipSuccess) return e;
    }
    int dev = 0, nb = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k, CX::NT, lds_bytes);
    if (e != hipSuccess) return e;
    wgs_per_cu = nb < 1 ? 1 : nb;
    if (getenv("RANDOMFIELD_FUSED_DEBUG")) fprintf(stderr, "[fused N=%d] CUs %d, workgroups per CU %d, LDS %d B\n", N, ncu, nb, lds_bytes);
  }
  if (prepare_only) return hipSuccess;
  if (nzl % CX::TC || gp.nbins > FAST_LDS_BINS - 1 || nx != ny) return hipErrorInvalidValue;
  FusedSched sc;
  sc.ctrl = ctrl;
  sc.abort_flag = abort_flag;
  sc.ktiles = nzl / CX::TC;
  constexpr int line_tiles = 128 / (CX::TC * 8) > 1 ? 128 / (CX::TC * 8) : 1;   // tiles per 128-byte line
  if (sc.ktiles % line_tiles) return hipErrorInvalidValue;
  sc.G = line_tiles;                            // an item owns whole 128-byte lines (the coherence argument of rf_fused.h)
  if (const char* e = getenv("RANDOMFIELD_FUSED_G")) { const int g = atoi(e); if (g >= line_tiles && g % line_tiles == 0 && sc.ktiles % g == 0) sc.G = g; }
  sc.nchunks = sc.ktiles / sc.G;
  sc.per = nx;
  // the Y stream starts one chunk plus `slack` items behind the X stream: by the time Y(c) items are dispatched
  // the last X(c) items have long been dispatched, so they wait little; slack = 2 x the resident workgroups
  unsigned slack = 2u * (unsigned)(wgs_per_cu * ncu);
  if (const char* e = getenv("RANDOMFIELD_FUSED_SLACK")) slack = (unsigned)atoi(e);
  sc.delay = ((unsigned)sc.per + slack + FUSED_BLOCK - 1) / FUSED_BLOCK * FUSED_BLOCK;
  if (sc.per % FUSED_BLOCK) return hipErrorInvalidValue;
  sc.spin_limit = 4u * 1000u * 1000u;           // x ~0.5 us per poll: seconds, far beyond any legitimate wait
  if (4 * sc.nchunks > fused_ctrl_words()) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(ctrl, 0, (size_t)4 * sc.nchunks * sizeof(unsigned), s);
  if (e != hipSuccess) return e;
  IOX io0; io0.base = W; io0.g = gx; io0.gp = gp; io0.kz0 = kz0; io0.nzl = nzl; io0.rec = nullptr;
  IOY ioy; ioy.base = W; ioy.g = gy;
  hipLaunchKernelGGL(k, dim3(fused_total_items(sc)), dim3(CX::NT), lds_bytes, s, io0, ioy, tw, sc, skip_kz0_tile);
  return hipGetLastError();
}
}  // namespace

int fused_ctrl_words() { return 4 * 256; }      // up to 256 chunks (2048^3: nz/2 / 8 / 2 = 64)

bool xy_fused_supported(int nx, int ny) { return nx == ny && (nx == 256 || nx == 512 || nx == 1024 || nx == 2048); }

hipError_t launch_xy_fused(int N, void* W, ColGeom gx, ColGeom gy, const FastGenParams& gp, int kz0, int nzl, int nx, int ny,
                           const void* tw, unsigned* ctrl, unsigned* abort_flag, int skip_kz0_tile, hipStream_t s, bool po) {
  switch (N) {
#define X(NN) case NN: return launch_fused_t<NN>((cplx<float>*)W, gx, gy, gp, kz0, nzl, nx, ny, (const cplx<float>*)tw, ctrl, abort_flag, skip_kz0_tile, s, po);
    X(256) X(512) X(1024) X(2048)
#undef X
    default: return hipErrorInvalidValue;
  }
}
}  // namespace rf
