// kbench.hip -- kernel micro-benchmark (development tool, not part of the product).
// Times variants of the FFT pass kernels and pure-copy kernels with the same access
// patterns (the "pattern ceiling") on a 1024^3 half-complex array.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I randomfield_amd/csrc tools/kbench.hip -o gpurun_out/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#include "rf_kernels.h"
#include "rf_host.h"

using namespace rf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static int NX = 1024, NY = 1024, NZ = 1024;
static rf::FastGenParams g_fp;

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <class F> float run(F f, int reps = 5) {
    f();  // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
  }
};

// ---- pattern ceilings ---------------------------------------------------------------------------
// tile copy: each workgroup reads (and/or writes) N rows x SEG bytes segments at a given row stride
template <int SEG_BYTES, int NT, bool READ, bool WRITE>
__global__ __launch_bounds__(NT) void tile_copy_kernel(float4* base, long long row_stride16, long long inner16,
                                                       long long outer_stride16, int nrows, long long ntiles) {
  constexpr int LPR = SEG_BYTES / 16;
  const long long tile = xcd_tile(blockIdx.x, ntiles);
  const int lp = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
  const long long C = tile * LPR + lp;
  float4* p = base + (C / inner16) * outer_stride16 + (C % inner16);
  float4 acc = make_float4(0, 0, 0, 0);
  constexpr int RPI = NT / LPR;
  if (READ) {
#pragma unroll 8
    for (int r = r0; r < nrows; r += RPI) {
      float4 v = p[(long long)r * row_stride16];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (WRITE) {
    acc.x += (float)threadIdx.x;
#pragma unroll 8
    for (int r = r0; r < nrows; r += RPI) p[(long long)r * row_stride16] = acc;
  } else if (acc.x == 123.456f) {
    p[0] = acc;
  }
}

__global__ __launch_bounds__(256) void linear_copy_kernel(float4* __restrict__ p, long long n16, bool read, bool write) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
    float4 v = make_float4(1, 2, 3, 4);
    if (read) v = p[i];
    v.x += 1.0f;
    if (write) p[i] = v;
    else if (v.x == 123.456f) p[0] = v;
  }
}

// ---- FFT pass variants ------------------------------------------------------------------------------
template <class C>
float bench_col(Timer& t, cplx<float>* W, const cplx<float>* tw, bool ypass) {
  const long long nzc = NZ / 2;
  PlainColIO<float> io;
  io.base = W;
  long long ncols;
  if (ypass) { io.g = ColGeom{nzc, (long long)NY * nzc, nzc}; ncols = (long long)NX * nzc; }
  else { io.g = ColGeom{(long long)NY * nzc, 0, (long long)NY * nzc}; ncols = (long long)NY * nzc; }
  const long long ntiles = ncols / C::TC;
  auto k = col_kernel<C, +1, PlainColIO<float>>;
  if (C::LDS_BYTES > 65536) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
  return t.run([&]() { hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), C::LDS_BYTES, 0, io, tw, ntiles, 1LL, 0LL, 0); });
}

template <class C, int AB = 0, int FIX = 1>
float bench_fastgen(Timer& t, cplx<float>* W, const cplx<float>* tw, const FastGenParams& fp) {
  const long long nzc = NZ / 2;
  FastGenColIOT<AB, FIX> io;
  io.rec = nullptr;
  io.base = W; io.g = ColGeom{(long long)NY * nzc, 0, (long long)NY * nzc}; io.gp = fp; io.kz0 = 0; io.nzl = (int)nzc;
  const long long ncols = (long long)NY * nzc, ntiles = ncols / C::TC;
  auto k = col_kernel<C, +1, FastGenColIOT<AB, FIX>>;
  constexpr int lds = C::LDS_BYTES + FastGenColIOT<AB, FIX>::LDS_EXTRA;
  if (lds > 65536) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return t.run([&]() { hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), lds, 0, io, tw, ntiles, 1LL, 0LL, 0); });
}

template <class C>
float bench_row(Timer& t, cplx<float>* W, const cplx<float>* tw, double* partials) {
  PlainRowIO<float> io;
  io.base = W; io.scale = 1.0f; io.M_of = C::M;   // scale 1 keeps repeated in-place runs finite-ish
  const long long nrows = (long long)NX * NY * (NZ / 2) / C::M, ntiles = (nrows + C::NRT - 1) / C::NRT;
  auto k = row_c2r_kernel<C, PlainRowIO<float>>;
  const int lds = C::LDS_BYTES > 64 ? C::LDS_BYTES : 64;   // tile + twiddles
  if (lds > 65536) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return t.run([&]() { hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), lds, 0, io, tw, nrows, partials); });
}

static void report(const char* name, float ms, double bytes) {
  printf("%-64s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const long long nzc = NZ / 2;
  const size_t ncplx = (size_t)NX * NY * nzc;
  const double sweep = (double)ncplx * 8;
  cplx<float>* W;
  CK(hipMalloc((void**)&W, ncplx * 8));
  CK(hipMemset(W, 0, ncplx * 8));
  auto twh = make_twiddles<float>(1024);
  cplx<float>* tw;
  CK(hipMalloc((void**)&tw, twh.size() * 8));
  CK(hipMemcpy(tw, twh.data(), twh.size() * 8, hipMemcpyHostToDevice));
  double* partials;
  CK(hipMalloc((void**)&partials, 2 * sizeof(double) * (size_t)NX * NY));
  Timer t;

  // --- ceilings
  {
    float4* p = (float4*)W;
    const long long n16 = ncplx / 2;
    report("linear copy read+write (float4)", t.run([&]() { hipLaunchKernelGGL(linear_copy_kernel, dim3(256 * 16), dim3(256), 0, 0, p, n16, true, true); }), 2 * sweep);
    report("linear read only", t.run([&]() { hipLaunchKernelGGL(linear_copy_kernel, dim3(256 * 16), dim3(256), 0, 0, p, n16, true, false); }), sweep);
    report("linear write only", t.run([&]() { hipLaunchKernelGGL(linear_copy_kernel, dim3(256 * 16), dim3(256), 0, 0, p, n16, false, true); }), sweep);
    // y-pass pattern: tile = (ix, kz segment), rows = iy at stride nzc complex
    const long long rs16 = nzc / 2, in16_y = nzc / 2, os16_y = (long long)NY * nzc / 2;
    const long long in16_x = (long long)NY * nzc / 2, rs16_x = (long long)NY * nzc / 2;
#define TILE(SEG, NT, RD, WR, label, in16, os16, rs, total16)                                                   \
    {                                                                                                            \
      const long long ntiles = (total16) / (SEG / 16);                                                           \
      report(label, t.run([&]() { hipLaunchKernelGGL((tile_copy_kernel<SEG, NT, RD, WR>), dim3((unsigned)ntiles), dim3(NT), 0, 0, p, rs, in16, os16, 1024, ntiles); }), \
             ((RD ? 1 : 0) + (WR ? 1 : 0)) * sweep);                                                             \
    }
    TILE(64, 256, true, true, "y-pattern tile copy r+w  64 B segments, 256 thr", in16_y, os16_y, rs16, (long long)NX * nzc / 2)
    TILE(64, 512, true, true, "y-pattern tile copy r+w  64 B segments, 512 thr", in16_y, os16_y, rs16, (long long)NX * nzc / 2)
    TILE(128, 512, true, true, "y-pattern tile copy r+w 128 B segments, 512 thr", in16_y, os16_y, rs16, (long long)NX * nzc / 2)
    TILE(256, 512, true, true, "y-pattern tile copy r+w 256 B segments, 512 thr", in16_y, os16_y, rs16, (long long)NX * nzc / 2)
    TILE(64, 512, false, true, "x-pattern tile write     64 B segments, 512 thr", in16_x, 0, rs16_x, (long long)NY * nzc / 2)
    TILE(128, 512, false, true, "x-pattern tile write    128 B segments, 512 thr", in16_x, 0, rs16_x, (long long)NY * nzc / 2)
    TILE(64, 512, true, true, "x-pattern tile copy r+w  64 B segments, 512 thr", in16_x, 0, rs16_x, (long long)NY * nzc / 2)
    TILE(128, 512, true, true, "x-pattern tile copy r+w 128 B segments, 512 thr", in16_x, 0, rs16_x, (long long)NY * nzc / 2)
  }

  // --- column pass variants (y geometry, in place, r+w) and x geometry
#define COL(R1, R2, R3, TC, NT)                                                                       \
  {                                                                                                   \
    using C = ColCfg<float, 1024, R1, R2, R3, TC, NT>;                                                \
    char nm[128];                                                                                     \
    snprintf(nm, sizeof nm, "y pass  radix %2d,%2d,%2d TC=%2d NT=%4d LDS=%6d", R1, R2, R3, TC, NT, C::LDS_BYTES); \
    report(nm, bench_col<C>(t, W, tw, true), 2 * sweep);                                              \
    snprintf(nm, sizeof nm, "x geom  radix %2d,%2d,%2d TC=%2d NT=%4d (plain r+w)", R1, R2, R3, TC, NT); \
    report(nm, bench_col<C>(t, W, tw, false), 2 * sweep);                                             \
  }
  COL(8, 16, 8, 8, 512)
  COL(16, 16, 4, 8, 256)
  COL(4, 16, 16, 8, 512)
  COL(16, 8, 8, 8, 512)

  // --- fused generation x pass
  {
    std::vector<double> lk(500), sg(500);
    for (int i = 0; i < 500; ++i) { lk[i] = -4.0 + i * (5.34 / 499); sg[i] = 1e5 * exp(-0.3 * (lk[i] + 2) * (lk[i] + 2)); }
    SigmaTableHost tab;
    build_sigma_table(lk.data(), sg.data(), 500, tab);
    std::vector<FastRec> rec;
    FastGenParams fp;
    const double k0 = 2 * M_PI / 2.5;
    double x0, dx;
    if (!build_fast_records(tab, log10(k0 / 1024) - 0.01, log10(k0 * sqrt(3.0) / 2) + 0.01, rec, x0, dx)) { printf("no fast records\n"); return 1; }
    fp.u_scale = (float)(0.5 * log10(2.0) / dx); fp.u_off = (float)(-x0 / dx); fp.dkx = (float)(k0 / 1024);
    std::vector<float> k2(1024);
    for (int i = 0; i < 1024; ++i) { int j = i < 512 ? i : i - 1024; double k = j * k0 / 1024; k2[i] = (float)(k * k); }
    float* dk2; FastRec* drec;
    CK(hipMalloc((void**)&dk2, 1024 * 4)); CK(hipMemcpy(dk2, k2.data(), 1024 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&drec, rec.size() * sizeof(FastRec))); CK(hipMemcpy(drec, rec.data(), rec.size() * sizeof(FastRec), hipMemcpyHostToDevice));
    fp.nx = NX; fp.ny = NY; fp.nz = NZ; fp.dky = fp.dkx; fp.dkz = fp.dkx; fp.rec = drec; fp.nbins = (int)rec.size();
    fp.seed = 123; fp.seed_dev = nullptr; fp.noise = nullptr;
    printf("fast records: %d bins\n", fp.nbins);
    g_fp = fp;
#define GEN(R1, R2, R3, TC, NT)                                                                     \
    {                                                                                                 \
      using C = ColCfg<float, 1024, R1, R2, R3, TC, NT>;                                              \
      char nm[128];                                                                                   \
      snprintf(nm, sizeof nm, "x pass fast-gen radix %2d,%2d,%2d TC=%2d NT=%4d", R1, R2, R3, TC, NT); \
      report(nm, bench_fastgen<C>(t, W, tw, fp), sweep);                                              \
    }
    GEN(8, 16, 8, 8, 512)
#define ABL(AB, label) { using C = ColCfg<float, 1024, 8, 16, 8, 8, 512>; report(label, bench_fastgen<C, AB, 0>(t, W, tw, fp), sweep); }
    ABL(8, "  variant: Philox4x32-7 instead of -10")
    ABL(1, "  ablation: no Philox")
    ABL(2, "  ablation: no sigma lookup")
    ABL(4, "  ablation: no Box-Muller")
    ABL(3, "  ablation: no Philox, no sigma")
    ABL(7, "  ablation: no Philox, no sigma, no Box-Muller (FFT + stores only)")
#define GENF(R1, R2, R3, TC, NT, FIX)                                                              \
    {                                                                                                 \
      using C = ColCfg<float, 1024, R1, R2, R3, TC, NT>;                                              \
      char nm[128];                                                                                   \
      snprintf(nm, sizeof nm, "x pass fast-gen radix %2d,%2d,%2d TC=%2d NT=%4d fix=%d", R1, R2, R3, TC, NT, FIX); \
      report(nm, bench_fastgen<C, 0, FIX>(t, W, tw, fp), sweep);                                      \
    }
    GENF(8, 16, 8, 8, 512, 0)
    GENF(4, 16, 16, 8, 512, 0)
    GENF(4, 16, 16, 8, 512, 1)
    GENF(16, 8, 8, 8, 512, 0)
    GENF(8, 8, 16, 8, 512, 0)
  }

  // --- do an x pass (VALU-bound) and a y pass (HBM-bound) co-execute from two streams?
  {
    cplx<float>* W2;
    CK(hipMalloc((void**)&W2, ncplx * 8));
    CK(hipMemset(W2, 0, ncplx * 8));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    using C = ColCfg<float, 1024, 8, 16, 8, 8, 512>;
    PlainColIO<float> yio; yio.base = W2; yio.g = ColGeom{nzc, (long long)NY * nzc, nzc};
    const long long yt = (long long)NX * nzc / C::TC;
    auto ky = col_kernel<C, +1, PlainColIO<float>>;
    CK(hipFuncSetAttribute((const void*)ky, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    FastGenColIOT<0, 0> xio; xio.rec = nullptr; xio.base = W; xio.g = ColGeom{(long long)NY * nzc, 0, (long long)NY * nzc}; xio.gp = g_fp; xio.kz0 = 0; xio.nzl = (int)nzc;
    auto kx = col_kernel<C, +1, FastGenColIOT<0, 0>>;
    constexpr int xl = C::LDS_BYTES + FastGenColIOT<0, 0>::LDS_EXTRA;
    CK(hipFuncSetAttribute((const void*)kx, hipFuncAttributeMaxDynamicSharedMemorySize, xl));
    const long long xt = (long long)NY * nzc / C::TC;
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::high_resolution_clock::now();
      hipLaunchKernelGGL(kx, dim3((unsigned)xt), dim3(C::NT), xl, s1, xio, tw, xt, 1LL, 0LL, 0);
      hipLaunchKernelGGL(ky, dim3((unsigned)yt), dim3(C::NT), C::LDS_BYTES, s2, yio, tw, yt, 1LL, 0LL, 0);
      CK(hipDeviceSynchronize());
      auto t1 = std::chrono::high_resolution_clock::now();
      hipLaunchKernelGGL(kx, dim3((unsigned)xt), dim3(C::NT), xl, s1, xio, tw, xt, 1LL, 0LL, 0);
      CK(hipDeviceSynchronize());
      auto t2 = std::chrono::high_resolution_clock::now();
      hipLaunchKernelGGL(ky, dim3((unsigned)yt), dim3(C::NT), C::LDS_BYTES, s2, yio, tw, yt, 1LL, 0LL, 0);
      CK(hipDeviceSynchronize());
      auto t3 = std::chrono::high_resolution_clock::now();
      auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      printf("two streams: x||y together %.3f ms; x alone %.3f ms; y alone %.3f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3));
    }
  }

  // --- spatial split with CU masks: x pass on `nx_cu` CUs per XCD, y pass on the other CUs, concurrently
  {
    cplx<float>* W3;
    CK(hipMalloc((void**)&W3, ncplx * 8));
    CK(hipMemset(W3, 0, ncplx * 8));
    using C = ColCfg<float, 1024, 8, 16, 8, 8, 512>;
    using CY = ColCfg<float, 1024, 16, 8, 8, 8, 512>;
    PlainColIO<float> yio; yio.base = W3; yio.g = ColGeom{nzc, (long long)NY * nzc, nzc};
    const long long yt = (long long)NX * nzc / CY::TC;
    auto ky = col_kernel<CY, +1, PlainColIO<float>>;
    CK(hipFuncSetAttribute((const void*)ky, hipFuncAttributeMaxDynamicSharedMemorySize, CY::LDS_BYTES));
    FastGenColIOT<0, 0> xio; xio.rec = nullptr; xio.base = W; xio.g = ColGeom{(long long)NY * nzc, 0, (long long)NY * nzc}; xio.gp = g_fp; xio.kz0 = 0; xio.nzl = (int)nzc;
    auto kx = col_kernel<C, +1, FastGenColIOT<0, 0>>;
    constexpr int xl = C::LDS_BYTES + FastGenColIOT<0, 0>::LDS_EXTRA;
    CK(hipFuncSetAttribute((const void*)kx, hipFuncAttributeMaxDynamicSharedMemorySize, xl));
    const long long xt = (long long)NY * nzc / C::TC;
    for (int nx_cu : {8, 12, 16, 20}) {
      // CU numbering: assume 32 consecutive bits per XCD (8 XCDs); x gets the low nx_cu bits of each group
      uint32_t mx[8], my[8];
      for (int x = 0; x < 8; ++x) { mx[x] = (nx_cu >= 32) ? 0xFFFFFFFFu : ((1u << nx_cu) - 1u); my[x] = ~mx[x]; }
      hipStream_t sx, sy;
      if (hipExtStreamCreateWithCUMask(&sx, 8, mx) != hipSuccess || hipExtStreamCreateWithCUMask(&sy, 8, my) != hipSuccess) { printf("CU mask streams not available\n"); break; }
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::high_resolution_clock::now();
        hipLaunchKernelGGL(kx, dim3((unsigned)xt), dim3(C::NT), xl, sx, xio, tw, xt, 1LL, 0LL, 0);
        hipLaunchKernelGGL(ky, dim3((unsigned)yt), dim3(CY::NT), CY::LDS_BYTES, sy, yio, tw, yt, 1LL, 0LL, 0);
        CK(hipDeviceSynchronize());
        auto t1 = std::chrono::high_resolution_clock::now();
        hipLaunchKernelGGL(kx, dim3((unsigned)xt), dim3(C::NT), xl, sx, xio, tw, xt, 1LL, 0LL, 0);
        CK(hipDeviceSynchronize());
        auto t2 = std::chrono::high_resolution_clock::now();
        hipLaunchKernelGGL(ky, dim3((unsigned)yt), dim3(CY::NT), CY::LDS_BYTES, sy, yio, tw, yt, 1LL, 0LL, 0);
        CK(hipDeviceSynchronize());
        auto t3 = std::chrono::high_resolution_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        printf("CU split x:%2d/32 y:%2d/32 per XCD: together %.3f ms; x alone (masked) %.3f; y alone (masked) %.3f\n", nx_cu, 32 - nx_cu, ms(t0, t1), ms(t1, t2), ms(t2, t3));
      }
      CK(hipStreamDestroy(sx)); CK(hipStreamDestroy(sy));
    }
  }

  // --- row pass variants
  CK(hipMemset(W, 0, ncplx * 8));
#define ROW(R1, R2, R3, NRT, NT)                                                                   \
  {                                                                                                  \
    using C = RowCfg<float, 512, R1, R2, R3, NRT, NT>;                                               \
    char nm[128];                                                                                    \
    snprintf(nm, sizeof nm, "z pass  radix %2d,%2d,%2d NRT=%2d NT=%4d LDS=%6d", R1, R2, R3, NRT, NT, C::LDS_BYTES); \
    report(nm, bench_row<C>(t, W, twz, partials), 2 * sweep);                                        \
  }
  auto twzh = make_twiddles<float>(1024);
  cplx<float>* twz = tw;
  ROW(8, 8, 8, 8, 256)
  ROW(8, 8, 8, 16, 256)
  ROW(16, 8, 4, 16, 256)
  ROW(16, 8, 4, 8, 256)
  ROW(8, 16, 4, 16, 256)
  ROW(16, 4, 8, 16, 256)
  ROW(16, 8, 4, 16, 512)
  // rows of nz = 2048 (M = 1024): the z pass of the 8-GPU 2048^3 job; same number of cells
  {
    auto tw2h = make_twiddles<float>(2048);
    cplx<float>* tw2;
    CK(hipMalloc((void**)&tw2, tw2h.size() * 8));
    CK(hipMemcpy(tw2, tw2h.data(), tw2h.size() * 8, hipMemcpyHostToDevice));
#define ROW2(R1, R2, R3, NRT, NT)                                                                  \
  {                                                                                                  \
    using C = RowCfg<float, 1024, R1, R2, R3, NRT, NT>;                                              \
    char nm[128];                                                                                    \
    snprintf(nm, sizeof nm, "z pass M=1024 radix %2d,%2d,%2d NRT=%2d NT=%4d LDS=%6d", R1, R2, R3, NRT, NT, C::LDS_BYTES); \
    report(nm, bench_row<C>(t, W, tw2, partials), 2 * sweep);                                        \
  }
    ROW2(8, 8, 16, 4, 256)
    ROW2(8, 16, 8, 4, 256)
    ROW2(16, 8, 8, 4, 256)
    ROW2(8, 8, 16, 8, 256)
    ROW2(16, 8, 8, 8, 256)
    ROW2(16, 16, 4, 8, 256)
    ROW2(8, 16, 8, 8, 512)
  }
  return 0;
}
