// xbench.hip -- round-2 experiments on the generation + x-FFT pass and on the streaming ceilings
// (development tool, not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I randomfield_amd/csrc tools/xbench.hip -o tools/bin/xbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>
#include <vector>
#include "rf_kernels.h"
#include "rf_host.h"

using namespace rf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static int NX = 1024, NY = 1024, NZ = 1024;

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <class F> float run(F f, int reps = 6) {
    f(); f();
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

static void report(const char* name, float ms, double bytes) {
  printf("%-72s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
  fflush(stdout);
}

// ---- streaming ceilings with cache hints -----------------------------------------------------------
typedef float vec4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0 plain, 1 nontemporal
__global__ __launch_bounds__(256) void lin_kernel(vec4* __restrict__ p, long long n16, int rd, int wr) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
    vec4 v = {1, 2, 3, 4};
    if (rd) v = MODE ? __builtin_nontemporal_load(p + i) : p[i];
    v.x += 1.0f;
    if (wr) { if (MODE) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
    else if (v.x == 123.456f) p[0] = v;
  }
}
// block-contiguous copy: each workgroup owns a contiguous chunk (like the z pass: 8 rows of 4 KiB)
template <int MODE, int CHUNK16>
__global__ __launch_bounds__(256) void chunk_kernel(vec4* __restrict__ p, long long nchunks, int rd, int wr) {
  for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    vec4* q = p + c * CHUNK16;
    vec4 v[CHUNK16 / 256];
#pragma unroll
    for (int k = 0; k < CHUNK16 / 256; ++k) {
      if (rd) v[k] = MODE ? __builtin_nontemporal_load(q + k * 256 + threadIdx.x) : q[k * 256 + threadIdx.x];
      else v[k] = vec4{1, 2, 3, 4};
    }
#pragma unroll
    for (int k = 0; k < CHUNK16 / 256; ++k) {
      v[k].x += 1.0f;
      if (wr) { if (MODE) __builtin_nontemporal_store(v[k], q + k * 256 + threadIdx.x); else q[k * 256 + threadIdx.x] = v[k]; }
      else if (v[k].x == 123.456f) p[0] = v[k];
    }
  }
}
// strided tile copy (y / x pass pattern), optional nontemporal hints
template <int SEG_BYTES, int NT, bool READ, bool WRITE, int MODE>
__global__ __launch_bounds__(NT) void tile_copy_kernel(vec4* base, long long row_stride16, long long inner16,
                                                       long long outer_stride16, int nrows, long long ntiles) {
  constexpr int LPR = SEG_BYTES / 16;
  const long long tile = xcd_tile(blockIdx.x, ntiles);
  const int lp = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
  const long long C = tile * LPR + lp;
  vec4* p = base + (C / inner16) * outer_stride16 + (C % inner16);
  constexpr int RPI = NT / LPR;
  vec4 v[8];
  for (int rb = r0; rb < nrows; rb += 8 * RPI) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      vec4* q = p + (long long)(rb + k * RPI) * row_stride16;
      if (READ) v[k] = MODE ? __builtin_nontemporal_load(q) : *q; else v[k] = vec4{1, 2, 3, (float)k};
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      vec4* q = p + (long long)(rb + k * RPI) * row_stride16;
      v[k].x += 1.0f;
      if (WRITE) { if (MODE) __builtin_nontemporal_store(v[k], q); else *q = v[k]; }
      else if (v[k].x == 123.456f) p[0] = v[k];
    }
  }
}


// x-pattern write where a thread writes BOTH 64-byte halves of a 128-byte line with two back-to-back 16-byte stores
// (4 lanes per half): do half-line writes merge when they come from the same wave within nanoseconds?
template <int NT, int GAP>
__global__ __launch_bounds__(NT) void pair_write_kernel(vec4* base, long long row_stride16, long long inner16, int nrows, long long ntiles) {
  const long long tile = xcd_tile(blockIdx.x, ntiles);          // tile = one 128-byte column group
  const int lp = threadIdx.x % 4, r0 = threadIdx.x / 4;
  vec4* p = base + tile * 8 + lp;
  constexpr int RPI = NT / 4;
  vec4 v = {1, 2, 3, (float)threadIdx.x};
  if (GAP == 0) {
    for (int rb = r0; rb < nrows; rb += RPI) {
      vec4* q = p + (long long)rb * row_stride16;
      q[0] = v; q[4] = v;
    }
  } else {
    // all first halves, then all second halves (the pattern of two consecutive 8-column tiles in one workgroup)
    for (int rb = r0; rb < nrows; rb += RPI) p[(long long)rb * row_stride16] = v;
    for (int rb = r0; rb < nrows; rb += RPI) p[(long long)rb * row_stride16 + 4] = v;
  }
}


// ---- Infinity-Cache residency probe: a streaming read+write pass over a 4.3 GB array that ALSO bounces every chunk
// through a small ring (G x D x 64 KiB): written, then read back by another workgroup a couple of iterations later.
// If the ring stays in the 256 MB memory-side cache, the pass costs about as much as the plain copy; if the ring
// traffic goes to HBM it costs twice as much.  (Premise of fusing the y and z passes plane by plane.)
template <int MODE>   // 0: plain copy (no ring), 1: with the ring bounce
__global__ __launch_bounds__(512, 4) void mall_probe_kernel(vec4* __restrict__ A, long long nchunks, vec4* __restrict__ S, int D) {
  constexpr int K = 8, NT = 512;
  const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
  int it = 0;
  for (long long c = w; c < nchunks; c += G, ++it) {
    vec4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = __builtin_nontemporal_load(A + c * (NT * K) + k * NT + tid);
    if (MODE == 1) {
      vec4* wr = S + ((long long)(it % D) * G + w) * (NT * K);
#pragma unroll
      for (int k = 0; k < K; ++k) wr[k * NT + tid] = v[k];
      const int src = (w + 8 * 5) % G;                       // another workgroup of the same XCD group
      const vec4* rd = S + ((long long)((it + D - 1) % D) * G + src) * (NT * K);
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = __builtin_nontemporal_load(rd + k * NT + tid);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k].x += 1.0f; __builtin_nontemporal_store(v[k], A + c * (NT * K) + k * NT + tid); }
  }
}

namespace rf {
// (experiment; not in the product: in the pipeline it is no faster than one workgroup per tile)
// Persistent variant of col_kernel: gridDim.x workgroups (a multiple of 8, normally as many as are resident)
// walk all tiles.  The LDS tables are staged once per workgroup, and because a workgroup barrier on gfx950 does
// not wait for outstanding global stores, the stores of tile i drain while tile i+1 is being generated /
// loaded -- a fresh workgroup per tile instead holds its LDS and wave slots until its stores are acknowledged
// and then pays the table staging latency again.  XCD x owns the contiguous tile run [x, x+1) * ntiles/8 and
// its workgroups advance through it side by side, so tiles that share 128-byte lines meet in one L2.
template <class C, int DIR, class IO>
__global__ __launch_bounds__(C::NT, (col_min_waves<C, IO>())) void col_kernel_persistent(IO io, const cplx<typename C::T>* __restrict__ tw,
                                                                                      long long ntiles, int skip_period) {
  using F = ColFFT<C, DIR, IO>;
  using cx = cplx<typename C::T>;
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  const int tid = threadIdx.x;
  const cx* ltw = tw;
  io.bind_seed();
  if (F::HAS_PROLOGUE) {
    F::prologue(tid, io, tw, lds);
    if (C::NPASS >= 2) ltw = F::lds_tw(lds);
    __syncthreads();
  }
  const long long per_xcd = ntiles >> 3;                 // the launcher guarantees ntiles % 8 == 0, gridDim.x % 8 == 0
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  for (long long t = wl; t < per_xcd; t += wpx) {
    long long tile = (long long)xcd * per_xcd + t;
    if (skip_period > 0) {                               // all tiles except those = 0 mod skip_period
      const unsigned u = (unsigned)tile;
      tile = (long long)(u + u / (unsigned)(skip_period - 1) + 1u);
    }
    // everything a pass derives from the thread index is recomputed per tile: hoisting those values out of the
    // loop costs more registers than the kernel has (spills, whose reloads wait for the outstanding stores)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    __builtin_assume(tl >= 0 && tl < C::NT);
    F::pass_first(tl, tile, io, lds);
    if (C::NPASS == 3) {
      typename F::Regs r;
      __syncthreads();
      F::pass_mid_read(tl, ltw, lds, r);
      __syncthreads();
      F::pass_mid_write(tl, lds, r);
    }
    if (C::NPASS >= 2) {
      __syncthreads();
      F::pass_last(tl, tile, io, ltw, lds);
      __syncthreads();                                   // the tile image is free again
    }
  }
}

}  // namespace rf


// ---- intermediate-layout experiment: pairs of x rows interleaved at 64 B granularity, so that the 16 rows a wave
// stores with one instruction are 8 whole 128-byte lines:  (ix, iy, kz) -> ((ix/2)*ny + iy)*2*nzc + (kz/8)*16 + (ix%2)*8 + kz%8
struct IlvXGenIO : FastGenColIOT<0, 0, 0> {
  int nzc_, ny_;
  __device__ __forceinline__ void store(long long C0, int cl, int rb, int ro, const V16<float>& v) const {
    const long long iy = C0 / nzc_, kz0 = C0 % nzc_;
    cplx<float>* ub = base + ((long long)(ro >> 1) * ny_ + iy) * (2LL * nzc_) + (kz0 >> 3) * 16;
    const uint32_t lane = (uint32_t)(rb >> 1) * (uint32_t)(2 * ny_ * nzc_) + (uint32_t)(rb & 1) * 8u + (uint32_t)cl;
    v16_store<float>(reinterpret_cast<char*>(ub) + (size_t)(lane * 8u), v);
  }
};
// y pass reading that layout (tile = (ix, 8 kz)) and writing the standard layout into another buffer
struct IlvYIO : PlainColIO<float> {
  const cplx<float>* src;
  int nzc_, ny_;
  __device__ __forceinline__ V16<float> load(long long C0, int cl, int rb, int ro) const {
    const long long ix = C0 / nzc_, kz0 = C0 % nzc_;
    const cplx<float>* ub = src + ((ix >> 1) * ny_ + ro) * (2LL * nzc_) + (kz0 >> 3) * 16 + (ix & 1) * 8;
    const uint32_t lane = (uint32_t)rb * (uint32_t)(2 * nzc_) + (uint32_t)cl;
    return v16_load<float>(reinterpret_cast<const char*>(ub) + (size_t)(lane * 8u));
  }
};

// ---- x pass variants -----------------------------------------------------------------------------------
template <class C, class IO>
float bench_x(Timer& t, const IO& io, const cplx<float>* tw, long long ncols, int extra_lds, int persistent_grid) {
  const long long ntiles = ncols / C::TC;
  constexpr int lds0 = C::LDS_BYTES + IO::LDS_EXTRA;
  const int lds = lds0 + extra_lds;
  if (persistent_grid > 0) {
    auto k = col_kernel_persistent<C, +1, IO>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return t.run([&]() { hipLaunchKernelGGL(k, dim3((unsigned)persistent_grid), dim3(C::NT), lds, 0, io, tw, ntiles, 0); });
  }
  auto k = col_kernel<C, +1, IO>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return t.run([&]() { hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(C::NT), lds, 0, io, tw, ntiles, 1LL, 0LL, 0); });
}


// ---- co-residency probe: one persistent launch, two workgroups per CU; the first workgroup to arrive on a CU takes
// the VALU-bound role (generation + x-FFT tiles), the second a software-pipelined streaming copy (stand-in for a
// well-pipelined HBM-bound FFT pass).  Does max(VALU, HBM) hold, or the sum?
struct ProbeCtl { unsigned cu_count[1024]; unsigned x_next; unsigned c_next; };
template <class C, class IO>
__global__ __launch_bounds__(C::NT, 4) void probe_kernel(IO io, const cplx<float>* __restrict__ tw, long long ntiles,
                                                         vec4* __restrict__ cp, long long nchunks, ProbeCtl* ctl, int mode, int order) {
  using F = ColFFT<C, +1, IO>;
  using cx = cplx<float>;
  extern __shared__ __attribute__((aligned(16))) char rf_smem[];
  cx* lds = reinterpret_cast<cx*>(rf_smem);
  // the kernel's 80 KiB of dynamic LDS are exactly half a CU: no static LDS on top (it would halve the occupancy);
  // the two control words live in the unused tail of the sigma-record table (510 of 512 bins used)
  unsigned& s_role = reinterpret_cast<unsigned*>(rf_smem + C::LDS_BYTES + IO::LDS_EXTRA)[-1];
  unsigned& s_tile = reinterpret_cast<unsigned*>(rf_smem + C::LDS_BYTES + IO::LDS_EXTRA)[-2];
  const int tid = threadIdx.x;
  if (tid == 0) {
    unsigned r;
    if (mode == 0) r = atomicAdd(&ctl->cu_count[__smid() & 1023], 1u) & 1u;   // per-CU alternation
    else if (mode == 1) r = 0; else r = 1;                                     // all x / all copy
    s_role = r;
  }
  io.bind_seed();
  F::prologue(tid, io, tw, lds);
  __syncthreads();
  const unsigned role = s_role;
  const cx* ltw = F::lds_tw(lds);
  if (role == 0) {
    if (tid == 0) s_tile = atomicAdd(&ctl->x_next, 1u);       // rank among the x-role workgroups
    __syncthreads();
    const long long nxw = mode == 0 ? gridDim.x / 2 : gridDim.x;
    for (long long t = s_tile; t < ntiles; t += nxw) {
      // order 0: tile = t (adjacent tiles on different XCDs); 1: XCD-contiguous runs; 2: iy-major (concurrent tiles 4 KiB apart)
      long long tile = t;
      if (order == 1) { const long long per = ntiles >> 3; const int xcc = (int)(__smid() >> 6) & 7; const long long r = s_tile >> 3; tile = xcc * per + (t / nxw) * (nxw >> 3) + (r % (nxw >> 3)); }
      if (order == 2) tile = (t % 1024) * (ntiles / 1024) + t / 1024;
      int tl = tid;
      asm volatile("" : "+v"(tl));
      __builtin_assume(tl >= 0 && tl < C::NT);
      F::pass_first(tl, tile, io, lds);
      typename F::Regs r;
      __syncthreads();
      F::pass_mid_read(tl, ltw, lds, r);
      __syncthreads();
      F::pass_mid_write(tl, lds, r);
      __syncthreads();
      F::pass_last(tl, tile, io, ltw, lds);
      __syncthreads();
    }
  } else {
    // chunk = C::NT * 8 vec4 = 64 KiB; register double buffer
    constexpr int K = 8;
    vec4 cur[K], nxt[K];
    if (tid == 0) s_tile = atomicAdd(&ctl->c_next, 1u);       // rank among the copy-role workgroups
    __syncthreads();
    const long long ncw = mode == 0 ? gridDim.x / 2 : gridDim.x;
    long long c = s_tile;
    if (c < nchunks) {
#pragma unroll
      for (int k = 0; k < K; ++k) cur[k] = __builtin_nontemporal_load(cp + c * (C::NT * K) + k * C::NT + tid);
    }
    while (c < nchunks) {
      const long long cn = c + ncw;
      if (cn < nchunks) {
#pragma unroll
        for (int k = 0; k < K; ++k) nxt[k] = __builtin_nontemporal_load(cp + cn * (C::NT * K) + k * C::NT + tid);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) { cur[k].x += 1.0f; __builtin_nontemporal_store(cur[k], cp + c * (C::NT * K) + k * C::NT + tid); }
#pragma unroll
      for (int k = 0; k < K; ++k) cur[k] = nxt[k];
      c = cn;
    }
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const bool do_copy = argc < 2 || strchr(argv[1], 'c');
  const bool do_x = argc < 2 || strchr(argv[1], 'x');
  const bool do_y = argc < 2 || strchr(argv[1], 'y');
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
  Timer t;

  if (do_copy) {
    vec4* p = (vec4*)W;
    const long long n16 = ncplx / 2;
    for (int grid : {4096, 16384, 65536}) {
      char nm[128];
      snprintf(nm, sizeof nm, "linear r+w plain grid=%d", grid);
      report(nm, t.run([&]() { hipLaunchKernelGGL(lin_kernel<0>, dim3(grid), dim3(256), 0, 0, p, n16, 1, 1); }), 2 * sweep);
      snprintf(nm, sizeof nm, "linear r+w nontemporal grid=%d", grid);
      report(nm, t.run([&]() { hipLaunchKernelGGL(lin_kernel<1>, dim3(grid), dim3(256), 0, 0, p, n16, 1, 1); }), 2 * sweep);
    }
    report("linear read plain", t.run([&]() { hipLaunchKernelGGL(lin_kernel<0>, dim3(16384), dim3(256), 0, 0, p, n16, 1, 0); }), sweep);
    report("linear read nontemporal", t.run([&]() { hipLaunchKernelGGL(lin_kernel<1>, dim3(16384), dim3(256), 0, 0, p, n16, 1, 0); }), sweep);
    report("linear write plain", t.run([&]() { hipLaunchKernelGGL(lin_kernel<0>, dim3(16384), dim3(256), 0, 0, p, n16, 0, 1); }), sweep);
    report("linear write nontemporal", t.run([&]() { hipLaunchKernelGGL(lin_kernel<1>, dim3(16384), dim3(256), 0, 0, p, n16, 0, 1); }), sweep);
    {
      constexpr int CH = 2048;   // 32 KiB per chunk = 8 rows of the z pass
      const long long nch = n16 / CH;
      for (int grid : {2048, 8192, (int)nch}) {
        char nm[128];
        snprintf(nm, sizeof nm, "chunk(32KiB) r+w plain grid=%d", grid);
        report(nm, t.run([&]() { hipLaunchKernelGGL((chunk_kernel<0, CH>), dim3(grid), dim3(256), 0, 0, p, nch, 1, 1); }), 2 * sweep);
        snprintf(nm, sizeof nm, "chunk(32KiB) r+w nontemporal grid=%d", grid);
        report(nm, t.run([&]() { hipLaunchKernelGGL((chunk_kernel<1, CH>), dim3(grid), dim3(256), 0, 0, p, nch, 1, 1); }), 2 * sweep);
      }
    }
    const long long rs16 = nzc / 2, in16_y = nzc / 2, os16_y = (long long)NY * nzc / 2;
    const long long in16_x = (long long)NY * nzc / 2, rs16_x = (long long)NY * nzc / 2;
#define TILE(SEG, NT, RD, WR, MODE, label, in16, os16, rs, total16)                                              \
    {                                                                                                            \
      const long long ntiles = (total16) / (SEG / 16);                                                           \
      report(label, t.run([&]() { hipLaunchKernelGGL((tile_copy_kernel<SEG, NT, RD, WR, MODE>), dim3((unsigned)ntiles), dim3(NT), 0, 0, p, rs, in16, os16, 1024, ntiles); }), \
             ((RD ? 1 : 0) + (WR ? 1 : 0)) * sweep);                                                             \
    }
    TILE(64, 512, true, true, 0, "y-pattern tile copy r+w  64 B seg plain", in16_y, os16_y, rs16, (long long)NX * nzc / 2)
    TILE(64, 512, true, true, 1, "y-pattern tile copy r+w  64 B seg nontemporal", in16_y, os16_y, rs16, (long long)NX * nzc / 2)
    TILE(128, 512, true, true, 0, "y-pattern tile copy r+w 128 B seg plain", in16_y, os16_y, rs16, (long long)NX * nzc / 2)
    TILE(128, 512, true, true, 1, "y-pattern tile copy r+w 128 B seg nontemporal", in16_y, os16_y, rs16, (long long)NX * nzc / 2)
    TILE(64, 512, false, true, 0, "x-pattern tile write     64 B seg plain", in16_x, 0, rs16_x, (long long)NY * nzc / 2)
    TILE(64, 512, false, true, 1, "x-pattern tile write     64 B seg nontemporal", in16_x, 0, rs16_x, (long long)NY * nzc / 2)
    TILE(128, 512, false, true, 0, "x-pattern tile write    128 B seg plain", in16_x, 0, rs16_x, (long long)NY * nzc / 2)
    TILE(128, 512, false, true, 1, "x-pattern tile write    128 B seg nontemporal", in16_x, 0, rs16_x, (long long)NY * nzc / 2)
  }

  // generation tables
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

  if (argc >= 2 && strchr(argv[1], 's')) {
    // does the power-of-two row stride of the x pass (4 MiB) camp on memory channels?  pad it.
    vec4* big;
    CK(hipMalloc((void**)&big, ncplx * 8 + (size_t)1024 * (1 << 20)));
    CK(hipMemset(big, 0, ncplx * 8 + (size_t)1024 * (1 << 20)));
    const long long in16_x = (long long)NY * nzc / 2;
    {
      const long long nt2 = in16_x / 8;
      report("x-pattern write: both 64 B halves by the same thread, back to back", t.run([&]() { hipLaunchKernelGGL((pair_write_kernel<512, 0>), dim3((unsigned)nt2), dim3(512), 0, 0, big, in16_x, in16_x, 1024, nt2); }), sweep);
      report("x-pattern write: first halves of all rows, then second halves (same workgroup)", t.run([&]() { hipLaunchKernelGGL((pair_write_kernel<512, 1>), dim3((unsigned)nt2), dim3(512), 0, 0, big, in16_x, in16_x, 1024, nt2); }), sweep);
    }
    for (long long pad_bytes : {0LL, 1LL << 20}) if (0) {}
    for (long long pad_bytes : {0LL, 1LL << 20}) {
      const long long rs = (long long)NY * nzc / 2 + pad_bytes / 16;
      char nm[160];
#define TILES(SEG, RD, WR, what)                                                                                    \
      {                                                                                                           \
        const long long ntiles = in16_x / (SEG / 16);                                                             \
        snprintf(nm, sizeof nm, "x-pattern %s %3d B seg, row stride 4 MiB + %lld B", what, SEG, pad_bytes);       \
        report(nm, t.run([&]() { hipLaunchKernelGGL((tile_copy_kernel<SEG, 512, RD, WR, 0>), dim3((unsigned)ntiles), dim3(512), 0, 0, big, rs, in16_x, 0LL, 1024, ntiles); }), \
               ((RD ? 1 : 0) + (WR ? 1 : 0)) * sweep);                                                            \
      }
      TILES(64, false, true, "write")
      TILES(64, true, true, "r+w  ")
      TILES(128, false, true, "write")
    }
  }
  if (do_x) {
    using C = ColCfg<float, 1024, 8, 16, 8, 8, 512>;
    using IO = FastGenColIOT<0, 0, 0>;
    IO io; io.rec = nullptr; io.base = W; io.g = ColGeom{(long long)NY * nzc, 0, (long long)NY * nzc}; io.gp = fp; io.kz0 = 0; io.nzl = (int)nzc;
    const long long ncols = (long long)NY * nzc;
    report("x pass fast-gen fix=0: one workgroup per tile (baseline)", bench_x<C, IO>(t, io, tw, ncols, 0, 0), sweep);
    report("  same, LDS padded to 100 KiB (1 workgroup per CU)", bench_x<C, IO>(t, io, tw, ncols, 100 * 1024 - (C::LDS_BYTES + IO::LDS_EXTRA), 0), sweep);
    for (int grid : {256, 512, 1024, 2048, 8192})
    {
      char nm[128];
      snprintf(nm, sizeof nm, "  persistent, grid=%d", grid);
      report(nm, bench_x<C, IO>(t, io, tw, ncols, 0, grid), sweep);
    }
    report("  persistent grid=256, 1 workgroup per CU", bench_x<C, IO>(t, io, tw, ncols, 100 * 1024 - (C::LDS_BYTES + IO::LDS_EXTRA), 256), sweep);
    using IOA = FastGenColIOT<7, 0, 0>;
    IOA ioa; ioa.rec = nullptr; ioa.base = W; ioa.g = io.g; ioa.gp = fp; ioa.kz0 = 0; ioa.nzl = (int)nzc;
    report("  FFT + stores only (no Philox / sigma / Box-Muller): baseline", bench_x<C, IOA>(t, ioa, tw, ncols, 0, 0), sweep);
    report("  FFT + stores only: persistent grid=512", bench_x<C, IOA>(t, ioa, tw, ncols, 0, 512), sweep);
    {
      using CW = ColCfg<float, 1024, 8, 16, 8, 16, 1024>;
      report("x pass TC=16 (128 B rows), 1024 threads: one workgroup per tile", bench_x<CW, IO>(t, io, tw, ncols, 0, 0), sweep);
      for (int grid : {256, 512, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "x pass TC=16, 1024 threads: persistent grid=%d", grid);
        report(nm, bench_x<CW, IO>(t, io, tw, ncols, 0, grid), sweep);
      }
      using CW2 = ColCfg<float, 1024, 8, 8, 16, 16, 1024>;
      report("x pass TC=16 radix 8,8,16: one workgroup per tile", bench_x<CW2, IO>(t, io, tw, ncols, 0, 0), sweep);
      report("x pass TC=16 radix 8,8,16: persistent grid=256", bench_x<CW2, IO>(t, io, tw, ncols, 0, 256), sweep);
    }
    using C2 = ColCfg<float, 1024, 8, 8, 16, 8, 512>;
    report("x pass radix 8,8,16: baseline", bench_x<C2, IO>(t, io, tw, ncols, 0, 0), sweep);
    report("x pass radix 8,8,16: persistent grid=512", bench_x<C2, IO>(t, io, tw, ncols, 0, 512), sweep);
    using C3 = ColCfg<float, 1024, 4, 16, 16, 8, 512>;
    report("x pass radix 4,16,16: persistent grid=512", bench_x<C3, IO>(t, io, tw, ncols, 0, 512), sweep);
  }

  if (argc >= 2 && strchr(argv[1], 'p')) {
    using C = ColCfg<float, 1024, 8, 16, 8, 8, 512>;
    using IO = FastGenColIOT<0, 0, 0>;
    IO io; io.rec = nullptr; io.base = W; io.g = ColGeom{(long long)NY * nzc, 0, (long long)NY * nzc}; io.gp = fp; io.kz0 = 0; io.nzl = (int)nzc;
    const long long ntiles = (long long)NY * nzc / C::TC;
    cplx<float>* W2;
    CK(hipMalloc((void**)&W2, ncplx * 8));
    CK(hipMemset(W2, 0, ncplx * 8));
    ProbeCtl* ctl;
    CK(hipMalloc((void**)&ctl, sizeof(ProbeCtl)));
    CK(hipMemset(ctl, 0, sizeof(ProbeCtl)));
    auto k = probe_kernel<C, IO>;
    constexpr int lds = C::LDS_BYTES + IO::LDS_EXTRA;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    using IOS = FastGenColIOT<0, 0, 1>;
    IOS ios; ios.rec = nullptr; ios.base = W; ios.g = io.g; ios.gp = fp; ios.kz0 = 0; ios.nzl = (int)nzc; ios.x0 = 0; ios.x1 = 0;   // never stores
    auto ks = probe_kernel<C, IOS>;
    CK(hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const long long nchunks = (long long)(ncplx / 2) / (C::NT * 8);
    auto go = [&](int grid, int mode, long long nt, long long nc, int order) {
      CK(hipMemsetAsync(&ctl->x_next, 0, 8, 0));
      hipLaunchKernelGGL(k, dim3(grid), dim3(C::NT), lds, 0, io, tw, nt, (vec4*)W2, nc, ctl, mode, order);
    };
    auto gos = [&](int grid, int mode, long long nt, long long nc) {
      CK(hipMemsetAsync(&ctl->x_next, 0, 8, 0));
      hipLaunchKernelGGL(ks, dim3(grid), dim3(C::NT), lds, 0, ios, tw, nt, (vec4*)W2, nc, ctl, mode, 0);
    };
    cplx<float>* Wp;
    CK(hipMalloc((void**)&Wp, ncplx * 8 + (size_t)1024 * (1 << 20) + (1 << 20)));
    for (long long pad : {0LL, 4096LL + 256, 1LL << 20})
    for (int order : {1}) {
      io.base = Wp; io.g.row_stride = (long long)NY * nzc + pad / 8; io.g.inner = (long long)NY * nzc;
      printf("row stride 4 MiB + %lld B: ", pad);
      char nm[128];
      snprintf(nm, sizeof nm, "probe: x role only, 512 workgroups, tile order %d", order);
      report(nm, t.run([&]() { go(512, 1, ntiles, 0, order); }), sweep);
      snprintf(nm, sizeof nm, "probe: x role only, 256 workgroups, tile order %d", order);
      report(nm, t.run([&]() { go(256, 1, ntiles, 0, order); }), sweep);
      snprintf(nm, sizeof nm, "probe: x + copy co-resident, tile order %d", order);
      report(nm, t.run([&]() { go(512, 0, ntiles, nchunks, order); }), 3 * sweep);
    }
    report("probe: copy role only (r+w one sweep each), 512 workgroups", t.run([&]() { go(512, 2, 0, nchunks, 0); }), 2 * sweep);
    report("probe: copy role only, 256 workgroups (1 per CU)", t.run([&]() { go(256, 2, 0, nchunks, 0); }), 2 * sweep);
    report("probe: x role WITHOUT stores, 512 workgroups", t.run([&]() { gos(512, 1, ntiles, 0); }), sweep);
    report("probe: x role WITHOUT stores, 256 workgroups", t.run([&]() { gos(256, 1, ntiles, 0); }), sweep);
    report("probe: x WITHOUT stores + copy co-resident", t.run([&]() { gos(512, 0, ntiles, nchunks); }), 2 * sweep);
    unsigned h[1024];
    CK(hipMemcpy(h, ctl->cu_count, sizeof h, hipMemcpyDeviceToHost));
    int used = 0, odd = 0; unsigned mx = 0, mn = ~0u;
    for (int i = 0; i < 1024; ++i) if (h[i]) { ++used; odd += h[i] & 1; mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
    printf("CU ids seen: %d, counts min %u max %u, odd counts %d\n", used, mn, mx, odd);
  }

  if (argc >= 2 && strchr(argv[1], 'm')) {
    vec4* S;
    const size_t smax = (size_t)512 * 8 * 65536;
    CK(hipMalloc((void**)&S, smax));
    CK(hipMemset(S, 0, smax));
    const long long nchunks = (long long)(ncplx / 2) / (512 * 8);
    report("cache probe: plain streaming r+w, 512 persistent workgroups", t.run([&]() { hipLaunchKernelGGL(mall_probe_kernel<0>, dim3(512), dim3(512), 0, 0, (vec4*)W, nchunks, S, 1); }), 2 * sweep);
    for (int D : {1, 2, 3, 4, 8}) {
      char nm[160];
      snprintf(nm, sizeof nm, "cache probe: + bounce through a %d MiB ring (written, re-read by another workgroup)", 32 * D);
      report(nm, t.run([&]() { hipLaunchKernelGGL(mall_probe_kernel<1>, dim3(512), dim3(512), 0, 0, (vec4*)W, nchunks, S, D); }), 2 * sweep);
    }
  }


  if (argc >= 2 && strchr(argv[1], 'L')) {
    // length-2048 x pass (one 152 KiB tile per CU): fresh workgroup per tile vs persistent workgroups
    const int NXL = 2048;
    cplx<float>* WL;
    CK(hipMalloc((void**)&WL, (size_t)NXL * NY * nzc * 8));
    auto tw2h = make_twiddles<float>(2048);
    cplx<float>* tw2;
    CK(hipMalloc((void**)&tw2, tw2h.size() * 8));
    CK(hipMemcpy(tw2, tw2h.data(), tw2h.size() * 8, hipMemcpyHostToDevice));
    using C = ColCfg<float, 2048, 8, 16, 16, 8, 1024>;
    using IO = FastGenColIOT<0, 0, 0>;
    FastGenParams fpl = fp; fpl.nx = NXL;
    IO io; io.rec = nullptr; io.base = WL; io.g = ColGeom{(long long)NY * nzc, 0, (long long)NY * nzc}; io.gp = fpl; io.kz0 = 0; io.nzl = (int)nzc;
    const long long ncols = (long long)NY * nzc;
    const double sw = (double)NXL * NY * nzc * 8;
    report("x pass N=2048: one workgroup per tile", bench_x<C, IO>(t, io, tw2, ncols, 0, 0), sw);
    for (int grid : {256, 512, 1024, 4096}) {
      char nm[128];
      snprintf(nm, sizeof nm, "x pass N=2048: persistent grid=%d", grid);
      report(nm, bench_x<C, IO>(t, io, tw2, ncols, 0, grid), sw);
    }
    using CP = ColCfg<float, 2048, 8, 16, 16, 8, 1024>;
    PlainColIO<float> yio; yio.base = WL; yio.g = ColGeom{nzc, (long long)NY * nzc, nzc};   // y-geometry with 2048 rows: treat as [1024 planes][2048 iy]
    yio.g = ColGeom{nzc, (long long)2048 * nzc, nzc};
    const long long ycols = (long long)1024 * nzc;
    report("y pass N=2048: one workgroup per tile", bench_x<CP, PlainColIO<float>>(t, yio, tw2, ycols, 0, 0), 2 * sw);
    report("y pass N=2048: persistent grid=256", bench_x<CP, PlainColIO<float>>(t, yio, tw2, ycols, 0, 256), 2 * sw);
  }

  if (argc >= 2 && strchr(argv[1], 'i')) {
    using C = ColCfg<float, 1024, 8, 16, 8, 8, 512>;
    using IO = FastGenColIOT<0, 0, 0>;
    IO io; io.rec = nullptr; io.base = W; io.g = ColGeom{(long long)NY * nzc, 0, (long long)NY * nzc}; io.gp = fp; io.kz0 = 0; io.nzl = (int)nzc;
    const long long ncols = (long long)NY * nzc;
    cplx<float>* W2;
    CK(hipMalloc((void**)&W2, ncplx * 8));
    CK(hipMemset(W2, 0, ncplx * 8));
    IlvXGenIO xi; static_cast<IO&>(xi) = io; xi.nzc_ = (int)nzc; xi.ny_ = NY;
    for (int rep = 0; rep < 2; ++rep) {
      report("x pass, standard layout (half-line stores)", bench_x<C, IO>(t, io, tw, ncols, 0, 0), sweep);
      report("x pass, x-pair interleaved layout (whole-line stores)", bench_x<C, IlvXGenIO>(t, xi, tw, ncols, 0, 0), sweep);
    }
    using CY = ColCfg<float, 1024, 16, 8, 8, 8, 512>;
    PlainColIO<float> yio; yio.base = W; yio.g = ColGeom{nzc, (long long)NY * nzc, nzc};
    IlvYIO yi; static_cast<PlainColIO<float>&>(yi) = yio; yi.base = W2; yi.src = W; yi.nzc_ = (int)nzc; yi.ny_ = NY;
    for (int rep = 0; rep < 2; ++rep) {
      report("y pass in place, standard layout", bench_x<CY, PlainColIO<float>>(t, yio, tw, (long long)NX * nzc, 0, 0), 2 * sweep);
      report("y pass reading the interleaved layout, writing the standard one (out of place)", bench_x<CY, IlvYIO>(t, yi, tw, (long long)NX * nzc, 0, 0), 2 * sweep);
    }
  }

  if (do_y) {
    using C = ColCfg<float, 1024, 16, 8, 8, 8, 512>;
    using IO = PlainColIO<float>;
    IO io; io.base = W; io.g = ColGeom{nzc, (long long)NY * nzc, nzc};
    const long long ncols = (long long)NX * nzc;
    report("y pass 16,8,8: one workgroup per tile (baseline)", bench_x<C, IO>(t, io, tw, ncols, 0, 0), 2 * sweep);
    {
      using C8 = ColCfg<float, 1024, 8, 16, 8, 8, 512>;
      report("y pass 8,16,8: one workgroup per tile", bench_x<C8, IO>(t, io, tw, ncols, 0, 0), 2 * sweep);
    }
    for (int grid : {512, 1024})
    {
      char nm[128];
      snprintf(nm, sizeof nm, "  persistent, grid=%d", grid);
      report(nm, bench_x<C, IO>(t, io, tw, ncols, 0, grid), 2 * sweep);
    }
  }
  return 0;
}
