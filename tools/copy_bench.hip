// copy_bench.hip -- what read+write bandwidth can a streaming kernel reach on this MI355X? (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int U, bool INPLACE>
__global__ __launch_bounds__(256) void copyk(const float4* __restrict__ src, float4* __restrict__ dst, long long n16) {
  // each block handles a contiguous chunk of U * 256 float4; U loads in flight per thread
  long long base = (long long)blockIdx.x * (U * 256) + threadIdx.x;
  const long long stride = (long long)gridDim.x * (U * 256);
  for (; base < n16; base += stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u].x += 1.0f; dst[base + u * 256] = v[u]; }
  }
}

template <int U, bool INPLACE> int run(float4* a, float4* b, long long n16, int blocks_per_cu) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long long maxblocks = n16 / (U * 256);
  long long grid = blocks_per_cu > 0 ? 256LL * blocks_per_cu : maxblocks;
  if (grid > maxblocks) grid = maxblocks;
  float4* d = INPLACE ? a : b;
  hipLaunchKernelGGL((copyk<U, INPLACE>), dim3((unsigned)grid), dim3(256), 0, 0, a, d, n16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((copyk<U, INPLACE>), dim3((unsigned)grid), dim3(256), 0, 0, a, d, n16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("U=%d %s grid=%8lld  %.3f ms  %.1f GB/s (read+write)\n", U, INPLACE ? "in-place " : "out-place", grid, ms, 2.0 * n16 * 16 / ms / 1e6);
  return 0;
}

int main() {
  const long long n16 = 4294967296LL / 16;   // 4.29 GB, the 1024^3 field
  float4 *a, *b;
  CK(hipMalloc((void**)&a, n16 * 16)); CK(hipMalloc((void**)&b, n16 * 16));
  CK(hipMemset(a, 0, n16 * 16)); CK(hipMemset(b, 0, n16 * 16));
  run<1, true>(a, b, n16, 16); run<4, true>(a, b, n16, 16); run<8, true>(a, b, n16, 16); run<8, true>(a, b, n16, 0);
  run<4, true>(a, b, n16, 0); run<16, true>(a, b, n16, 0); run<16, true>(a, b, n16, 8);
  run<4, false>(a, b, n16, 16); run<8, false>(a, b, n16, 0); run<16, false>(a, b, n16, 0);
  hipMemcpyAsync(b, a, n16 * 16, hipMemcpyDeviceToDevice, 0); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) hipMemcpyAsync(b, a, n16 * 16, hipMemcpyDeviceToDevice, 0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("hipMemcpy D2D          %.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * n16 * 16 / ms / 1e6);
  return 0;
}
