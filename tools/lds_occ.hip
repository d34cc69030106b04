// how many 512-thread workgroups with a given dynamic LDS size run at once on a CU? (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void spin(float* out, int iters) {
  extern __shared__ float lds[];
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0000001f + 0.5f;
  lds[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[1];
}
int main() {
  float* out; hipMalloc(&out, 4096 * 4);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const int sizes[] = {40960, 65536, 80000, 81408, 81920, 82432, 90112, 163840};
  for (int s : sizes) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(spin, dim3(512), dim3(512), s, 0, out, 200000);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(spin, dim3(512), dim3(512), s, 0, out, 200000);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("lds %6d B: %.3f ms (%s)\n", s, ms, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
