// valu_rate.hip -- VALU issue-rate microbenchmark (development tool): scalar f32 FMA vs packed v_pk_fma_f32,
// quarter-rate v_mad_u64_u32, transcendental ops, at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
  unsigned long long m0 = threadIdx.x + 1, m1 = m0 + 7;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {        // 8 independent scalar FMAs
      x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
      x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b;
    } else if (MODE == 1) { // 4 independent packed FMAs (same 8 flops-pairs)
      p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
      p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
    } else if (MODE == 2) { // 2 x 32x32->64 products
      m0 = (unsigned long long)(unsigned)m0 * 0xD2511F53u + (m1 >> 32);
      m1 = (unsigned long long)(unsigned)m1 * 0xCD9E8D57u + (m0 >> 32);
    } else if (MODE == 3) { // transcendental: log + sin
      x0 = __builtin_amdgcn_logf(x0 + 1.5f); x1 = __builtin_amdgcn_sinf(x1); x2 = __builtin_amdgcn_logf(x2 + 1.5f); x3 = __builtin_amdgcn_sinf(x3);
    }
  }
  float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(m0 + m1);
  if (r == 123.456f) out[0] = r;
}

template <int MODE> int run(const char* name, int ops_per_iter, float* d) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int wps : {1, 2, 4}) {               // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD = wps * iters * ops_per_iter ; cycles at 2.4 GHz nominal
    double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * ops_per_iter);
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles (at 2.4 GHz) per wave-instruction per SIMD\n", name, wps, ms, cyc);
  }
  return 0;
}

int main() {
  float* d; CK(hipMalloc((void**)&d, 4096));
  run<0>("v_fma_f32 (8 indep)", 8, d);
  run<1>("v_pk_fma_f32 (4 indep)", 4, d);
  run<2>("v_mad_u64_u32 (2, chained)", 2, d);
  run<3>("v_log/v_sin (4 indep)", 4, d);
  return 0;
}
