// dep_rate.hip -- issue rate of DEPENDENT VALU chains on gfx950 as a function of the chains per wave (ILP) and the
// waves per SIMD: how much (ILP x occupancy) a VALU-bound kernel needs before the SIMD stops idling (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X

#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %4, %5\n"
#define A_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %4, %5 bitop3:0x96\n"
#define A_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %4\n"

#define KERNEL(NAME, BODY)                                                                               \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters, float a, float b) {                  \
    float x[4];                                                                                            \
    for (int i = 0; i < 4; ++i) x[i] = threadIdx.x * 1e-3f + i + 1.5f;                                     \
    for (int it = 0; it < iters; ++it) {                                                                   \
      REP16(asm volatile(BODY : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(a), "v"(b));)         \
    }                                                                                                      \
    float r = x[0] + x[1] + x[2] + x[3];                                                                   \
    if (r == 123.456f) out[0] = r;                                                                         \
  }
KERNEL(fma1, A_FMA(0) A_FMA(0) A_FMA(0) A_FMA(0))
KERNEL(fma2, A_FMA(0) A_FMA(1) A_FMA(0) A_FMA(1))
KERNEL(fma4, A_FMA(0) A_FMA(1) A_FMA(2) A_FMA(3))
KERNEL(bit1, A_BITOP3(0) A_BITOP3(0) A_BITOP3(0) A_BITOP3(0))
KERNEL(bit2, A_BITOP3(0) A_BITOP3(1) A_BITOP3(0) A_BITOP3(1))
KERNEL(bit4, A_BITOP3(0) A_BITOP3(1) A_BITOP3(2) A_BITOP3(3))
KERNEL(log1, A_LOG(0) A_LOG(0) A_LOG(0) A_LOG(0))
KERNEL(log2, A_LOG(0) A_LOG(1) A_LOG(0) A_LOG(1))
KERNEL(log4, A_LOG(0) A_LOG(1) A_LOG(2) A_LOG(3))
KERNEL(mul1, A_MULLO(0) A_MULLO(0) A_MULLO(0) A_MULLO(0))
KERNEL(mul2, A_MULLO(0) A_MULLO(1) A_MULLO(0) A_MULLO(1))
KERNEL(mul4, A_MULLO(0) A_MULLO(1) A_MULLO(2) A_MULLO(3))
// mixed like a Philox round: quarter-rate multiply -> 3-input xor -> multiply ...: one chain / two chains per wave
__global__ __launch_bounds__(256) void philox1(float* out, int iters, float a, float b) {
  unsigned c = threadIdx.x * 7u, ka = __float_as_uint(a), kb = __float_as_uint(b);
  for (int it = 0; it < iters; ++it) {
    REP16(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n"
                       "v_mul_hi_u32 %0, %0, %1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n" : "+v"(c) : "v"(ka), "v"(kb));)
  }
  if (c == 12345u) out[0] = (float)c;
}
__global__ __launch_bounds__(256) void philox2(float* out, int iters, float a, float b) {
  unsigned c = threadIdx.x * 7u, d = threadIdx.x * 5u, ka = __float_as_uint(a), kb = __float_as_uint(b);
  for (int it = 0; it < iters; ++it) {
    REP16(asm volatile("v_mul_hi_u32 %0, %0, %2\n v_mul_hi_u32 %1, %1, %2\n v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n"
                       : "+v"(c), "+v"(d) : "v"(ka), "v"(kb));)
  }
  if (c + d == 12345u) out[0] = (float)(c + d);
}

typedef void (*kern_t)(float*, int, float, float);
static int run(const char* name, kern_t k, float* d, int instr_per_rep) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  printf("%-10s", name);
  for (int wps : {1, 2, 4, 6, 8}) {
    hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * 16 * instr_per_rep);   // per wave-instruction per SIMD
    printf("  w=%d: %6.2f", wps, cyc);
  }
  printf("\n");
  return 0;
}
int main() {
  float* d; CK(hipMalloc((void**)&d, 4096));
  printf("nominal 2.4 GHz cycles per wave64 instruction per SIMD; name = op + independent chains per wave; w = waves per SIMD\n");
#define RUN(k) run(#k, k, d, 4);
  RUN(fma1) RUN(fma2) RUN(fma4) RUN(bit1) RUN(bit2) RUN(bit4) RUN(mul1) RUN(mul2) RUN(mul4) RUN(log1) RUN(log2) RUN(log4)
  RUN(philox1) RUN(philox2)
  return 0;
}
