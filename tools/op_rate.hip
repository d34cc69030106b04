// op_rate.hip -- per-opcode VALU throughput on gfx950 (development tool): 8 independent chains of one
// instruction, 16x unrolled, 4 and 8 waves per SIMD; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define REP16(X) X X X X X X X X X X X X X X X X

#define KERNEL32(NAME, ASM)                                                                          \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters, float a, float b) {              \
    float x[8];                                                                                        \
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i + 1.5f;                                 \
    for (int it = 0; it < iters; ++it) {                                                               \
      REP16(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                       \
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));) \
    }                                                                                                  \
    float r = 0; for (int i = 0; i < 8; ++i) r += x[i];                                                \
    if (r == 123.456f) out[0] = r;                                                                     \
  }
#define KERNEL64(NAME, ASM)                                                                          \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters, float a, float b) {              \
    double x[8]; double da = a, db = b;                                                                \
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i + 1.5;                                   \
    for (int it = 0; it < iters; ++it) {                                                               \
      REP16(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                       \
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(da), "v"(db));) \
    }                                                                                                  \
    double r = 0; for (int i = 0; i < 8; ++i) r += x[i];                                               \
    if (r == 123.456) out[0] = (float)r;                                                               \
  }

#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_ADDF(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_MULHI24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define A_SIN(i) "v_sin_f32 %" #i ", %" #i "\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define A_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_CVTIF(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define A_FRACT(i) "v_fract_f32 %" #i ", %" #i "\n"
#define A_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
// 64-bit register pairs
#define A_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define A_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define A_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define A_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define A_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define A_MADU64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
#define A_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 1, %8\n"
#define A_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %" #i "\n"
#define A_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define A_CVT64(i) "v_cvt_f64_u32 %" #i ", %8\n"

KERNEL32(k_fma, A_FMA) KERNEL32(k_addf, A_ADDF) KERNEL32(k_mulf, A_MULF) KERNEL32(k_addu, A_ADDU) KERNEL32(k_xor, A_XOR)
KERNEL32(k_bitop3, A_BITOP3) KERNEL32(k_lshladd, A_LSHLADD) KERNEL32(k_mullo, A_MULLO) KERNEL32(k_mulhi, A_MULHI)
KERNEL32(k_mul24, A_MUL24) KERNEL32(k_mulhi24, A_MULHI24) KERNEL32(k_mad24, A_MAD24) KERNEL32(k_log, A_LOG) KERNEL32(k_sin, A_SIN)
KERNEL32(k_sqrt, A_SQRT) KERNEL32(k_rcp, A_RCP) KERNEL32(k_exp, A_EXP) KERNEL32(k_cvtfu, A_CVTFU) KERNEL32(k_cvtif, A_CVTIF)
KERNEL32(k_fract, A_FRACT) KERNEL32(k_med3, A_MED3) KERNEL32(k_mov, A_MOV) KERNEL32(k_alignbit, A_ALIGNBIT) KERNEL32(k_perm, A_PERM)
KERNEL64(k_fma64, A_FMA64) KERNEL64(k_add64, A_ADD64) KERNEL64(k_mul64, A_MUL64) KERNEL64(k_pkfma, A_PKFMA) KERNEL64(k_pkadd, A_PKADD)
KERNEL64(k_pkmul, A_PKMUL) KERNEL64(k_lshladd64, A_LSHLADD64) KERNEL64(k_lshl64, A_LSHL64)
KERNEL64(k_rcp64, A_RCP64)

__global__ __launch_bounds__(256) void k_madu64(float* out, int iters, float a, float b) {
  unsigned long long x[8];
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
    REP16(asm volatile(A_MADU64(0) A_MADU64(1) A_MADU64(2) A_MADU64(3) A_MADU64(4) A_MADU64(5) A_MADU64(6) A_MADU64(7)
          : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(ua), "v"(ub) : "vcc");)
  }
  unsigned long long r = 0; for (int i = 0; i < 8; ++i) r += x[i];
  if (r == 123456ull) out[0] = (float)r;
}

typedef void (*kern_t)(float*, int, float, float);
static int run(const char* name, kern_t k, float* d) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  printf("%-22s", name);
  for (int wps : {1, 4, 8}) {               // waves per SIMD: a block of 256 threads = 1 wave on each SIMD of a CU
    hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * 128);
    printf("  w/SIMD=%d: %6.2f cyc", wps, cyc);
  }
  printf("\n");
  return 0;
}
#define RUN(k) run(#k, k, d);
int main() {
  float* d; CK(hipMalloc((void**)&d, 4096));
  printf("cycles (at 2.4 GHz nominal) per wave64 instruction per SIMD\n");
  RUN(k_fma) RUN(k_addf) RUN(k_mulf) RUN(k_pkfma) RUN(k_pkadd) RUN(k_pkmul) RUN(k_addu) RUN(k_xor) RUN(k_bitop3) RUN(k_lshladd) RUN(k_mov)
  RUN(k_alignbit) RUN(k_perm) RUN(k_med3) RUN(k_fract) RUN(k_cvtfu) RUN(k_cvtif)
  RUN(k_mullo) RUN(k_mulhi) RUN(k_mul24) RUN(k_mulhi24) RUN(k_mad24) RUN(k_madu64) RUN(k_lshladd64) RUN(k_lshl64)
  RUN(k_log) RUN(k_sin) RUN(k_sqrt) RUN(k_rcp) RUN(k_exp)
  RUN(k_fma64) RUN(k_add64) RUN(k_mul64) RUN(k_rcp64)
  return 0;
}
