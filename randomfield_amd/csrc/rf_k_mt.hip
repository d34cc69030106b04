// rf_k_mt.hip -- on-GPU replay of numpy.random.RandomState(seed).normal(size = 2 M): MT19937 + the
// legacy polar method, i.e. the stream randomfield.random.randomize draws (random.py:24-28).
//
//   value[2c] = f * x2, value[2c+1] = f * x1 of the c-th ACCEPTED polar attempt; attempt a consumes the
//   four tempered outputs 4a .. 4a+3:  u = ((w >> 5) * 2^26 + (w' >> 6)) / 2^53, x = 2u - 1,
//   r2 = x1^2 + x2^2, accepted iff 0 < r2 < 1, f = sqrt(-2 ln r2 / r2).
//
// Parallelisation (randomfield_amd/mt19937.py has the mathematics):
//   1. jump tree: the state J words ahead is the XOR of the sequence words x_{i+j} over the set
//      coefficients j of t^J mod phi(t).  mt_jump_kernel builds 33 blocks of the sequence of a source state in
//      LDS and XORs them; level k of the tree doubles the number of segment start states.
//   2. mt_polar_kernel<false>: every segment (1024 blocks of 624 outputs, one WAVE each) counts its accepted attempts;
//      an exclusive scan of the counts gives each segment the index of its first cell;
//   3. mt_polar_kernel<true>: the same generation again, now writing the deviates of the accepted attempts.
#include <hip/hip_runtime.h>
#include "rf_launch.h"
#include "rf_core.h"

namespace rf {
namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_SEQ_BLOCKS = 33;                  // 33 * 624 = 20592 >= 19937 + 624 sequence words per source
constexpr int MT_SEQ_WORDS = MT_SEQ_BLOCKS * MT_N;

__device__ __forceinline__ uint32_t mt_f(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7FFFFFFFu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}

// regenerate: nxt = the 624 words that follow cur (blockDim.x >= 227 threads; ends with a barrier)
__device__ __forceinline__ void mt_next_block(const uint32_t* cur, uint32_t* nxt) {
  const int t = threadIdx.x;
  if (t < 227) nxt[t] = mt_f(cur[t], cur[t + 1], cur[t + MT_M]);
  __syncthreads();
  if (t < 227) nxt[t + 227] = mt_f(cur[t + 227], cur[t + 228], nxt[t]);
  __syncthreads();
  if (t < 169) nxt[t + 454] = mt_f(cur[t + 454], cur[t + 455], nxt[t + 227]);
  __syncthreads();
  if (t == 0) nxt[623] = mt_f(cur[623], nxt[0], nxt[396]);
  __syncthreads();
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9D2C5680u;
  y ^= (y << 15) & 0xEFC60000u;
  y ^= y >> 18;
  return y;
}

// One jump: destination state d = source state d advanced by the level's distance.  One workgroup per destination:
// the 33-block sequence window of the source (82 KB) is generated into LDS, then lane i XORs the window words
// i + pos[j] over the set coefficients j of the jump polynomial (~10^4 conflict-free LDS reads per lane instead
// of as many L2 reads: the global-memory version of this step took 13 of the replay's 28 ms).
__global__ __launch_bounds__(640) void mt_jump_kernel(const uint32_t* __restrict__ states_src, const uint32_t* __restrict__ pos,
                                                      int npos, uint32_t* __restrict__ states_dst) {
  extern __shared__ __attribute__((aligned(16))) uint32_t win[];     // MT_SEQ_WORDS words
  const int t = threadIdx.x;
  const uint32_t* st = states_src + (size_t)blockIdx.x * MT_N;
  if (t < MT_N) win[t] = st[t];
  __syncthreads();
  // x[n + 624] = f(x[n], x[n + 1], x[n + 397]): within a block of 624 new words, words [0, 227) need old words only,
  // [227, 454) need new words [0, 227), [454, 624) need new words [227, 397)
  for (int b = 1; b < MT_SEQ_BLOCKS; ++b) {
    uint32_t* nw = win + b * MT_N;
    const uint32_t* od = nw - MT_N;
    if (t < 227) nw[t] = mt_f(od[t], od[t + 1], od[t + MT_M]);
    __syncthreads();
    if (t >= 227 && t < 454) nw[t] = mt_f(od[t], od[t + 1], od[t + MT_M]);
    __syncthreads();
    if (t >= 454 && t < MT_N) nw[t] = mt_f(od[t], od[t + 1], od[t + MT_M]);
    __syncthreads();
  }
  if (t < MT_N) {
    uint32_t acc = 0;
    const uint32_t* w = win + t;
    int j = 0;                                     // pos[] is uniform: scalar loads, 8 positions per s_load_dwordx8
    for (; j + 8 <= npos; j += 8)
      acc ^= w[pos[j]] ^ w[pos[j + 1]] ^ w[pos[j + 2]] ^ w[pos[j + 3]] ^ w[pos[j + 4]] ^ w[pos[j + 5]] ^ w[pos[j + 6]] ^ w[pos[j + 7]];
    for (; j < npos; ++j) acc ^= w[pos[j]];
    states_dst[(size_t)blockIdx.x * MT_N + t] = acc;
  }
}

// One WAVE per segment (4 segments per 256-thread workgroup), no workgroup barriers: the 624-word state lives in the
// wave's own LDS window and is regenerated IN PLACE in 64-lane chunks -- word i needs the old words i, i+1 and either
// the old word i+397 (i < 227) or the new word i-227, which an earlier chunk has already written (LDS operations of
// one wave execute in order; `volatile` keeps the compiler from reordering them).
// FILL = false: counts[seg] = accepted attempts of the segment.
// FILL = true: writes (f x2, f x1) of every accepted attempt whose cell index < ncells.
template <bool FILL>
__global__ __launch_bounds__(256) void mt_polar_kernel(const uint32_t* __restrict__ states, int blocks_per_segment,
                                                       long long total_blocks, int nseg, unsigned long long* __restrict__ counts,
                                                       const unsigned long long* __restrict__ offsets,
                                                       double* __restrict__ noise, unsigned long long ncells) {
  __shared__ uint32_t lds[4][MT_N + 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long seg = (long long)blockIdx.x * 4 + wave;
  if (seg >= nseg) return;                                   // whole waves leave; nobody waits at a barrier
  volatile uint32_t* mt = lds[wave];
  long long nb = total_blocks - seg * blocks_per_segment;
  if (nb > blocks_per_segment) nb = blocks_per_segment;
  const uint32_t* st = states + (size_t)seg * MT_N;
  for (int i = lane; i < MT_N; i += 64) mt[i] = st[i];
  unsigned long long running = FILL ? offsets[seg] : 0ull;   // cell index of this segment's next accepted attempt
  for (long long b = 0; b < nb; ++b) {
    // regenerate: the outputs of this block are the tempered NEW words
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int i = 64 * k + lane;
      if (i < MT_N) {
        // word 623 follows the same rule with its neighbour wrapped to the (new) word 0: 623 - 227 = 396
        const uint32_t third = i < MT_N - MT_M ? mt[i + MT_M] : mt[i - (MT_N - MT_M)];
        const uint32_t nxt = mt[i + 1 == MT_N ? 0 : i + 1];
        const uint32_t v = mt_f(mt[i], nxt, third);
        mt[i] = v;                                           // every lane has read before any lane writes (lock step)
      }
    }
    // polar method: attempt a uses outputs 4a .. 4a+3; 156 attempts per block, in lane order
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int a = 64 * it + lane;
      bool acc = false;
      double x1 = 0, x2 = 0, r2 = 0;
      if (a < MT_N / 4) {
        const uint32_t w0 = mt_temper(mt[4 * a]), w1 = mt_temper(mt[4 * a + 1]);
        const uint32_t w2 = mt_temper(mt[4 * a + 2]), w3 = mt_temper(mt[4 * a + 3]);
        const double u1 = ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
        const double u2 = ((double)(w2 >> 5) * 67108864.0 + (double)(w3 >> 6)) / 9007199254740992.0;
        x1 = 2.0 * u1 - 1.0;
        x2 = 2.0 * u2 - 1.0;
        r2 = sum_of_squares(x1, x2);                          // no FMA: numpy's C code rounds both products
        acc = (r2 < 1.0) && (r2 != 0.0);
      }
      const unsigned long long ball = __ballot(acc);
      if (FILL && acc) {
        const unsigned long long cell = running + (unsigned long long)__popcll(ball & ((1ull << lane) - 1ull));
        if (cell < ncells) {
          const double f = sqrt(-2.0 * log(r2) / r2);
          noise[2 * cell] = f * x2;                           // legacy_gauss returns f*x2 first, then the saved f*x1
          noise[2 * cell + 1] = f * x1;
        }
      }
      running += (unsigned long long)__popcll(ball);
    }
  }
  if (!FILL && lane == 0) counts[seg] = running;
}

// exclusive scan of the per-segment counts by ONE wave: lane l owns a contiguous chunk, wave scan across lanes
__global__ __launch_bounds__(64) void mt_scan_kernel(const unsigned long long* __restrict__ counts, unsigned long long* __restrict__ offsets, int n) {
  const int lane = threadIdx.x;
  const int per = (n + 63) / 64, lo = lane * per, hi = lo + per < n ? lo + per : n;
  unsigned long long sum = 0;
  for (int i = lo; i < hi; ++i) sum += counts[i];
  unsigned long long inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  unsigned long long run = inc - sum;
  for (int i = lo; i < hi; ++i) { offsets[i] = run; run += counts[i]; }
  if (lane == 63) offsets[n] = inc;                    // total number of accepted attempts
}

}  // namespace

hipError_t launch_mt_jump(const uint32_t* states_src, const uint32_t* pos, int npos, uint32_t* states_dst, int ndst, hipStream_t s) {
  constexpr int lds = MT_SEQ_WORDS * (int)sizeof(uint32_t);
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)mt_jump_kernel, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(mt_jump_kernel, dim3(ndst), dim3(640), lds, s, states_src, pos, npos, states_dst);
  return hipGetLastError();
}
hipError_t launch_mt_polar(bool fill, const uint32_t* states, int nseg, int blocks_per_segment, long long total_blocks,
                           unsigned long long* counts, const unsigned long long* offsets, double* noise,
                           unsigned long long ncells, hipStream_t s) {
  const unsigned grid = (unsigned)((nseg + 3) / 4);
  if (fill) hipLaunchKernelGGL(mt_polar_kernel<true>, dim3(grid), dim3(256), 0, s, states, blocks_per_segment, total_blocks, nseg, counts, offsets, noise, ncells);
  else hipLaunchKernelGGL(mt_polar_kernel<false>, dim3(grid), dim3(256), 0, s, states, blocks_per_segment, total_blocks, nseg, counts, offsets, noise, ncells);
  return hipGetLastError();
}
hipError_t launch_mt_scan(const unsigned long long* counts, unsigned long long* offsets, int n, hipStream_t s) {
  hipLaunchKernelGGL(mt_scan_kernel, dim3(1), dim3(64), 0, s, counts, offsets, n);
  return hipGetLastError();
}

}  // namespace rf
