// rf_k_mt.hip -- on-GPU replay of numpy.random.RandomState(seed).normal(size = 2 M): MT19937 + the
// legacy polar method, i.e. the stream randomfield.random.randomize draws (random.py:24-28).
//
//   value[2c] = f * x2, value[2c+1] = f * x1 of the c-th ACCEPTED polar attempt; attempt a consumes the
//   four tempered outputs 4a .. 4a+3:  u = ((w >> 5) * 2^26 + (w' >> 6)) / 2^53, x = 2u - 1,
//   r2 = x1^2 + x2^2, accepted iff 0 < r2 < 1, f = sqrt(-2 ln r2 / r2).
//
// Parallelisation (randomfield_amd/mt19937.py has the mathematics):
//   1. jump tree: the state J words ahead is the XOR of the sequence words x_{i+j} over the set
//      coefficients j of t^J mod phi(t).  mt_expand_kernel writes 33 blocks of the sequence of each source
//      state, mt_combine_kernel XORs them; level k of the tree doubles the number of segment start states.
//   2. mt_polar_kernel<false>: every segment (1024 blocks of 624 outputs) counts its accepted attempts;
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

// sequence words x_p .. x_{p + MT_SEQ_WORDS - 1} of source state `src` (one workgroup per source)
__global__ __launch_bounds__(256) void mt_expand_kernel(const uint32_t* __restrict__ states, uint32_t* __restrict__ seq) {
  __shared__ uint32_t buf[2][MT_N];
  const uint32_t* st = states + (size_t)blockIdx.x * MT_N;
  uint32_t* out = seq + (size_t)blockIdx.x * MT_SEQ_WORDS;
  for (int i = threadIdx.x; i < MT_N; i += blockDim.x) buf[0][i] = st[i];
  __syncthreads();
  int c = 0;
  for (int b = 0; b < MT_SEQ_BLOCKS; ++b) {
    for (int i = threadIdx.x; i < MT_N; i += blockDim.x) out[(size_t)b * MT_N + i] = buf[c][i];
    if (b + 1 < MT_SEQ_BLOCKS) mt_next_block(buf[c], buf[c ^ 1]);
    c ^= 1;
  }
}

// dst state word i = XOR_j seq[src][i + pos[j]]   (one workgroup of 640 threads per destination)
__global__ __launch_bounds__(640) void mt_combine_kernel(const uint32_t* __restrict__ seq, const uint16_t* __restrict__ pos,
                                                         int npos, uint32_t* __restrict__ states_dst) {
  const int i = threadIdx.x;
  if (i >= MT_N) return;
  const uint32_t* s = seq + (size_t)blockIdx.x * MT_SEQ_WORDS + i;
  uint32_t acc = 0;
  for (int j = 0; j < npos; ++j) acc ^= s[pos[j]];
  states_dst[(size_t)blockIdx.x * MT_N + i] = acc;
}

// One workgroup per segment.  FILL = false: counts[seg] = accepted attempts of the segment.
// FILL = true: writes (f x2, f x1) of every accepted attempt whose cell index < ncells.
template <bool FILL>
__global__ __launch_bounds__(256) void mt_polar_kernel(const uint32_t* __restrict__ states, int blocks_per_segment,
                                                       long long total_blocks, unsigned long long* __restrict__ counts,
                                                       const unsigned long long* __restrict__ offsets,
                                                       double* __restrict__ noise, unsigned long long ncells) {
  __shared__ uint32_t buf[2][MT_N];
  __shared__ uint32_t wsum[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long long seg = blockIdx.x;
  long long nb = total_blocks - seg * blocks_per_segment;
  if (nb > blocks_per_segment) nb = blocks_per_segment;
  const uint32_t* st = states + (size_t)seg * MT_N;
  for (int i = t; i < MT_N; i += blockDim.x) buf[0][i] = st[i];
  __syncthreads();
  int c = 0;
  unsigned long long running = FILL ? offsets[seg] : 0ull;   // cell index of this segment's next accepted attempt
  for (long long b = 0; b < nb; ++b) {
    mt_next_block(buf[c], buf[c ^ 1]);      // the outputs of this block are the tempered words of buf[c ^ 1]
    c ^= 1;
    bool acc = false;
    double x1 = 0, x2 = 0, r2 = 0;
    if (t < MT_N / 4) {
      const uint32_t w0 = mt_temper(buf[c][4 * t]), w1 = mt_temper(buf[c][4 * t + 1]);
      const uint32_t w2 = mt_temper(buf[c][4 * t + 2]), w3 = mt_temper(buf[c][4 * t + 3]);
      const double u1 = ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
      const double u2 = ((double)(w2 >> 5) * 67108864.0 + (double)(w3 >> 6)) / 9007199254740992.0;
      x1 = 2.0 * u1 - 1.0;
      x2 = 2.0 * u2 - 1.0;
      r2 = sum_of_squares(x1, x2);                                // no FMA: numpy's C code rounds both products
      acc = (r2 < 1.0) && (r2 != 0.0);
    }
    const unsigned long long ball = __ballot(acc);
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(ball);
    __syncthreads();
    const uint32_t n0 = wsum[0], n1 = wsum[1], n2 = wsum[2];
    if (FILL && acc) {
      const uint32_t before = (wave > 0 ? n0 : 0u) + (wave > 1 ? n1 : 0u) + (wave > 2 ? n2 : 0u) +
                              (uint32_t)__popcll(ball & ((1ull << lane) - 1ull));
      const unsigned long long cell = running + before;
      if (cell < ncells) {
        // r2 - 1 is exact for r2 >= 1/2, and log1p keeps full relative accuracy where log(r2) -> 0
        const double lg = r2 > 0.5 ? log1p(r2 - 1.0) : log(r2);
        const double f = sqrt(-2.0 * lg / r2);
        noise[2 * cell] = f * x2;           // legacy_gauss returns f*x2 first, then the saved f*x1
        noise[2 * cell + 1] = f * x1;
      }
    }
    running += (unsigned long long)(n0 + n1 + n2);    // waves 0..2 hold the 156 attempts
    __syncthreads();                                   // wsum is rewritten next iteration
  }
  if (!FILL && t == 0) counts[seg] = running;
}

__global__ void mt_scan_kernel(const unsigned long long* __restrict__ counts, unsigned long long* __restrict__ offsets, int n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned long long s = 0;
    for (int i = 0; i < n; ++i) { offsets[i] = s; s += counts[i]; }
    offsets[n] = s;                                    // total number of accepted attempts
  }
}

}  // namespace

hipError_t launch_mt_expand(const uint32_t* states, uint32_t* seq, int nsrc, hipStream_t s) {
  hipLaunchKernelGGL(mt_expand_kernel, dim3(nsrc), dim3(256), 0, s, states, seq);
  return hipGetLastError();
}
hipError_t launch_mt_combine(const uint32_t* seq, const uint16_t* pos, int npos, uint32_t* states_dst, int ndst, hipStream_t s) {
  hipLaunchKernelGGL(mt_combine_kernel, dim3(ndst), dim3(640), 0, s, seq, pos, npos, states_dst);
  return hipGetLastError();
}
hipError_t launch_mt_polar(bool fill, const uint32_t* states, int nseg, int blocks_per_segment, long long total_blocks,
                           unsigned long long* counts, const unsigned long long* offsets, double* noise,
                           unsigned long long ncells, hipStream_t s) {
  if (fill) hipLaunchKernelGGL(mt_polar_kernel<true>, dim3(nseg), dim3(256), 0, s, states, blocks_per_segment, total_blocks, counts, offsets, noise, ncells);
  else hipLaunchKernelGGL(mt_polar_kernel<false>, dim3(nseg), dim3(256), 0, s, states, blocks_per_segment, total_blocks, counts, offsets, noise, ncells);
  return hipGetLastError();
}
hipError_t launch_mt_scan(const unsigned long long* counts, unsigned long long* offsets, int n, hipStream_t s) {
  hipLaunchKernelGGL(mt_scan_kernel, dim3(1), dim3(64), 0, s, counts, offsets, n);
  return hipGetLastError();
}
int mt_seq_words() { return MT_SEQ_WORDS; }

}  // namespace rf
