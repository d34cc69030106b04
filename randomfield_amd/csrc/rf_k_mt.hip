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
//      LDS and XORs them; stage t of the radix-16 tree multiplies the number of segment start states by 16.
//   2. mt_polar_kernel: every segment (1024 blocks of 624 outputs, one WAVE each) generates its outputs ONCE, writes the
//      deviates of its accepted attempts densely into its own run of a scratch array and counts them;
//   3. an exclusive scan of the counts gives each run the index of its first cell; float64 runs are moved into cell order
//      by mt_compact_kernel, float32 runs are read in place by the generation pass through a row table (mt_rowtab_kernel, rf_core.h RowLoc).
#include <hip/hip_runtime.h>
#include "rf_launch.h"
#include "rf_core.h"

namespace rf {
namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_SEQ_BLOCKS = 33;                  // 33 * 624 = 20592 >= 19937 + 624 sequence words per source
constexpr int MT_SEQ_WORDS = MT_SEQ_BLOCKS * MT_N;
constexpr int MT_POS_MAX = 12288;                  // staged positions per polynomial (a degree-19936 polynomial has ~10^4 set bits)

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

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// y ^ (t & mask) in one instruction (v_bitop3_b32, truth table 0x78 = a ^ (b & c)): the two masked steps of the tempering
// cost a shift and this instead of shift + and + xor -- tempering is 40 % of the replay pass's VALU work
__device__ __forceinline__ uint32_t xor_masked(uint32_t y, uint32_t t, uint32_t mask) { return __builtin_amdgcn_bitop3_b32(y, t, mask, 0x78); }

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y = xor_masked(y, y << 7, 0x9D2C5680u);
  y = xor_masked(y, y << 15, 0xEFC60000u);
  y ^= y >> 18;
  return y;
}

// Jumps of one tree stage.  Stage t of the radix-R tree turns the start states of segments [0, R^t) into those of
// [R^t, R^(t+1)): destination i + m R^t = source i advanced by m R^t segments, m = 1 .. R-1, with the jump polynomial
// t^(m R^t L) mod phi (row m-1 of the stage's position table).  One workgroup per (source, group of multipliers): the
// 33-block sequence window of the source (82 KB) is generated into LDS ONCE, then for every multiplier of the group
// lane i XORs the window words i + pos[j] over the set coefficients j of that polynomial (~10^4 conflict-free LDS
// reads per lane; as global-memory reads this step took 13 of the replay's 28 ms in round 1).  Building the window
// is about half of a single jump, and a binary tree spends nine of its thirteen levels waiting for one or a few
// workgroups: radix 16 needs four launches and shares each window between up to 15 jumps (5.6 -> 2 ms at 1024^3).
// Thread layout of the XOR loop (round 4: 16-byte reads): MT_JUMP_GROUPS groups of 160 threads, 157 of them active.  Lane u reads the
// four window words 4u - c + p .. + 3 of a position p = c (mod 4) as ONE aligned ds_read_b128 (256 B/clk/CU like ds_read_b64, but
// one address add and one issue slot per 16 bytes instead of per 8: the 8-byte form ran at ~130 B/clk, bound by its 2.25 vector
// instructions per read): the byte address is 16 u + 4 (p - c), so the positions of a polynomial come in FOUR lists, one per
// class c, stored as the byte offsets 4 (p - c) -- uniform values, read four at a time as one broadcast -- and every lane keeps
// one four-word accumulator per class: class c of lane u belongs to the output words 4u - c .. 4u - c + 3 (lane 156 exists for the
// words 624 - c .. 623 of the classes c > 0).  The lists are padded to whole chunks of 8 with a position inside a block of
// zero words behind the window, so the loop has no tail.  A group takes every MT_JUMP_GROUPS-th chunk of each list; the groups'
// (and classes') partial XORs meet in LDS, indexed by output word, in the space the position table occupied.
// History of one jump on a CU: 0.19 ms with one word per lane and 10 waves, 0.14 ms with ds_read2st64_b32 and 15 waves, 0.087 ms
// with ds_read_b64, see DESIGN.md section 3.6 for the 16-byte form.
constexpr int MT_JUMP_LANES = 160, MT_JUMP_USED = MT_N / 4 + 1, MT_JUMP_GROUPS = 4, MT_JUMP_THREADS = MT_JUMP_LANES * MT_JUMP_GROUPS;
constexpr int MT_ZERO_WORDS = 4 * MT_JUMP_LANES;                         // zero block behind the window: the padding position of every class
constexpr int MT_PART_WORDS = 4 * MT_JUMP_LANES;                          // per (group, class): output words 0 .. 623 (+ slack)
constexpr int MT_JUMP_POS_BYTES = MT_POS_MAX * (int)sizeof(uint32_t);
constexpr int MT_JUMP_PART_BYTES = MT_JUMP_GROUPS * 4 * MT_PART_WORDS * (int)sizeof(uint32_t);
constexpr int MT_JUMP_LDS = (MT_SEQ_WORDS + MT_ZERO_WORDS) * (int)sizeof(uint32_t) +
                            (MT_JUMP_POS_BYTES > MT_JUMP_PART_BYTES ? MT_JUMP_POS_BYTES : MT_JUMP_PART_BYTES);
static_assert(MT_JUMP_THREADS >= MT_N && MT_JUMP_THREADS <= 1024, "the window build and the final combine use one thread per state word");
__global__ __launch_bounds__(MT_JUMP_THREADS) void mt_jump_kernel(const uint32_t* __restrict__ states, const uint32_t* __restrict__ pos,
                                                      const int* __restrict__ npos, int pos_stride, int nsrc, long long dist,
                                                      int nmult, int mult_per_wg, int nseg) {
  extern __shared__ __attribute__((aligned(16))) uint32_t win[];     // MT_SEQ_WORDS words + MT_ZERO_WORDS zeros, then positions / partials
  const int t = threadIdx.x;
  const int src = blockIdx.x % nsrc, m0 = 1 + (blockIdx.x / nsrc) * mult_per_wg;
  if ((long long)src + (long long)m0 * dist >= nseg) return;         // uniform: no destination of this workgroup exists
  const uint32_t* st = states + (size_t)src * MT_N;
  if (t < MT_N) win[t] = st[t];
  if (t < MT_ZERO_WORDS) win[MT_SEQ_WORDS + t] = 0u;
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
  // the ~10^4 positions of a polynomial are staged in LDS behind the window (32-bit byte offsets, four per 16-byte broadcast read;
  // row layout in global memory: the four class lists one after the other, each padded to a multiple of 8 -- npos holds the four
  // padded counts)
  uint32_t* lpos = win + MT_SEQ_WORDS + MT_ZERO_WORDS;
  uint32_t* part = lpos;                                              // (after the loop: [group][class][output word])
  const int grp = t / MT_JUMP_LANES, u = t - grp * MT_JUMP_LANES;
  const bool active = u < MT_JUMP_USED;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const char* wb = reinterpret_cast<const char*>(win) + 16 * (active ? u : 0);
  for (int m = m0; m < m0 + mult_per_wg && m <= nmult; ++m) {
    const long long dst = (long long)src + (long long)m * dist;
    if (dst >= nseg) break;                                           // uniform
    const uint32_t* pm = pos + (size_t)(m - 1) * pos_stride;
    const int n0 = npos[4 * (m - 1)], n1 = npos[4 * (m - 1) + 1], n2 = npos[4 * (m - 1) + 2], n3 = npos[4 * (m - 1) + 3];
    const int total = n0 + n1 + n2 + n3;
    __syncthreads();                                                  // the previous multiplier is done with the partials
    for (int i = t; i < total; i += MT_JUMP_THREADS) lpos[i] = pm[i];
    __syncthreads();
    u4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int off = c == 0 ? 0 : (c == 1 ? n0 : (c == 2 ? n0 + n1 : n0 + n1 + n2));
      const int n = c == 0 ? n0 : (c == 1 ? n1 : (c == 2 ? n2 : n3));
      const u4* lp = reinterpret_cast<const u4*>(lpos + off);         // (off is a multiple of 8 words)
      u4 a = {0u, 0u, 0u, 0u};
#pragma unroll 2                           // (chunks of 8 reads in flight per lane; 1 and 4 measured: no difference)
      for (int j = 2 * grp; j < (n >> 2); j += 2 * MT_JUMP_GROUPS) {  // a chunk of 8 positions = two broadcast reads
        const u4 q0 = lp[j], q1 = lp[j + 1];
#define RF_MT_QUAD(o) (*reinterpret_cast<const u4*>(__builtin_assume_aligned(wb + (o), 16)))
        const u4 v0 = RF_MT_QUAD(q0.x), v1 = RF_MT_QUAD(q0.y), v2 = RF_MT_QUAD(q0.z), v3 = RF_MT_QUAD(q0.w), v4 = RF_MT_QUAD(q1.x),
                 v5 = RF_MT_QUAD(q1.y), v6 = RF_MT_QUAD(q1.z), v7 = RF_MT_QUAD(q1.w);
#undef RF_MT_QUAD
        a.x = xor3(xor3(xor3(xor3(a.x, v0.x, v1.x), v2.x, v3.x), v4.x, v5.x), v6.x, v7.x);
        a.y = xor3(xor3(xor3(xor3(a.y, v0.y, v1.y), v2.y, v3.y), v4.y, v5.y), v6.y, v7.y);
        a.z = xor3(xor3(xor3(xor3(a.z, v0.z, v1.z), v2.z, v3.z), v4.z, v5.z), v6.z, v7.z);
        a.w = xor3(xor3(xor3(xor3(a.w, v0.w, v1.w), v2.w, v3.w), v4.w, v5.w), v6.w, v7.w);
      }
      acc[c] = a;
    }
    __syncthreads();                                                  // every group has read its positions: the partials take their place
    if (active) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint32_t* pc = part + (grp * 4 + c) * MT_PART_WORDS;
        const int w0 = 4 * u - c;                                     // output word of component 0
        const uint32_t comp[4] = {acc[c].x, acc[c].y, acc[c].z, acc[c].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (w0 + k >= 0 && w0 + k < MT_N) pc[w0 + k] = comp[k];
      }
    }
    __syncthreads();
    if (t < MT_N) {
      uint32_t x = 0;
#pragma unroll
      for (int gc = 0; gc < MT_JUMP_GROUPS * 4; ++gc) x ^= part[gc * MT_PART_WORDS + t];
      const_cast<uint32_t*>(states)[(size_t)dst * MT_N + t] = x;
    }
  }
}

// One WAVE per segment (4 segments per 256-thread workgroup), no workgroup barriers.  The segment generates its outputs ONCE: it
// writes (f x2, f x1) of its accepted attempts densely from slot seg * cap of the scratch array `runs` (cap = attempts per segment,
// so a run always fits) and leaves their number in counts[seg]; the scan of the counts tells later stages which cells a run holds
// (mt_compact_kernel moves float64 runs into cell order, the generation pass reads float32 runs in place).  Round 1 generated
// every block twice: a count pass, then a fill pass.
//
// Round 5: TWO blocks per round.  The 624-word state ping-pongs between two windows of the wave's LDS (P <- Q, then Q <- P, in 64-lane
// chunks: word i needs the old words i, i + 1 and either the old word i + 397 (i < 227) or the new word i - 227, which an earlier
// chunk has written -- LDS operations of one wave execute in order, `volatile` keeps the compiler from reordering them), and the
// 2 x 156 polar attempts of the pair are dealt to the lanes together: five wave iterations at 97.5 % of the lanes instead of
// 2 x three at 81 %.
//
// F32: the pair is written as two float32 -- for float32 plans, whose cells sigma * g are float32 anyway.  What decides WHICH cell a
// deviate belongs to (accept / reject) is never left to a float32 rounding:
//   fast path, every lane: x = (2a + 1 - 2^27) 2^-27 from the first word of each uniform (the second word's contribution, uniform in
//     +-2^-27, has zero mean and lies below the float32 rounding of x for |x| > 1/16: dropping the always-positive b 2^-52 instead
//     shifts every deviate by -2^-27, which adds up coherently at the field's origin: 6e-5 of the rms at 512^3), r2 = x1^2 + x2^2 to ~2e-7,
//     f = sqrt(-2 ln r2 / r2) with the hardware log2 / rcp / sqrt.  Final for 2^-9 <= r2 <= 1 - 2^-8 (78 % of the attempts) and for
//     r2 >= 1 + 1e-6 (rejected for certain).
//   the bands r2 > 1 - 2^-8 (ln r2 -> 0 loses the relative accuracy of r2; acceptance uncertain within 1e-6 of 1) and r2 < 2^-9 (the
//     5-sigma deviates: the left-out +-2^-27 would be more than 2.4e-7 of x), and every lane of the float64 form: numpy's own float64
//     arithmetic -- x = a 2^-26 + (b 2^-52 - 1) in two fused multiply-adds (every step of numpy's (a 2^26 + b) / 2^53, 2u - 1 is exact
//     in float64, so this is the same number), r2 with both products rounded as numpy's C code rounds them, 0 < r2 < 1.  One wave
//     iteration in four holds such a lane and pays ~60 float64-rate instructions and two more temperings.
//   (Round 5 replaced the float64 band path by compensated float32 arithmetic -- d = r2 - 1 from the exact squares of the 28-bit
//   integers by TwoSum / fused-multiply-add residuals, log1p by its series, float64 only for |d| < 2^-11: no acceptance differed over
//   4e7 modelled attempts, and the replay took exactly as long (3.74 - 3.78 against 3.73 - 3.77 ms per 1024^3, profiles/r05_ab/
//   r05_d_mt_ab.log): the pass is bound by the latency chain of its in-order LDS regeneration, not by those instructions.  Removed.)
typedef __attribute__((address_space(3))) uint32_t mt_lds_u32;

// the next 624 words: nw <- f(od) (od == nw: in place)
__device__ __forceinline__ void mt_wave_regen(volatile mt_lds_u32* nw, volatile mt_lds_u32* od, int lane) {
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const int i = 64 * k + lane;
    if (i < MT_N) {
      // word 623 follows the same rule with its neighbour wrapped to the (new) word 0: 623 - 227 = 396
      const uint32_t third = i < MT_N - MT_M ? od[i + MT_M] : nw[i - (MT_N - MT_M)];
      const uint32_t nxt = i + 1 == MT_N ? nw[0] : od[i + 1];
      const uint32_t v = mt_f(od[i], nxt, third);
      nw[i] = v;                                           // (in place: every lane has read before any lane writes -- lock step)
    }
  }
}

// one polar attempt from its four raw (untempered) words: accepted?  F32: (g0, g1) = (f x2, f x1); else (x1, x2, r2) for the float64 store
template <bool F32>
__device__ __forceinline__ bool mt_polar_attempt(const unsigned raw_x, const unsigned raw_y, const unsigned raw_z, const unsigned raw_w,
                                                 float& g0, float& g1, double& x1, double& x2, double& r2) {
  const uint32_t w0 = mt_temper(raw_x), w2 = mt_temper(raw_z);
  bool acc = false, exact = true;
  if (F32) {
    const int S1 = (int)((w0 >> 4) | 1u) - (1 << 27), S2 = (int)((w2 >> 4) | 1u) - (1 << 27);
    const float hi1 = (float)S1, hi2 = (float)S2;
    const float x1f = hi1 * 0x1p-27f, x2f = hi2 * 0x1p-27f;
    const float r2f = fmaf(x1f, x1f, x2f * x2f);
    if (r2f <= 0.99609375f && r2f >= 0.001953125f) {                       // the fast path is final
      exact = false;
      acc = true;
      const float inv = __builtin_amdgcn_rcpf(r2f);
      const float f = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(r2f) * inv);      // -2 ln 2 log2 r2 / r2
      g0 = f * x2f;
      g1 = f * x1f;
    } else if (r2f >= 1.000001f) {                                         // rejected for certain
      exact = false;
    }
  }
  if (exact) {
    const uint32_t w1 = mt_temper(raw_y), w3 = mt_temper(raw_w);
    x1 = fma((double)(w0 >> 5), 0x1p-26, fma((double)(w1 >> 6), 0x1p-52, -1.0));
    x2 = fma((double)(w2 >> 5), 0x1p-26, fma((double)(w3 >> 6), 0x1p-52, -1.0));
    r2 = sum_of_squares(x1, x2);                          // no FMA: numpy's C code rounds both products
    acc = (r2 < 1.0) && (r2 != 0.0);
    if (F32 && acc) {
      // log(r2) = log(r2f) + (r2 - r2f) / r2f: the hardware log2 (1 ulp of its result, also where it -> 0) of the
      // rounded argument, plus the first-order term of the rounding (exact difference in float64)
      const float r2f = (float)r2, inv = __builtin_amdgcn_rcpf(r2f);
      const float lg = fmaf(__builtin_amdgcn_logf(r2f), 0.69314718056f, (float)(r2 - (double)r2f) * inv);
      const float f = __builtin_amdgcn_sqrtf(-2.0f * lg * inv);
      g0 = f * (float)x2;
      g1 = f * (float)x1;
    }
  }
  return acc;
}

template <bool F32>
__global__ __launch_bounds__(256) void mt_polar_kernel(const uint32_t* __restrict__ states, int blocks_per_segment,
                                                       long long total_blocks, int nseg, unsigned long long* __restrict__ counts,
                                                       double* __restrict__ runs, unsigned long long cap) {
  constexpr int NWIN = 2, WIN = MT_N + 16, NATT = MT_N / 4;
  __shared__ __attribute__((aligned(16))) uint32_t lds[4][NWIN * WIN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long seg = (long long)blockIdx.x * 4 + wave;
  if (seg >= nseg) return;                                   // whole waves leave; nobody waits at a barrier
  // explicitly LDS-qualified pointers: a plain `volatile uint32_t*` is a GENERIC pointer to the compiler, and every
  // access through it became a flat_load / flat_store (slow path into LDS, and ordered behind this wave's outstanding
  // global stores of deviates through the shared vmcnt counter) instead of ds_read / ds_write
  volatile mt_lds_u32* Q = (volatile mt_lds_u32*)(&lds[wave][0]);
  volatile mt_lds_u32* P = (volatile mt_lds_u32*)(&lds[wave][(NWIN - 1) * WIN]);     // (one window: P == Q, regeneration in place)
  long long nb = total_blocks - seg * blocks_per_segment;
  if (nb > blocks_per_segment) nb = blocks_per_segment;
  const uint32_t* st = states + (size_t)seg * MT_N;
  for (int i = lane; i < MT_N; i += 64) Q[i] = st[i];
  unsigned long long running = (unsigned long long)seg * cap;      // slot of this segment's next accepted attempt
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  // the attempts a = first + lane, first = 0, 64, ... < natt, of the windows [P (NATT attempts), then Q]: attempt a uses outputs 4a .. 4a+3
  auto attempts = [&](int natt) {
    for (int first = 0; first < natt; first += 64) {
      const int a = first + lane;
      bool acc = false;
      double x1 = 0, x2 = 0, r2 = 0;
      float g0 = 0, g1 = 0;
      if (a < natt) {
        // the attempt's four words with ONE 16-byte LDS read (four 4-byte reads at a lane stride of 16 bytes are
        // 4-way bank conflicts); the compiler barrier keeps it behind the volatile regeneration above,
        // and the LDS operations of one wave execute in program order
        asm volatile("" ::: "memory");
        const uint32_t* wbase = const_cast<const uint32_t*>(lds[wave]) + (a < NATT ? (NWIN - 1) * WIN + 4 * a : 4 * (a - NATT));
        const u4 raw = *reinterpret_cast<const u4*>(wbase);
        acc = mt_polar_attempt<F32>(raw.x, raw.y, raw.z, raw.w, g0, g1, x1, x2, r2);
      }
      const unsigned long long ball = __ballot(acc);
      if (acc) {
        const unsigned long long dst = running + (unsigned long long)__popcll(ball & ((1ull << lane) - 1ull));
        if (F32) {
          reinterpret_cast<float2*>(runs)[dst] = make_float2(g0, g1);
        } else {
          const double f = sqrt(-2.0 * log(r2) / r2);
          runs[2 * dst] = f * x2;                             // legacy_gauss returns f*x2 first, then the saved f*x1
          runs[2 * dst + 1] = f * x1;
        }
      }
      running += (unsigned long long)__popcll(ball);
    }
  };
  long long b = 0;
  for (; b + 2 <= nb; b += 2) {
    mt_wave_regen(P, Q, lane);                               // block b     : P <- f(Q)
    mt_wave_regen(Q, P, lane);                               // block b + 1 : Q <- f(P)
    attempts(2 * NATT);
  }
  for (; b < nb; ++b) {                                      // (the odd block of a segment; every block when pairs are off)
    mt_wave_regen(P, NWIN == 2 ? Q : P, lane);
    attempts(NATT);
    if (NWIN == 2) {                                         // the state moves back to Q for whatever follows
      asm volatile("" ::: "memory");
      for (int i = lane; i < MT_N; i += 64) Q[i] = P[i];
    }
  }
  if (lane == 0) counts[seg] = running - (unsigned long long)seg * cap;
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
  for (int i = lo; i < hi; ++i) {
    offsets[i] = run;
    run += counts[i];
  }
  if (lane == 63) offsets[n] = inc;                    // total number of accepted attempts
}

// The row table of the float32 form (rf_core.h RowLoc): the generation pass reads the pairs where the segments left them, and
// entry iy * nx + ix says where the nz/2 + 1 cells of row (ix, iy) start.  One thread per row, a binary search in the scan (32 KB at
// 1024^3: L2-resident).  flags[0] |= 1 when a row spans more than two segments (segments shorter than a row).
__global__ __launch_bounds__(256) void mt_rowtab_kernel(const unsigned long long* __restrict__ offsets, int nseg, RowLoc* __restrict__ tab,
                                                        int nx, int ny, int nzh, int* __restrict__ flags) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)nx * ny) return;
  const int iy = (int)(e / nx), ix = (int)(e - (long long)iy * nx);
  int bad = 0;
  tab[e] = make_rowloc(offsets, nseg, ((unsigned long long)ix * ny + iy) * (unsigned)nzh, (unsigned)nzh, &bad);
  if (bad) atomicOr(flags, 1);
}

// pairs [0, counts[seg]) of segment seg's scratch run -> cells offsets[seg] + i of the stream (cells >= ncells are
// dropped; a kz-slab rank keeps its own planes, as in mt_polar_kernel).  PAIR = float2 or double2.
constexpr int MT_COMPACT_CHUNK = 16384;
template <typename PAIR>
__global__ __launch_bounds__(256) void mt_compact_kernel(const PAIR* __restrict__ scratch, const unsigned long long* __restrict__ counts,
                                                         const unsigned long long* __restrict__ offsets, unsigned long long cap,
                                                         PAIR* __restrict__ noise, unsigned long long ncells, int nzh, int zpitch,
                                                         int zoff) {
  const unsigned long long seg = blockIdx.x, n = counts[seg], first = offsets[seg];
  const unsigned long long lo = (unsigned long long)blockIdx.y * MT_COMPACT_CHUNK;
  if (lo >= n) return;
  const unsigned long long hi = lo + MT_COMPACT_CHUNK < n ? lo + MT_COMPACT_CHUNK : n;
  const PAIR* src = scratch + seg * cap;
  for (unsigned long long i = lo + threadIdx.x; i < hi; i += 256) {
    const unsigned long long cell = first + i;
    if (cell >= ncells) break;
    unsigned long long dst = cell;
    if (zpitch != nzh) {
      const unsigned long long col = cell / (unsigned)nzh;
      const int kz = (int)(cell - col * (unsigned)nzh);
      const int sl = kz == nzh - 1 ? zpitch - 1 : kz - zoff;
      if (!(sl >= 0 && sl < zpitch && (kz == nzh - 1 || sl < zpitch - 1))) continue;
      dst = col * (unsigned)zpitch + (unsigned)sl;
    }
    noise[dst] = src[i];
  }
}


// ---- distributed replay (one stream, P ranks; rf_capi.hip rf_mt_share_*) ----------------------------------------------------
// Every rank replays a contiguous range of segments; the pairs a kz-slab rank q needs (its nzl planes of every row and the Nyquist
// plane, which every rank's side array carries) are "stream q": the cells (col, kz) with kz in q's planes or kz = nz/2, in stream
// order -- index col * (nzl + 1) + slot, slot = kz - q nzl (Nyquist plane: nzl), which is also the layout of the plan's resident
// deviates.  Pack: pair i of local segment seg is cell first[seg] + i; it goes to send[sbase[q] + index in stream q], sbase[q] =
// start of q's region in the send buffer minus the stream-q index of this rank's first cell, so the regions are dense (a device
// array of P entries).  PAIR = float2 or double2.
template <typename PAIR>
__global__ __launch_bounds__(256) void mt_share_pack_kernel(const PAIR* __restrict__ scratch, const unsigned long long* __restrict__ counts,
                                                            const unsigned long long* __restrict__ first_cell, unsigned long long cap,
                                                            PAIR* __restrict__ send, unsigned long long ncells, int nzh, int nzl, int nranks,
                                                            const long long* __restrict__ sbase) {
  const unsigned long long seg = blockIdx.x, n = counts[seg], first = first_cell[seg];
  const unsigned long long lo = (unsigned long long)blockIdx.y * MT_COMPACT_CHUNK;
  if (lo >= n) return;
  const unsigned long long hi = lo + MT_COMPACT_CHUNK < n ? lo + MT_COMPACT_CHUNK : n;
  const PAIR* src = scratch + seg * cap;
  for (unsigned long long i = lo + threadIdx.x; i < hi; i += 256) {
    const unsigned long long cell = first + i;
    if (cell >= ncells) break;
    const unsigned long long col = cell / (unsigned)nzh;
    const int kz = (int)(cell - col * (unsigned)nzh);
    const long long rowbase = (long long)col * (nzl + 1);
    const PAIR v = src[i];
    if (kz == nzh - 1) {
      for (int q = 0; q < nranks; ++q) send[sbase[q] + rowbase + nzl] = v;
    } else {
      const int q = kz / nzl;
      send[sbase[q] + rowbase + (kz - q * nzl)] = v;
    }
  }
}
// stream q as received, float32 pairs -> the plan's resident float64 deviates (same layout; float64 streams are received in place)
__global__ __launch_bounds__(256) void mt_share_widen_kernel(const float2* __restrict__ recv, double2* __restrict__ noise, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float2 v = recv[i];
  noise[i] = make_double2((double)v.x, (double)v.y);
}

}  // namespace

hipError_t launch_mt_compact(bool single, const void* scratch, const unsigned long long* counts, const unsigned long long* offsets,
                             int nseg, unsigned long long cap, void* noise, unsigned long long ncells, int nzh, int zpitch, int zoff,
                             hipStream_t s) {
  const dim3 grid((unsigned)nseg, (unsigned)((cap + MT_COMPACT_CHUNK - 1) / MT_COMPACT_CHUNK));
  if (single) hipLaunchKernelGGL(mt_compact_kernel<float2>, grid, dim3(256), 0, s, (const float2*)scratch, counts, offsets, cap, (float2*)noise, ncells, nzh, zpitch, zoff);
  else hipLaunchKernelGGL(mt_compact_kernel<double2>, grid, dim3(256), 0, s, (const double2*)scratch, counts, offsets, cap, (double2*)noise, ncells, nzh, zpitch, zoff);
  return hipGetLastError();
}

hipError_t launch_mt_jump(uint32_t* states, const uint32_t* pos, const int* npos, int pos_stride, int nsrc, long long dist,
                          int nmult, int nseg, hipStream_t s) {
  if (pos_stride > MT_POS_MAX) return hipErrorInvalidValue;
  constexpr int lds = MT_JUMP_LDS;                                                    // window + zeros, positions / partial XORs
  static LdsAttrLatch latch;
  if (hipError_t e = latch.ensure((const void*)mt_jump_kernel, lds); e != hipSuccess) return e;
  // one 82 KB window (+ 24 KB of positions) per CU: share a source's window between as many multipliers as it takes to fit one round
  const int cus = device_cu_count();
  int per = (int)(((long long)nsrc * nmult + cus - 1) / cus);
  per = per < 1 ? 1 : (per > nmult ? nmult : per);
  const int groups = (nmult + per - 1) / per;
  hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned)(nsrc * groups)), dim3(MT_JUMP_THREADS), lds, s, states, pos, npos, pos_stride, nsrc, dist, nmult,
                     per, nseg);
  return hipGetLastError();
}
hipError_t launch_mt_polar(bool single, const uint32_t* states, int nseg, int blocks_per_segment, long long total_blocks,
                           unsigned long long* counts, void* runs, unsigned long long cap, hipStream_t s) {
  const unsigned grid = (unsigned)((nseg + 3) / 4);
  if (single) hipLaunchKernelGGL(mt_polar_kernel<true>, dim3(grid), dim3(256), 0, s, states, blocks_per_segment, total_blocks, nseg, counts, (double*)runs, cap);
  else hipLaunchKernelGGL(mt_polar_kernel<false>, dim3(grid), dim3(256), 0, s, states, blocks_per_segment, total_blocks, nseg, counts, (double*)runs, cap);
  return hipGetLastError();
}
hipError_t launch_mt_scan(const unsigned long long* counts, unsigned long long* offsets, int n, hipStream_t s) {
  hipLaunchKernelGGL(mt_scan_kernel, dim3(1), dim3(64), 0, s, counts, offsets, n);
  return hipGetLastError();
}

hipError_t launch_mt_rowtab(const unsigned long long* offsets, int nseg, void* tab, int nx, int ny, int nzh, int* flags, hipStream_t s) {
  if (nseg < 1 || nzh >= (1 << ROWLOC_NBITS) || nseg >= (1 << (32 - ROWLOC_NBITS))) return hipErrorInvalidValue;
  const long long n = (long long)nx * ny;
  hipLaunchKernelGGL(mt_rowtab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, offsets, nseg, (RowLoc*)tab, nx, ny, nzh, flags);
  return hipGetLastError();
}

hipError_t launch_mt_share_pack(bool single, const void* scratch, const unsigned long long* counts, const unsigned long long* first_cell,
                                int nseg, unsigned long long cap, void* send, unsigned long long ncells, int nzh, int nzl, int nranks,
                                const long long* sbase_dev, hipStream_t s) {
  if (nseg <= 0 || nzl <= 0 || nranks <= 0) return hipErrorInvalidValue;
  const dim3 grid((unsigned)nseg, (unsigned)((cap + MT_COMPACT_CHUNK - 1) / MT_COMPACT_CHUNK));
  if (single) hipLaunchKernelGGL(mt_share_pack_kernel<float2>, grid, dim3(256), 0, s, (const float2*)scratch, counts, first_cell, cap, (float2*)send, ncells, nzh, nzl, nranks, sbase_dev);
  else hipLaunchKernelGGL(mt_share_pack_kernel<double2>, grid, dim3(256), 0, s, (const double2*)scratch, counts, first_cell, cap, (double2*)send, ncells, nzh, nzl, nranks, sbase_dev);
  return hipGetLastError();
}
hipError_t launch_mt_share_widen(const void* recv, double* noise, long long npairs, hipStream_t s) {
  if (npairs <= 0 || (npairs + 255) / 256 >= (1LL << 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_share_widen_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, s, (const float2*)recv, (double2*)noise, npairs);
  return hipGetLastError();
}

}  // namespace rf
