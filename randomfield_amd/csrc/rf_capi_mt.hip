// rf_capi_mt.hip -- C-ABI entry points of the on-GPU replay of np.random.RandomState(seed).normal (random.py:24-28): the jump table,
// the replay (float64 / float32 pairs), the stream shared between the ranks of a kz-slab job, same-seed batches.  Kernels: rf_k_mt.hip.
#include "rf_plan.h"

using namespace rfc;

extern "C" {

int rf_mt_set_jump(rf_plan* p, int npolys, const uint16_t* pos, const int* npos, int stride, int blocks_per_segment, int radix) {
  RF_REQUIRE(p && pos && npos, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(npolys >= 1 && stride >= 1 && blocks_per_segment >= 1 && radix >= 2 && npolys % (radix - 1) == 0, "invalid jump table");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipStreamSynchronize(p->stream));
  if (p->mt_pos) RF_HIP(hipFree(p->mt_pos));
  if (p->mt_npos_dev) RF_HIP(hipFree(p->mt_npos_dev));
  p->mt_pos = nullptr;
  p->mt_npos_dev = nullptr;
  // device rows (rf_k_mt.hip mt_jump_kernel): four lists, one per class c = position mod 4, each padded to a multiple of 8 entries;
  // an entry is the byte offset 4 (position - c) of an aligned 16-byte read; the padding points into the block of zero words
  // behind the 33-block window (33 * 624 words); four padded counts per polynomial
  const uint32_t null_off = 4u * 33u * 624u;
  const int wstride = ((stride + 7) / 8 + 4) * 8;
  std::vector<uint32_t> wide((size_t)npolys * wstride, null_off);
  std::vector<int> counts(4 * (size_t)npolys);
  for (int l = 0; l < npolys; ++l) {
    int n[4] = {0, 0, 0, 0};
    for (int j = 0; j < npos[l]; ++j) ++n[pos[(size_t)l * stride + j] & 3];
    int off[4], padded[4];
    for (int c = 0, o = 0; c < 4; ++c) { padded[c] = (n[c] + 7) & ~7; off[c] = o; o += padded[c]; }
    RF_REQUIRE(off[3] + padded[3] <= wstride, "jump table row too long");
    int k[4] = {0, 0, 0, 0};
    for (int j = 0; j < npos[l]; ++j) {
      const uint32_t q = pos[(size_t)l * stride + j];
      const int c = (int)(q & 3u);
      wide[(size_t)l * wstride + off[c] + k[c]++] = 4u * (q - (uint32_t)c);
    }
    for (int c = 0; c < 4; ++c) counts[4 * l + c] = padded[c];
  }
  RF_HIP(hipMalloc((void**)&p->mt_pos, wide.size() * sizeof(uint32_t)));
  RF_HIP(hipMemcpy(p->mt_pos, wide.data(), wide.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  RF_HIP(hipMalloc((void**)&p->mt_npos_dev, counts.size() * sizeof(int)));
  RF_HIP(hipMemcpy(p->mt_npos_dev, counts.data(), counts.size() * sizeof(int), hipMemcpyHostToDevice));
  p->mt_npos.assign(npos, npos + npolys);
  p->mt_stride = wstride;
  p->mt_bps = blocks_per_segment;
  p->mt_radix = radix;
  return 0;
}

int rf_noise_mt19937(rf_plan* p, const uint32_t* state624, unsigned long long* accepted) {
  return rf_noise_mt19937_ex(p, state624, accepted, 0);
}

namespace {
// sizes of one replay of RandomState(seed).normal for this plan's grid (rf_k_mt.hip)
struct MtGeom {
  unsigned long long ncells, attempts, cap;
  long long total_blocks;
  int nseg, stages;
  size_t need;          // bytes of the runs: nseg * cap pairs
};
int mt_geom(rf_plan* p, int single, MtGeom& g) {
  g.ncells = (unsigned long long)p->nx * p->ny * (p->nzc + 1);
  // polar attempts to generate: acceptance pi/4, margin of 10 sigma + 1024 (mt19937.attempts_needed)
  const double pa = 0.78539816339744830962;
  g.attempts = (unsigned long long)std::ceil((double)g.ncells / pa + 10.0 * std::sqrt((double)g.ncells * (1 - pa)) / pa + 1024.0);
  g.total_blocks = (long long)((4 * g.attempts + 623) / 624);
  g.nseg = (int)((g.total_blocks + p->mt_bps - 1) / p->mt_bps);
  // stages of the radix-R jump tree: stage t needs the R - 1 polynomials t^(m R^t L), rows t (R - 1) .. of the table
  const int R = p->mt_radix;
  g.stages = 0;
  long long reach = 1;
  while (reach < g.nseg) { reach *= R; ++g.stages; }
  RF_REQUIRE(g.stages * (R - 1) <= (int)p->mt_npos.size(), "grid too large for the uploaded jump table");
  g.cap = (unsigned long long)p->mt_bps * (624 / 4);
  g.need = (size_t)g.nseg * g.cap * (single ? 2 * sizeof(float) : 2 * sizeof(double));
  return 0;
}
int mt_ensure_buffers(rf_plan* p, const MtGeom& g) {
  const size_t nstates = (size_t)g.nseg;
  if (p->mt_states_cap < nstates) {
    if (p->mt_states) RF_HIP(hipFree(p->mt_states));
    p->mt_states = nullptr;
    RF_HIP(hipMalloc((void**)&p->mt_states, nstates * 624 * sizeof(uint32_t)));
    p->mt_states_cap = nstates;
  }
  if (p->mt_seg_cap < (size_t)g.nseg + 1) {
    if (p->mt_counts) RF_HIP(hipFree(p->mt_counts));
    if (p->mt_offsets) RF_HIP(hipFree(p->mt_offsets));
    p->mt_counts = p->mt_offsets = nullptr;
    RF_HIP(hipMalloc((void**)&p->mt_counts, ((size_t)g.nseg + 1) * sizeof(unsigned long long)));
    RF_HIP(hipMalloc((void**)&p->mt_offsets, ((size_t)g.nseg + 1) * sizeof(unsigned long long)));
    p->mt_seg_cap = (size_t)g.nseg + 1;
  }
  if (!p->mt_rowtab) RF_HIP(hipMalloc(&p->mt_rowtab, (size_t)p->nx * p->ny * sizeof(RowLoc)));
  if (!p->mt_flags) RF_HIP(hipMalloc((void**)&p->mt_flags, sizeof(int)));
  if (p->mt_scratch_bytes < g.need) {
    if (p->mt_scratch) RF_HIP(hipFree(p->mt_scratch));
    p->mt_scratch = nullptr; p->mt_scratch_bytes = 0;
    RF_HIP(hipMalloc(&p->mt_scratch, g.need));
    p->mt_scratch_bytes = g.need;
  }
  return 0;
}
// the replay itself on stream s, from the start state in p->mt_states[0 .. 624): jump tree, ONE generation pass, scan (and the
// move into cell order for float64 deviates).  No host synchronisation.
int mt_queue(rf_plan* p, const MtGeom& g, int single, hipStream_t s) {
  const int R = p->mt_radix;
  // jump tree: stage t turns the start states of segments [0, R^t) into those of [R^t, R^(t+1))
  long long dist = 1;
  for (int t = 0; t < g.stages; ++t, dist *= R) {
    const int nsrc = (int)(dist < g.nseg ? dist : g.nseg);
    RF_HIP(launch_mt_jump(p->mt_states, p->mt_pos + (size_t)t * (R - 1) * p->mt_stride, p->mt_npos_dev + 4 * t * (R - 1), p->mt_stride, nsrc,
                          dist, R - 1, g.nseg, s));
  }
  // ONE generation pass: every segment writes its accepted pairs densely into its own run of the scratch array
  // (capacity = its attempts) and counts them; a scan of the counts gives each run its first cell, and a copy kernel
  // moves the runs into place.  (Round 1 generated every block twice -- a count pass, then a fill pass that knew the
  // offsets: 2.5 + 3.3 ms against 3.3 + 1.x ms for fill + move.)  A kz-slab rank replays the WHOLE stream (where a
  // deviate goes depends on every earlier acceptance) and keeps the deviates of its own planes while moving.
  RF_HIP(launch_mt_polar(single != 0, p->mt_states, g.nseg, p->mt_bps, g.total_blocks, p->mt_counts, p->mt_scratch, g.cap, s));
  RF_HIP(launch_mt_scan(p->mt_counts, p->mt_offsets, g.nseg, s));
  // float64 deviates are moved into cell order (and cut to this rank's planes); float32 ones stay in the segments' runs:
  // the generation pass finds cell c through the scan (slack_cell), which saves the 1.7 ms copy per 1024^3
  if (single) {
    RF_HIP(hipMemsetAsync(p->mt_flags, 0, sizeof(int), s));
    RF_HIP(launch_mt_rowtab(p->mt_offsets, g.nseg, p->mt_rowtab, p->nx, p->ny, (int)p->nzc + 1, p->mt_flags, s));
  }
  if (!single)
    RF_HIP(launch_mt_compact(false, p->mt_scratch, p->mt_counts, p->mt_offsets, g.nseg, g.cap, p->noise, g.ncells, (int)p->nzc + 1,
                             (int)p->nzl + 1, p->kz0, s));
  return 0;
}
}  // namespace

int rf_noise_mt19937_ex(rf_plan* p, const uint32_t* state624, unsigned long long* accepted, int single) {
  RF_REQUIRE(p && state624, "null argument");
  // `single` is a request: plans without the fast float32 generation pass (float64, generic shapes, exact-generation
  // flag, tables too dense for the per-bin records) read float64 deviates and get them
  if (single && (p->f64 || p->generic || !p->have_fast || p->exact_gen)) single = 0;
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->mt_pos && !p->mt_npos.empty(), "rf_mt_set_jump must be called first");
  RF_HIP(hipSetDevice(p->device));
  if (!single)
    if (int rc = ensure_noise(p)) return rc;
  MtGeom g;
  if (int rc = mt_geom(p, single, g)) return rc;
  // the float32 form locates a row's pairs through a table that allows ONE segment boundary per row: segments (cap attempts,
  // ~0.785 cap pairs) must be longer than a row by a wide margin, or the float64 form (moved into cell order) serves
  if (single && g.cap < 4ull * (unsigned long long)(p->nzc + 1)) {
    single = 0;
    if (int rc = ensure_noise(p)) return rc;
    if (int rc = mt_geom(p, single, g)) return rc;
  }
  if (int rc = mt_ensure_buffers(p, g)) return rc;
  hipStream_t s = p->stream;
  RF_HIP(hipMemcpyAsync(p->mt_states, state624, 624 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  if (int rc = mt_queue(p, g, single, s)) return rc;
  const int nseg = g.nseg;
  unsigned long long total = 0;
  int flags = 0;
  p->noise_resident = false;
  p->noise32_resident = false;
  RF_HIP(hipMemcpyAsync(&total, p->mt_offsets + nseg, sizeof(total), hipMemcpyDeviceToHost, s));
  if (single) RF_HIP(hipMemcpyAsync(&flags, p->mt_flags, sizeof(flags), hipMemcpyDeviceToHost, s));
  RF_HIP(hipStreamSynchronize(s));
  p->nseg = nseg;
  p->seg_cap = g.cap;
  if (accepted) *accepted = total;
  RF_REQUIRE(total >= g.ncells, "MT19937 replay: not enough accepted polar attempts (increase the margin)");
  RF_REQUIRE(!(flags & 1), "MT19937 replay: a segment holds fewer deviate pairs than a row of the grid has cells (segment length too short for the float32 form)");
  p->noise_resident = !single;
  p->noise32_resident = single != 0;
  if (!single) {
    // the runs are dead once the compaction has moved them into p->noise (the stream is idle here).  They are 1.27x the noise
    // buffer -- 11 GB at 1024^3, the difference between fitting and not fitting a 2048^3 float64 plan with a saved potential
    // into 288 GB -- so they are given back when the device is getting full; otherwise they stay for the next seed (allocating
    // and releasing 11 GB costs ~0.5 s per call, a hundred times the replay).  (float32 deviates live IN the runs and keep them.)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total_b / 3) {
      RF_HIP(hipFree(p->mt_scratch));
      p->mt_scratch = nullptr;
      p->mt_scratch_bytes = 0;
    }
  }
  return 0;
}

/* ---- one stream, P ranks: the replay of RandomState(seed).normal shared between the ranks of a kz-slab job -----------------
 * rf_noise_mt19937_ex on a multi-rank plan replays the WHOLE stream on every rank (where a deviate goes depends on every
 * earlier acceptance).  Here rank r replays only segments [r nseg / P, (r + 1) nseg / P): (1) rf_mt_share_begin jumps to its first
 * segment (one jump per radix-16 digit), grows the local tree, runs the generation pass and returns its per-segment counts; (2) the
 * host gathers all counts (a few thousand integers) and hands them to rf_mt_share_pack, which scans them -- now every rank knows
 * which cells every rank holds -- and packs the local pairs by destination (kz slab); (3) ONE all-to-all of deviates, 8 B per
 * cell in float32 mode: the same volume as the field's exchange (rf_mt_share_exchange over RCCL, or rf_mt_share_exchange_local
 * between virtual ranks on one device); (4) rf_mt_share_finish leaves them as the plan's resident float64 deviates, the form
 * rf_realise(RF_NOISE_RESIDENT) and the other consumers already read on multi-rank plans.  Per rank: 1/P of the replay's time
 * and of its scratch.  float64 mode moves the exact deviates (16 B per cell) and gives bit for bit what the replicated
 * replay gives; float32 mode (complex64 plans) rounds the Box-Muller factor as rf_noise_mt19937_ex(single = 1) does. */
namespace {
inline int sh_seg_begin(int r, int nseg, int nranks) { return (int)((long long)r * nseg / nranks); }
}
int rf_mt_share_segments(rf_plan* p, int* nseg_total, int* seg_first, int* seg_count) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked && !p->generic, "the distributed replay serves packed plans on the tiled kernels");
  RF_REQUIRE(p->mt_pos && !p->mt_npos.empty(), "rf_mt_set_jump must be called first");
  MtGeom g;
  if (int rc = mt_geom(p, 1, g)) return rc;
  const int a = sh_seg_begin(p->rank, g.nseg, p->nranks), b = sh_seg_begin(p->rank + 1, g.nseg, p->nranks);
  if (nseg_total) *nseg_total = g.nseg;
  if (seg_first) *seg_first = a;
  if (seg_count) *seg_count = b - a;
  return 0;
}

int rf_mt_share_begin(rf_plan* p, const uint32_t* state624, int single, unsigned long long* counts_out) {
  RF_REQUIRE(p && state624 && counts_out, "null argument");
  RF_REQUIRE(!p->unpacked && !p->generic, "the distributed replay serves packed plans on the tiled kernels");
  RF_REQUIRE(!p->replicate, "replicated-generation plans draw native deviates only");
  RF_REQUIRE(p->mt_pos && !p->mt_npos.empty(), "rf_mt_set_jump must be called first");
  RF_REQUIRE(p->nranks >= 1 && p->nranks <= 64, "unsupported number of ranks");
  if (single && p->f64) single = 0;                      // float64 cells: keep the exact deviates
  RF_HIP(hipSetDevice(p->device));
  p->sh_state = 0;
  if (int rc = ensure_noise(p)) return rc;
  MtGeom g;
  if (int rc = mt_geom(p, single, g)) return rc;
  const int first = sh_seg_begin(p->rank, g.nseg, p->nranks), nloc = sh_seg_begin(p->rank + 1, g.nseg, p->nranks) - first;
  RF_REQUIRE(nloc >= 1, "more ranks than segments: use rf_noise_mt19937_ex on this grid");
  MtGeom gl = g;
  gl.nseg = nloc + 8;                                     // + slots for the jump to the first segment
  gl.need = (size_t)nloc * g.cap * (single ? 2 * sizeof(float) : 2 * sizeof(double));
  if (int rc = mt_ensure_buffers(p, gl)) return rc;
  hipStream_t s = p->stream;
  const int R = p->mt_radix;
  // the start state of segment `first`: one jump per non-zero radix-R digit of `first` (digit d of weight R^t: polynomial
  // t (R - 1) + d - 1 of the table), hopping through the spare slots behind the local states
  int slot = nloc;
  RF_HIP(hipMemcpyAsync(p->mt_states + (size_t)slot * 624, state624, 624 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  {
    int rest = first;
    for (int t = 0; rest > 0; ++t, rest /= R) {
      const int d = rest % R;
      if (d == 0) continue;
      RF_REQUIRE(t < g.stages && slot + 1 < nloc + 8, "segment index beyond the uploaded jump table");
      const int row = t * (R - 1) + d - 1;
      RF_HIP(launch_mt_jump(p->mt_states + (size_t)slot * 624, p->mt_pos + (size_t)row * p->mt_stride, p->mt_npos_dev + 4 * row, p->mt_stride,
                            1, 1, 1, 2, s));
      ++slot;
    }
  }
  RF_HIP(hipMemcpyAsync(p->mt_states, p->mt_states + (size_t)slot * 624, 624 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  // the local tree and the generation pass over the local segments (mt_queue with a shifted origin)
  long long dist = 1;
  for (int t = 0; dist < nloc; ++t, dist *= R) {
    const int nsrc = (int)(dist < nloc ? dist : nloc);
    RF_HIP(launch_mt_jump(p->mt_states, p->mt_pos + (size_t)t * (R - 1) * p->mt_stride, p->mt_npos_dev + 4 * t * (R - 1), p->mt_stride, nsrc,
                          dist, R - 1, nloc, s));
  }
  RF_HIP(launch_mt_polar(single != 0, p->mt_states, nloc, p->mt_bps, g.total_blocks - (long long)first * p->mt_bps, p->mt_counts, p->mt_scratch, g.cap, s));
  RF_HIP(hipMemcpyAsync(counts_out, p->mt_counts, (size_t)nloc * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  RF_HIP(hipStreamSynchronize(s));
  p->sh_single = single; p->sh_first = first; p->sh_nloc = nloc;
  p->noise_resident = false;                              // (p->noise is about to be overwritten)
  p->noise32_resident = false;                            // (the runs in mt_scratch are this rank's share only)
  p->sh_state = 1;
  return 0;
}

// every rank's per-segment counts, in segment order, on every rank: an integer sum over the communicator of arrays that are zero
// outside the rank's own range (a few thousand values; also the first collective after the local replays)
int rf_mt_share_gather(rf_plan* p, unsigned long long* counts_all) {
  RF_REQUIRE(p && counts_all, "null argument");
  RF_REQUIRE(p->sh_state == 1, "rf_mt_share_begin must be called first");
  RF_REQUIRE(p->nranks == 1 || p->comm != nullptr, "rf_comm_init has not been called on this multi-rank plan");
  RF_HIP(hipSetDevice(p->device));
  MtGeom g;
  if (int rc = mt_geom(p, p->sh_single, g)) return rc;
  if (p->comm_stream) RF_HIP(hipStreamSynchronize(p->comm_stream));          // (the communicator is used from one stream at a time)
  unsigned long long* tmp = nullptr;
  RF_HIP(hipMalloc((void**)&tmp, (size_t)g.nseg * sizeof(unsigned long long)));
  hipStream_t s = p->stream;
  hipError_t e = hipMemsetAsync(tmp, 0, (size_t)g.nseg * sizeof(unsigned long long), s);
  if (e == hipSuccess) e = hipMemcpyAsync(tmp + p->sh_first, p->mt_counts, (size_t)p->sh_nloc * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess && p->comm && g_rccl.AllReduce(tmp, tmp, (size_t)g.nseg, ncclUint64, ncclSum, p->comm, s) != ncclSuccess) e = hipErrorUnknown;
  if (e == hipSuccess) e = hipMemcpyAsync(counts_all, tmp, (size_t)g.nseg * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(tmp);
  RF_HIP(e);
  return 0;
}

int rf_mt_share_pack(rf_plan* p, const unsigned long long* counts_all) {
  RF_REQUIRE(p && counts_all, "null argument");
  RF_REQUIRE(p->sh_state == 1, "rf_mt_share_begin must be called first");
  RF_HIP(hipSetDevice(p->device));
  MtGeom g;
  if (int rc = mt_geom(p, p->sh_single, g)) return rc;
  const int P = p->nranks, nzl = (int)p->nzl, nzh = (int)p->nzc + 1;
  std::vector<unsigned long long> off((size_t)g.nseg + 1);
  off[0] = 0;
  for (int i = 0; i < g.nseg; ++i) {
    RF_REQUIRE(counts_all[i] <= g.cap, "a segment cannot hold more pairs than attempts: the gathered counts are corrupt");
    off[i + 1] = off[i] + counts_all[i];
  }
  RF_REQUIRE(off[g.nseg] >= g.ncells, "MT19937 replay: not enough accepted polar attempts (increase the margin)");
  p->sh_total = off[g.nseg];
  // first cell of every rank's share, and the index in "stream q" (rows of nzl + 1 pairs: q's planes, then the Nyquist plane) of
  // the first stream-q cell at or behind stream cell c
  std::vector<unsigned long long> cb((size_t)P + 1);
  for (int r = 0; r <= P; ++r) {
    const unsigned long long c = off[sh_seg_begin(r, g.nseg, P)];
    cb[r] = c < g.ncells ? c : g.ncells;
  }
  cb[P] = g.ncells;
  auto fq = [&](int q, unsigned long long c) -> unsigned long long {
    const unsigned long long col = c / (unsigned)nzh;
    long long k = (long long)(c - col * (unsigned)nzh) - (long long)q * nzl;
    k = k < 0 ? 0 : (k > nzl ? nzl : k);
    return col * (unsigned)(nzl + 1) + (unsigned long long)k;
  };
  const int me = p->rank;
  p->sh_sendoff.assign(P + 1, 0); p->sh_sendcnt.assign(P, 0); p->sh_recvoff.assign(P, 0); p->sh_recvcnt.assign(P, 0);
  std::vector<long long> sbase(P);
  for (int q = 0; q < P; ++q) {
    p->sh_sendcnt[q] = fq(q, cb[me + 1]) - fq(q, cb[me]);
    p->sh_sendoff[q + 1] = p->sh_sendoff[q] + p->sh_sendcnt[q];
    sbase[q] = (long long)p->sh_sendoff[q] - (long long)fq(q, cb[me]);
    p->sh_recvoff[q] = fq(me, cb[q]);                     // (q = the sending rank here)
    p->sh_recvcnt[q] = fq(me, cb[q + 1]) - fq(me, cb[q]);
  }
  const size_t es = p->sh_single ? 2 * sizeof(float) : 2 * sizeof(double);
  const size_t send_bytes = (size_t)(p->sh_sendoff[P] > 0 ? p->sh_sendoff[P] : 1) * es;
  const size_t recv_pairs = (size_t)p->nx * p->ny * (nzl + 1);
  if (p->mt_send_bytes < send_bytes) {
    if (p->mt_send) RF_HIP(hipFree(p->mt_send));
    p->mt_send = nullptr; p->mt_send_bytes = 0;
    const size_t want = send_bytes + send_bytes / 64;    // (the shares differ from seed to seed by the counts' binomial noise)
    RF_HIP(hipMalloc(&p->mt_send, want));
    p->mt_send_bytes = want;
  }
  if (p->sh_single && p->mt_recv_bytes < recv_pairs * es) {
    if (p->mt_recv) RF_HIP(hipFree(p->mt_recv));
    p->mt_recv = nullptr; p->mt_recv_bytes = 0;
    RF_HIP(hipMalloc(&p->mt_recv, recv_pairs * es));
    p->mt_recv_bytes = recv_pairs * es;
  }
  if (!p->mt_sbase) RF_HIP(hipMalloc((void**)&p->mt_sbase, 64 * sizeof(long long)));
  if (p->mt_first_cap < (size_t)p->sh_nloc) {
    if (p->mt_first) RF_HIP(hipFree(p->mt_first));
    p->mt_first = nullptr; p->mt_first_cap = 0;
    RF_HIP(hipMalloc((void**)&p->mt_first, (size_t)p->sh_nloc * sizeof(unsigned long long)));
    p->mt_first_cap = (size_t)p->sh_nloc;
  }
  hipStream_t s = p->stream;
  RF_HIP(hipMemcpyAsync(p->mt_sbase, sbase.data(), (size_t)P * sizeof(long long), hipMemcpyHostToDevice, s));
  RF_HIP(hipMemcpyAsync(p->mt_first, off.data() + p->sh_first, (size_t)p->sh_nloc * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
  RF_HIP(launch_mt_share_pack(p->sh_single != 0, p->mt_scratch, p->mt_counts, p->mt_first, p->sh_nloc, g.cap, p->mt_send, g.ncells, nzh, nzl, P,
                              p->mt_sbase, s));
  RF_HIP(hipStreamSynchronize(s));                        // (sbase / off are host temporaries)
  p->sh_state = 2;
  return 0;
}

namespace {
// where rank p's stream arrives: the resident deviates themselves (float64) or the float32 staging buffer
inline char* sh_recv_base(rf_plan* p) { return p->sh_single ? (char*)p->mt_recv : (char*)p->noise; }
}

int rf_mt_share_exchange(rf_plan* p) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(p->sh_state == 2, "rf_mt_share_pack must be called first");
  RF_REQUIRE(p->nranks == 1 || p->comm != nullptr, "rf_comm_init has not been called on this multi-rank plan");
  RF_HIP(hipSetDevice(p->device));
  const size_t es = p->sh_single ? 2 * sizeof(float) : 2 * sizeof(double);
  hipStream_t s = p->stream;
  const int me = p->rank;
  RF_HIP(hipMemcpyAsync(sh_recv_base(p) + p->sh_recvoff[me] * es, (const char*)p->mt_send + p->sh_sendoff[me] * es, p->sh_sendcnt[me] * es,
                        hipMemcpyDeviceToDevice, s));
  if (p->nranks > 1) {
    RF_NCCL(g_rccl.GroupStart());
    for (int h = 0; h < p->nranks; ++h) {
      if (h == me) continue;
      if (p->sh_sendcnt[h]) RF_NCCL(g_rccl.Send((const char*)p->mt_send + p->sh_sendoff[h] * es, p->sh_sendcnt[h] * es, ncclUint8, h, p->comm, s));
      if (p->sh_recvcnt[h]) RF_NCCL(g_rccl.Recv(sh_recv_base(p) + p->sh_recvoff[h] * es, p->sh_recvcnt[h] * es, ncclUint8, h, p->comm, s));
    }
    RF_NCCL(g_rccl.GroupEnd());
  }
  p->sh_state = 3;
  return 0;
}

int rf_mt_share_exchange_local(rf_plan** plans, int n) {
  RF_REQUIRE(plans && n >= 1, "null argument");
  for (int g = 0; g < n; ++g) {
    RF_REQUIRE(plans[g] && plans[g]->nranks == n && plans[g]->rank == g, "plans must be ranks 0..n-1 of one n-rank job");
    RF_REQUIRE(plans[g]->device == plans[0]->device, "virtual ranks must live on one device");
    RF_REQUIRE(plans[g]->sh_state == 2 && plans[g]->sh_single == plans[0]->sh_single, "rf_mt_share_pack must have run on every plan (same mode)");
    RF_HIP(hipStreamSynchronize(plans[g]->stream));
  }
  const size_t es = plans[0]->sh_single ? 2 * sizeof(float) : 2 * sizeof(double);
  for (int g = 0; g < n; ++g)        // sender g, receiver h
    for (int h = 0; h < n; ++h) {
      RF_REQUIRE(plans[g]->sh_sendcnt[h] == plans[h]->sh_recvcnt[g], "send / receive counts disagree");
      if (plans[g]->sh_sendcnt[h])
        RF_HIP(hipMemcpy(sh_recv_base(plans[h]) + plans[h]->sh_recvoff[g] * es, (const char*)plans[g]->mt_send + plans[g]->sh_sendoff[h] * es,
                         plans[g]->sh_sendcnt[h] * es, hipMemcpyDeviceToDevice));
    }
  RF_HIP(hipDeviceSynchronize());       // (see rf_slab_exchange_local)
  for (int g = 0; g < n; ++g) plans[g]->sh_state = 3;
  return 0;
}

int rf_mt_share_finish(rf_plan* p, unsigned long long* accepted) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(p->sh_state == 3, "the exchange must have run first");
  RF_HIP(hipSetDevice(p->device));
  if (p->sh_single)
    RF_HIP(launch_mt_share_widen(p->mt_recv, p->noise, (long long)p->nx * p->ny * (p->nzl + 1), p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  p->noise_resident = true;
  p->noise32_resident = false;
  p->sh_state = 0;
  if (accepted) *accepted = p->sh_total;
  return 0;
}

// Same-seed realisations back to back (random.py:24-28 for n seeds): the replay of seed i + 1 (VALU / LDS-bound, second stream)
// runs under the y and z passes of seed i (HBM-bound); ONE set of runs -- the replay of seed i + 1 starts when the generation
// pass of seed i has read them, the generation pass of seed i + 1 when the replay has finished.  complex64 plans with the fast
// generation path (float32 pairs).  states: n x 624 words (mt19937.seed_state); rms_out: n, optional.
int rf_realise_batch_reference(rf_plan* p, const uint32_t* states, int n, double* rms_out) {
  RF_REQUIRE(p && states, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(n >= 1, "need at least one seed");
  RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
  RF_REQUIRE(p->mt_pos && !p->mt_npos.empty(), "rf_mt_set_jump must be called first");
  RF_REQUIRE(p->nranks == 1 && !p->force_slab && !p->generic && !p->f64 && p->have_fast && !p->exact_gen,
             "rf_realise_batch_reference is for single-GPU complex64 plans on the fast generation path; loop rf_noise_mt19937 + rf_realise otherwise");
  RF_HIP(hipSetDevice(p->device));
  MtGeom g;
  if (int rc = mt_geom(p, 1, g)) return rc;
  RF_REQUIRE(g.cap >= 4ull * (unsigned long long)(p->nzc + 1), "the replay's segments are too short for the float32 form on this grid: loop rf_noise_mt19937 + rf_realise");
  if (int rc = mt_ensure_buffers(p, g)) return rc;
  if (int rc = ensure_x(p)) return rc;
  RF_HIP(hipStreamSynchronize(p->stream));
  if (p->stats_cap < n) {
    drop_graphs(p);
    if (p->stats) RF_HIP(hipFree(p->stats));
    p->stats = nullptr;
    RF_HIP(hipMalloc((void**)&p->stats, 2 * (size_t)(n + 64) * sizeof(double)));
    p->stats_cap = n + 64;
  }
  if (!p->aux_stream) {
    RF_HIP(hipStreamCreateWithFlags(&p->aux_stream, hipStreamNonBlocking));
    for (auto& e : p->bev) RF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  // all start states and the per-seed accepted totals live on the device for the length of the batch (the plan keeps the block)
  if (p->br_cap < n) {
    if (p->br_tmp) RF_HIP(hipFree(p->br_tmp));
    p->br_tmp = nullptr; p->br_cap = 0;
    const int cap = n > 16 ? n : 16;
    RF_HIP(hipMalloc(&p->br_tmp, (size_t)cap * (624 * sizeof(uint32_t) + sizeof(unsigned long long) + sizeof(int) + 4)));
    p->br_cap = cap;
  }
  uint32_t* dstates = (uint32_t*)p->br_tmp;
  unsigned long long* dtotals = (unsigned long long*)((char*)p->br_tmp + (size_t)p->br_cap * 624 * sizeof(uint32_t));
  int* dflags = (int*)(dtotals + p->br_cap);
  RF_HIP(hipMemcpy(dstates, states, (size_t)n * 624 * sizeof(uint32_t), hipMemcpyHostToDevice));
  hipStream_t S = p->stream, R = p->aux_stream;
  // whatever deviates were resident are about to be overwritten; the plan claims the new ones (and a field) only once every
  // replay of the batch has been checked
  p->noise_resident = false;
  p->noise32_resident = false;
  p->real_valid = false;
  p->stats_valid = false;
  p->nseg = g.nseg;
  p->seg_cap = g.cap;
  // both streams are drained before any return from here on
  auto drain = [&](int rc) { (void)hipStreamSynchronize(R); (void)hipStreamSynchronize(S); return rc; };
  auto replay = [&](int i) -> int {
    RF_HIP(hipMemcpyAsync(p->mt_states, dstates + (size_t)i * 624, 624 * sizeof(uint32_t), hipMemcpyDeviceToDevice, R));
    if (int r = mt_queue(p, g, 1, R)) return r;
    RF_HIP(hipMemcpyAsync(dtotals + i, p->mt_offsets + g.nseg, sizeof(unsigned long long), hipMemcpyDeviceToDevice, R));
    RF_HIP(hipMemcpyAsync(dflags + i, p->mt_flags, sizeof(int), hipMemcpyDeviceToDevice, R));
    RF_HIP(hipEventRecord(p->bev[0], R));
    return 0;
  };
  auto issue = [&]() -> int {
    RF_HIP(hipEventRecord(p->ev[0], S));
    RF_HIP(hipEventRecord(p->bev[1], S));
    RF_HIP(hipStreamWaitEvent(R, p->bev[1], 0));          // (whatever ran on the main stream before the batch has finished with the runs)
    if (int rc = replay(0)) return rc;
    for (int i = 0; i < n; ++i) {
      RF_HIP(hipStreamWaitEvent(S, p->bev[0], 0));        // the runs of seed i are complete
      p->resident_fast = true;
      p->noise32_resident = true;                         // (queue_x selects the float32-pair kernel by it; cleared again on failure)
      const bool xp = p->X && xpose_ok(p);
      const int rc = queue_x(p, make_gen(p, 0, RF_NOISE_RESIDENT, false), nullptr, xp ? p->X : p->W, S, false);
      p->resident_fast = false;
      if (rc) return rc;
      RF_HIP(hipEventRecord(p->bev[1], S));               // the generation pass of seed i has read the runs
      if (i + 1 < n) {
        RF_HIP(hipStreamWaitEvent(R, p->bev[1], 0));
        if (int rc2 = replay(i + 1)) return rc2;
      }
      if (int rc3 = queue_yz(p, p->W, S, p->stats + 2 * i, false)) return rc3;
    }
    RF_HIP(hipEventRecord(p->ev[4], S));
    return 0;
  };
  std::vector<unsigned long long> totals((size_t)n);
  std::vector<int> flags((size_t)n);
  std::vector<double> st(2 * (size_t)n);
  int rc = issue();
  if (!rc) {
    hipError_t e = hipStreamSynchronize(R);
    if (e == hipSuccess) e = hipMemcpyAsync(st.data(), p->stats, st.size() * sizeof(double), hipMemcpyDeviceToHost, S);
    if (e == hipSuccess) e = hipStreamSynchronize(S);
    if (e == hipSuccess) e = hipMemcpy(totals.data(), dtotals, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(flags.data(), dflags, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(2, std::string("rf_realise_batch_reference: ") + hipGetErrorString(e));
  }
  if (!rc)
    for (int i = 0; i < n && !rc; ++i) {
      if (totals[i] < g.ncells) rc = fail(1, "MT19937 replay: not enough accepted polar attempts (increase the margin)");
      else if (flags[i] & 1) rc = fail(1, "MT19937 replay: a segment holds fewer deviate pairs than a row of the grid has cells");
    }
  if (rc) {
    p->noise32_resident = false;
    return drain(rc);
  }
  p->noise32_resident = true;                             // the last seed's deviates, as float32 pairs in the runs
  p->cur = p->W; p->timed = false; p->real_valid = true; p->stats_valid = true; p->stats_slot = n - 1; p->k_valid = false;
  if (rms_out) {
    const double cnt = (double)p->nx * p->ny * p->nz;
    for (int i = 0; i < n; ++i) {
      const double m = st[2 * i] / cnt, v = st[2 * i + 1] / cnt - m * m;
      rms_out[i] = v > 0 ? std::sqrt(v) : 0.0;
    }
  }
  return 0;
}

int rf_can_batch_reference(rf_plan* p) {
  if (!p || p->unpacked || p->nranks != 1 || p->force_slab || p->generic || p->f64 || !p->have_fast || p->exact_gen || !p->have_kgrid ||
      !p->have_power || !p->mt_pos || p->mt_npos.empty())
    return 0;
  MtGeom g;
  if (mt_geom(p, 1, g)) return 0;
  return g.cap >= 4ull * (unsigned long long)(p->nzc + 1) ? 1 : 0;
}

int rf_download_noise(rf_plan* p, double* host, unsigned long long first, unsigned long long count) {
  RF_REQUIRE(p && host, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->noise_resident, "no float64 deviates resident on the device");
  RF_REQUIRE(first + count <= 2ull * p->nx * p->ny * (p->nzl + 1), "range outside the noise buffer");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipMemcpyAsync(host, p->noise + first, count * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}
}  // extern "C"
