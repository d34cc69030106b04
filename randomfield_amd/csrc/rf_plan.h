// rf_plan.h -- the plan object behind the C-ABI and the helpers its translation units share (rf_capi.hip: plans, inputs,
// realisations, transforms, host <-> device; rf_capi_mt.hip: the MT19937 replay and its sharing between ranks; rf_capi_slab.hip: the
// communicator and the slab pipeline in separate steps).  Internal: nothing here crosses the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/randomfield_hip.h"
#include "../../include/randomfield_hip_diag.h"
#include "rf_host.h"
#include "rf_launch.h"

namespace rfc {
using namespace rf;


extern thread_local std::string g_err;      // last error message of the calling thread (rf_last_error)
int fail(int code, const std::string& msg);

#define RF_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(2, std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                         std::to_string(__LINE__) + ")");                                         \
  } while (0)

#define RF_REQUIRE(cond, msg) \
  do {                        \
    if (!(cond)) return fail(1, msg); \
  } while (0)

// RCCL entry points, resolved lazily with dlopen so that single-GPU use never depends on librccl
// being loadable (types and enums come from <rccl/rccl.h>, no link-time dependency).
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
extern Rccl g_rccl;
int load_rccl();

#define RF_NCCL(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess)                                                                              \
      return fail(5, std::string(#expr) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?")); \
  } while (0)

}  // namespace rfc

struct rf_plan {
  int nx = 0, ny = 0, nz = 0, nzc = 0, f64 = 0, device = 0, nranks = 1, rank = 0;
  int nxl = 0, nzl = 0, kz0 = 0;          // this rank's x-slab height, kz-slab width and first kz plane
  void* R = nullptr;                      // receive buffer of the all-to-all (slab-path plans only)
  void *W2 = nullptr, *R2 = nullptr;      // second buffer pair of pipelined slab batches
  hipStream_t comm_stream = nullptr;      // exchange stream of pipelined slab batches
  hipEvent_t pev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // fwd[2], exch[2], z[2]
  bool force_slab = false;                // single-rank plan routed through the slab pipeline (tests)
  int standin_read_pct = 100, standin_write_pct = 100;   // ... and the share of the blocks it reads / writes (rf_slab_set_exchange_standin_ex)
  int standin_wg = 0;                     // rf_slab_set_exchange_standin: workgroups of the copy kernel that stands in for the all-to-all of a rank without a communicator
  // RF_FLAG_EXCHANGE_CHUNKS: the rank's kz slab as `xchunks` sub-slabs of nzl / xchunks planes, each generated, x- and y-transformed
  // and SENT on its own, so that the exchange of sub-slab c runs under the forward passes of sub-slab c + 1 (queue_c2r).  Layout:
  // W = [chunk][nx][ny][nzl / xchunks]; R = [source rank][chunk][nxl][ny][nzl / xchunks], which is what the gathering z pass reads
  // anyway with nranks * xchunks segments per row
  int xchunks = 1;
  std::vector<hipEvent_t> chunk_ev;       // forward half of chunk c queued (no timing)
  bool replicate = false;                 // multi-rank plan without an exchange: every rank generates all of k space (see queue_x)
  // "direct" exchange: the y pass stores its output tiles straight into the receive buffers of the ranks that own
  // their x planes (rf_fft.h DirectColIO) -- peer-mapped pointers between processes (rf_comm_enable_direct), plain device pointers between
  // virtual ranks (rf_slab_link_direct) -- and one tiny all-reduce per realisation is the barrier between the peers' stores and the z pass
  bool direct = false;
  std::vector<void*> peer_R[2];           // host: every rank's R and R2 as THIS process addresses them ([rank] = its own)
  void** peer_tab = nullptr;              // device: [2 buffers][chunks][nranks] destination bases, shifted as DirectColIO wants them
  int peer_tab_chunks = 0;                // ... built for this many exchange chunks
  std::vector<void*> ipc_open;            // peer mappings this process opened (closed at destroy)
  bool direct_standin = false;            // rf_slab_set_direct_standin: the stores land in this rank's own buffers (no field comes out)
  int direct_overlap = 1;                 // batches: 1 = the storing y pass on the exchange stream beside the neighbours' x / z passes, 0 = everything on one stream
  ncclComm_t comm = nullptr;
  size_t csize = 8;                       // bytes per complex element
  hipStream_t own_stream = nullptr, stream = nullptr;
  void* W = nullptr;                      // [nx][ny][nz] real == [nx][ny][nz/2] complex (packed Nyquist)
  void* K = nullptr;                      // lazy: API-layout k-space [nx][ny][nz/2+1]
  // lazy: the x pass's TRANSPOSED intermediate [kz tile][ny][nx][tile width]: the x pass stores whole
  // contiguous tiles there and the y pass goes X -> W out of place.  xposed = the plan may use it (RF_FLAG_TRANSPOSED_INTERMEDIATE).
  void* X = nullptr;
  bool xposed = false;
  // y and z passes slab by slab of x planes: -1 = automatic (slabs of about the Infinity Cache's size),
  // 0 = whole-grid passes, > 0 = this many x planes per slab (RF_FLAG_YZ_SLAB_PLANES)
  int yz_slab = -1;
  // rf_set_host_sink: the NEXT single-rank realisation delivers its field to host memory slab by slab, each slab's device -> host copy
  // on dl_stream behind that slab's z pass while the GPU runs the next slab (generate.py:184-189,230 returns a host array: the copy is
  // 15 x the realisation)
  void* sink_host = nullptr;              // armed destination (one shot), RF_LAYOUT_DENSE / RF_LAYOUT_PADDED rows
  int sink_layout = 0;
  bool sink_delivered = false;            // the last armed call has delivered
  hipStream_t dl_stream = nullptr;
  std::vector<hipEvent_t> sink_ev;
  hipStream_t aux_stream = nullptr;       // rf_realise_batch_reference: the stream the MT19937 replays run on
  hipEvent_t bev[2] = {nullptr, nullptr}; // ... replay finished / generation pass has read the runs
  std::vector<hipEvent_t> slab_ev;        // timed runs: after y(i), after z(i)
  int slab_timed = 0;                     // slabs of the last timed run (0: whole-grid passes, ev[2] / ev[3] apply)
  int yz_merge = 1;                       // rf_set_merged_yz: 0 never, 1 untimed calls (default), 2 timed calls too (events per launch)
  int slab_merged = 0;                    // slabs of the last timed run that used merged launches: slab_ev = after y(0), after every merged launch, after the last z
  void* P = nullptr;                      // lazy: saved potential, API layout (= P_base + an offset chosen by ensure_p)
  void* P_base = nullptr;                 // the allocation P lives in
  size_t w_bytes = 0, k_bytes = 0, p_bytes = 0;      // field buffer, k-space side array, potential array (padded rows)
  int ppitch = 0;                         // cells per row of the potential array: nzl + 1, rounded up to even on float32 plans
  void *tw_x = nullptr, *tw_y = nullptr, *tw_z = nullptr;
  double *kx2 = nullptr, *ky2 = nullptr, *kz2 = nullptr;
  double *xt = nullptr, *st = nullptr, *sl = nullptr;
  int* bin = nullptr;
  int nt = 0, nbins = 0;
  double x0 = 0, inv_dx = 0;
  bool have_kgrid = false, have_power = false;
  // fast float32 native generation: float copies of the k^2 tables + per-bin sigma records
  rf::FastRec* frec = nullptr;
  int fnbins = 0;
  float fdkx = 0, fdky = 0, fdkz = 0, fu_scale = 0, fu_off = 0;
  bool have_fast = false, exact_gen = false;
  std::vector<double> h_kx2, h_ky2, h_kz2;   // host copies (k range of the grid for the fast records)
  rf::SigmaTableHost h_tab;
  double* noise = nullptr;
  size_t noise_cap = 0;
  bool noise_resident = false;            // the device noise buffer holds a full set of deviates
  // float32 deviates (rf_noise_mt19937_ex(single = 1)) stay where the one-pass replay writes them: mt_scratch, every
  // segment's accepted pairs from slot seg * seg_cap, located through mt_offsets (FastGenParams::seg_*).  Only one of
  // the two forms (float64 in cell order / float32 in segment order) is valid at a time.
  bool noise32_resident = false;
  unsigned long long seg_cap = 0;
  int nseg = 0;
  void* mt_rowtab = nullptr;           // float32 form: where each row (ix, iy) of the stream starts in the runs (rf_core.h RowLoc, 8 B x nx ny)
  int* mt_flags = nullptr;             // device word: bit 0 = a row spans more than two segments (mt_rowtab_kernel)
  void* fixbuf = nullptr;              // nx * ny complex: the repaired kz = 0 slots of the fast generation pass (fix_fill_kernel)
  // MT19937 replay (rf_noise_mt19937): jump-polynomial bit positions per tree level, scratch
  uint32_t* mt_pos = nullptr;          // set-bit positions of the jump polynomials, widened to 32 bits (scalar loads)
  std::vector<int> mt_npos;
  int mt_stride = 0, mt_bps = 0, mt_radix = 2;   // positions per polynomial (padded), blocks of 624 words per segment, tree radix
  int* mt_npos_dev = nullptr;
  uint32_t* mt_states = nullptr;
  unsigned long long *mt_counts = nullptr, *mt_offsets = nullptr;
  size_t mt_states_cap = 0, mt_seg_cap = 0;
  void* mt_scratch = nullptr;          // one-pass replay: every segment's accepted pairs, densely from slot seg * (attempts per segment)
  size_t mt_scratch_bytes = 0;
  // distributed replay (rf_mt_share_*): this rank replays segments [sh_first, sh_first + sh_nloc) of the one stream
  void *mt_send = nullptr, *mt_recv = nullptr;   // pairs packed by destination rank / stream of this rank as received (float32 mode)
  size_t mt_send_bytes = 0, mt_recv_bytes = 0;
  long long* mt_sbase = nullptr;                 // device, [nranks]: see mt_share_pack_kernel
  unsigned long long* mt_first = nullptr;        // device: first stream cell of every local segment
  size_t mt_first_cap = 0;
  int sh_state = 0;                              // 0 idle, 1 replayed (begin), 2 packed, 3 exchanged
  int sh_single = 0, sh_first = 0, sh_nloc = 0;
  unsigned long long sh_total = 0;
  std::vector<unsigned long long> sh_sendoff, sh_sendcnt, sh_recvoff, sh_recvcnt;      // pairs, per peer
  double* partials = nullptr;
  long long npartials = 0;
  double* stats = nullptr;                // [2 * stats_cap] (sum, sumsq) per realisation
  int stats_cap = 0;
  uint64_t* seeds_dev = nullptr;
  int seeds_cap = 0;
  // the caller's seed array may be a temporary: it is copied into one of two plan-owned pinned staging slots before
  // the asynchronous upload (a slot is reused only after the upload that last read it has completed)
  uint64_t* seeds_pin[2] = {nullptr, nullptr};
  hipEvent_t seeds_ev[2] = {nullptr, nullptr};
  int seeds_pin_cap = 0, seeds_turn = 0;
  bool resident_fast = false;             // the current call draws from the device-resident deviates (RF_NOISE_RESIDENT)
  bool emit_potential = false;            // the current call transforms emit_pscale * delta(k) / k^2 instead of delta(k) (rf_realise_scaled_potential)
  double emit_pscale = 0.0;
  const double* zscale = nullptr;         // the current call's z pass multiplies plane z by zscale[z] (device table: ztab)
  void* pot_target = nullptr;             // non-null while rf_realise_potential queues its x pass: where delta(k)/k^2 goes
  double* coll_scratch = nullptr;         // 2 doubles on the device for host-side all-reduces (never aliases `stats`)
  void* br_tmp = nullptr;                 // rf_realise_batch_reference: [start states n x 624][accepted totals n][flags n], kept between calls
  int br_cap = 0;                         // (allocating and freeing them cost a device synchronisation per call: one-seed batches are the Generator's call)
  double* ztab = nullptr;                 // 2 * nz doubles for lognormal / affine tables
  // fused lognormal realisations (rf_realise_lognormal): [growth nz][density nz][A nz][B nz][sigma 8] and the y pass's Parseval partials
  double* lntab = nullptr;
  double* ypart = nullptr;
  long long nypart = 0;
  bool ln_tables = false, ln_density = false;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // start, after x, y, z, reduce; [5] = after the kz = 0 repair launch
  bool repair_timed = false;
  bool aux_valid = false;              // the k buffer's memory currently holds an auxiliary REAL field (lensing potential)
  bool unpacked = false;               // c2c plan: W is the full [nx][ny][nz] complex array, only rf_*_c / rf_execute_c2c apply
  // non-power-of-two grid (rf_generic.h): the transforms run on API-layout arrays, K -> G -> W; no fused generation,
  // no graphs, one rank.  gax / gay factor nx / ny, gaz factors nz/2 (packed plans) or nz (c2c plans)
  bool generic = false;
  void* G = nullptr;                   // lazy scratch [nx][ny][nz/2+1] complex (c2c plans: [nx][ny][nz], for a long axis)
  void* G2 = nullptr;                  // lazy second scratch: only when an axis is too long for one line (four-step form, rf_generic.h)
  rf::GenericAxis gax, gay, gaz;
  rf::GenericDims gdims;               // the same + the split of the long axes, as the sequences of rf_generic.h take them
  bool timed = false;
  struct BatchGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
  std::map<int, BatchGraph> graphs;       // captured batch graphs, keyed by the number of realisations
  bool real_valid = false, k_valid = false, stats_valid = false;
  void* cur = nullptr;                    // buffer holding the current real-space field
  int stats_slot = 0;                     // which (sum, sumsq) pair of `stats` belongs to the current field                    // x-planes per y/z slab (0 = whole grid in one launch pair)
};

namespace rfc {
// (defined in rf_capi.hip)
void drop_graphs(rf_plan* p);
int ensure_k(rf_plan* p);
int ensure_x(rf_plan* p);
int ensure_noise(rf_plan* p);
bool xpose_ok(const rf_plan* p);
int slab_chunks(const rf_plan* p);
GenParams make_gen(rf_plan* p, uint64_t seed, int mode, bool seed_from_dev);
int upload_noise(rf_plan* p, int mode, const double* noise_host);
int queue_x(rf_plan* p, const GenParams& gp, const void* kspace, void* W, hipStream_t sx, bool timed = false, int kz0c = -1, int nzlc = -1);
int queue_xy(rf_plan* p, const GenParams& gp, const void* kspace, void* W, hipStream_t s, bool timed, int rbuf = 0);
int ensure_batch_buffers(rf_plan* p);
int rebuild_peer_tab(rf_plan* p);
int direct_barrier(rf_plan* p, hipStream_t s);
int queue_yz(rf_plan* p, void* W, hipStream_t s, double* stats_out, bool timed);
int queue_z_slab(rf_plan* p, const void* R, void* W, double* stats_out, hipStream_t s);
int queue_r2c_slab_rows(rf_plan* p, hipStream_t s);
int queue_r2c_slab_cols(rf_plan* p, hipStream_t s);
int potential_forward(rf_plan* p, uint64_t seed, int mode, const double* noise_host, bool whole);
}  // namespace rfc
