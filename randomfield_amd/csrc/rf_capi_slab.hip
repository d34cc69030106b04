// rf_capi_slab.hip -- C-ABI entry points of the multi-GPU plumbing: the RCCL communicator (dlopen'ed) and the slab pipeline in
// separate steps with virtual ranks (diagnostics, include/randomfield_hip_diag.h).
#include "rf_plan.h"

using namespace rfc;

extern "C" {

int rf_comm_unique_id(void* id128) {
  RF_REQUIRE(id128, "null argument");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  RF_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int rf_comm_init(rf_plan* p, const void* id128) {
  RF_REQUIRE(p && id128, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->comm == nullptr, "communicator already initialised");
  if (int rc = load_rccl()) return rc;
  RF_HIP(hipSetDevice(p->device));
  // RCCL inspects hipGetLastError(): make sure no stale (non-sticky) error of an earlier call is pending
  {
    hipError_t stale = hipGetLastError();
    if (stale != hipSuccess && getenv("RANDOMFIELD_DEBUG"))
      fprintf(stderr, "rf_comm_init: cleared stale HIP error: %s\n", hipGetErrorString(stale));
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  RF_NCCL(g_rccl.CommInitRank(&p->comm, p->nranks, id, p->rank));
  // one tiny collective now: a broken communicator should fail here, not inside a timed region
  RF_HIP(hipMemsetAsync(p->coll_scratch, 0, 4 * sizeof(double), p->stream));
  p->standin_wg = 0;                       // (a stand-in exchange is a thing of ranks without a communicator)
  RF_NCCL(g_rccl.AllReduce(p->coll_scratch, p->coll_scratch, 2, ncclFloat64, ncclSum, p->comm, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

// ranks of the plan's RCCL communicator as RCCL itself counts them (ncclCommCount); 0 = no communicator (rf_comm_init has not run)
int rf_comm_size(rf_plan* p, int* nranks) {
  RF_REQUIRE(p && nranks, "null argument");
  *nranks = 0;
  if (!p->comm) return 0;
  RF_NCCL(g_rccl.CommCount(p->comm, nranks));
  return 0;
}

int rf_comm_allreduce_f64(rf_plan* p, double* inout, int n, int op) {
  RF_REQUIRE(p && inout, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(n >= 1 && n <= 2, "n must be 1 or 2");
  RF_REQUIRE(op == 0 || op == 1, "op must be 0 (sum) or 1 (max)");
  RF_HIP(hipSetDevice(p->device));
  if (!p->comm) { RF_HIP(hipStreamSynchronize(p->stream)); return 0; }     // a one-rank communicator still runs the collective
  if (p->comm_stream) RF_HIP(hipStreamSynchronize(p->comm_stream));          // (the communicator is used from one stream at a time)
  double* d = p->coll_scratch;            // its own two doubles: `stats` holds the moments of up to stats_cap realisations
  RF_HIP(hipMemcpyAsync(d, inout, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
  RF_NCCL(g_rccl.AllReduce(d, d, n, ncclFloat64, op == 0 ? ncclSum : ncclMax, p->comm, p->stream));
  RF_HIP(hipMemcpyAsync(inout, d, n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

/* ---- slab pipeline in separate steps (tests / custom exchanges) ------------------------------- */
int rf_slab_forward(rf_plan* p, uint64_t seed, int mode, const double* noise_host) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->nranks > 1, "rf_slab_* are for multi-rank plans");
  RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = upload_noise(p, mode, noise_host)) return rc;
  p->resident_fast = (mode == RF_NOISE_RESIDENT);
  const int rc = queue_xy(p, make_gen(p, seed, mode, false), nullptr, p->W, p->stream, false);
  p->resident_fast = false;
  if (rc) return rc;
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

// the forward half with the other two sources of rf_realise_potential / rf_execute_c2r
int rf_slab_forward_ex(rf_plan* p, uint64_t seed, int mode, const double* noise_host, int source) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->nranks > 1, "rf_slab_* are for multi-rank plans");
  RF_REQUIRE(source == RF_SLAB_GENERATE || source == RF_SLAB_GENERATE_SAVE_POTENTIAL || source == RF_SLAB_FROM_KSPACE, "invalid source");
  if (source == RF_SLAB_GENERATE) return rf_slab_forward(p, seed, mode, noise_host);
  RF_HIP(hipSetDevice(p->device));
  if (source == RF_SLAB_GENERATE_SAVE_POTENTIAL) {
    if (int rc = potential_forward(p, seed, mode, noise_host, false)) return rc;
  } else {
    RF_REQUIRE(p->K && p->k_valid, "no k-space data: call rf_generate, rf_load_potential or rf_upload_k first");
    if (int rc = queue_xy(p, make_gen(p, 0, RF_NOISE_NATIVE, false), p->K, p->W, p->stream, false)) return rc;
  }
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

int rf_slab_exchange_local(rf_plan** plans, int n) {
  RF_REQUIRE(plans && n >= 1, "null argument");
  for (int g = 0; g < n; ++g) {
    RF_REQUIRE(plans[g] && plans[g]->nranks == n && plans[g]->rank == g, "plans must be ranks 0..n-1 of one n-rank job");
    RF_REQUIRE(plans[g]->device == plans[0]->device, "virtual ranks must live on one device");
    RF_HIP(hipStreamSynchronize(plans[g]->stream));
  }
  const rf_plan* p0 = plans[0];
  const int C = slab_chunks(p0);
  for (int g = 0; g < n; ++g) RF_REQUIRE(slab_chunks(plans[g]) == C, "every rank must use the same number of exchange chunks");
  const size_t blk = (size_t)p0->nxl * p0->ny * p0->nzl * p0->csize / (size_t)C, cb = p0->w_bytes / (size_t)C;
  for (int g = 0; g < n; ++g)        // sender g, receiver h: block h of sub-slab c of W_g -> segment (g, c) of R_h
    for (int h = 0; h < n; ++h)
      for (int c = 0; c < C; ++c)
        RF_HIP(hipMemcpy((char*)plans[h]->R + ((size_t)g * C + c) * blk, (char*)plans[g]->W + (size_t)c * cb + (size_t)h * blk, blk, hipMemcpyDeviceToDevice));
  // a device-to-device hipMemcpy may return before the copy has run (it is only ordered on the null stream), and the
  // plans' streams do not synchronise with the null stream: without this the gathering z pass of a large grid read
  // blocks that had not arrived yet (caught by the full-size config-4 test; small grids happened to win the race)
  RF_HIP(hipDeviceSynchronize());
  return 0;
}

// the multi-rank forward transform in separate steps (virtual ranks): rows on the x slab, rf_slab_exchange_local_reverse, columns
int rf_slab_r2c_rows(rf_plan* p) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked && !p->generic, "this call applies to packed plans on the tiled kernels");
  RF_REQUIRE(p->nranks > 1, "rf_slab_* are for multi-rank plans");
  RF_REQUIRE(p->real_valid && p->cur == p->W, "no real-space field on the device: call rf_upload_real (or a c2r) first");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = queue_r2c_slab_rows(p, p->stream)) return rc;
  p->real_valid = false;
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}
int rf_slab_exchange_local_reverse(rf_plan** plans, int n) {
  RF_REQUIRE(plans && n >= 1, "null argument");
  for (int g = 0; g < n; ++g) {
    RF_REQUIRE(plans[g] && plans[g]->nranks == n && plans[g]->rank == g, "plans must be ranks 0..n-1 of one n-rank job");
    RF_REQUIRE(plans[g]->device == plans[0]->device, "virtual ranks must live on one device");
    RF_HIP(hipStreamSynchronize(plans[g]->stream));
  }
  const rf_plan* p0 = plans[0];
  const size_t blk = (size_t)p0->nxl * p0->ny * p0->nzl * p0->csize;
  for (int h = 0; h < n; ++h)        // sender h (x slab), receiver g (kz slab): block g of R_h -> block h of W_g
    for (int g = 0; g < n; ++g)
      RF_HIP(hipMemcpy((char*)plans[g]->W + h * blk, (const char*)plans[h]->R + g * blk, blk, hipMemcpyDeviceToDevice));
  RF_HIP(hipDeviceSynchronize());       // (see rf_slab_exchange_local)
  return 0;
}
int rf_slab_r2c_cols(rf_plan* p) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked && !p->generic, "this call applies to packed plans on the tiled kernels");
  RF_REQUIRE(p->nranks > 1, "rf_slab_* are for multi-rank plans");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = ensure_k(p)) return rc;
  if (int rc = queue_r2c_slab_cols(p, p->stream)) return rc;
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

/* ---- the direct exchange: the y pass stores into the peers' receive buffers ------------------- */
namespace {
// forget the peers (a failed set-up, or switching the mode off): unmap what this process mapped
int direct_reset(rf_plan* p) {
  RF_HIP(hipStreamSynchronize(p->stream));
  if (p->comm_stream) RF_HIP(hipStreamSynchronize(p->comm_stream));
  p->direct = false;
  for (void* m : p->ipc_open) (void)hipIpcCloseMemHandle(m);
  p->ipc_open.clear();
  p->peer_R[0].clear(); p->peer_R[1].clear();
  if (p->peer_tab) { RF_HIP(hipFree(p->peer_tab)); p->peer_tab = nullptr; }
  p->peer_tab_chunks = 0;
  return 0;
}
bool direct_shape_ok(const rf_plan* p) {
  return !p->unpacked && !p->generic && col_direct_supported(p->f64, p->ny, p->nzl / slab_chunks(p));
}
// What a rank tells the others about its two receive buffers: [2 IPC handles][ok flag, 8 bytes][PCI bus id of its device, 32 bytes], RF_DIRECT_RECORD_BYTES per rank
constexpr size_t DIRECT_REC = 192;
static_assert(2 * sizeof(hipIpcMemHandle_t) + 8 + 32 <= DIRECT_REC, "record too small for two IPC handles, the flag and the PCI bus id");
// this rank's record (all zero = "not here": shape without a storing y pass, no buffers, no handles)
bool direct_fill_record(rf_plan* p, unsigned char* mine) {
  memset(mine, 0, DIRECT_REC);
  if (!(direct_shape_ok(p) && !p->replicate) || ensure_batch_buffers(p) != 0) return false;
  hipIpcMemHandle_t hd[2];
  if (hipIpcGetMemHandle(&hd[0], p->R) != hipSuccess || hipIpcGetMemHandle(&hd[1], p->R2) != hipSuccess) { (void)hipGetLastError(); return false; }
  memcpy(mine, hd, sizeof(hd));
  mine[2 * sizeof(hipIpcMemHandle_t)] = 1;
  // where the buffers live, so that a peer can ask the runtime BEFORE it maps them whether its device reaches this one at all
  char* bus = (char*)mine + 2 * sizeof(hipIpcMemHandle_t) + 8;
  if (hipDeviceGetPCIBusId(bus, 32, p->device) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
  return true;
}
// map the two buffers of every peer named in rec[P][DIRECT_REC]; false (and nothing left mapped beyond p->ipc_open, which direct_reset closes) on any failure
bool direct_map_peers(rf_plan* p, const unsigned char* rec, std::vector<void*>& R0, std::vector<void*>& R1) {
  const int P = p->nranks;
  for (int h = 0; h < P; ++h)
    if (rec[DIRECT_REC * h + 2 * sizeof(hipIpcMemHandle_t)] != 1) return false;
  R0.assign(P, nullptr); R1.assign(P, nullptr);
  R0[p->rank] = p->R; R1[p->rank] = p->R2;
  for (int h = 0; h < P; ++h) {
    if (h == p->rank) continue;
    hipIpcMemHandle_t ph[2];
    memcpy(ph, rec + DIRECT_REC * h, sizeof(ph));
    // a peer on another device this process can see: no peer access between the two, no direct exchange (a store through such a
    // mapping would fault, not fail).  A device this process cannot see (masked by HIP_VISIBLE_DEVICES) is left to hipIpcOpenMemHandle.
    char bus[32];
    memcpy(bus, rec + DIRECT_REC * h + 2 * sizeof(hipIpcMemHandle_t) + 8, sizeof(bus));
    bus[31] = 0;
    int pdev = -1, can = 0;
    if (bus[0] && hipDeviceGetByPCIBusId(&pdev, bus) == hipSuccess && pdev >= 0 && pdev != p->device) {
      if (hipDeviceCanAccessPeer(&can, p->device, pdev) != hipSuccess || !can) { (void)hipGetLastError(); return false; }
    } else {
      (void)hipGetLastError();
    }
    for (int b = 0; b < 2; ++b) {
      void* m = nullptr;
      if (hipIpcOpenMemHandle(&m, ph[b], hipIpcMemLazyEnablePeerAccess) != hipSuccess || !m) { (void)hipGetLastError(); return false; }
      p->ipc_open.push_back(m);
      (b ? R1 : R0)[h] = m;
    }
  }
  return true;
}
// min over the communicator's ranks of a flag (1 = fine here), on the plan's stream
int agree(rf_plan* p, bool mine, bool* all) {
  double v[2] = {mine ? 1.0 : 0.0, 0.0};
  RF_HIP(hipMemcpyAsync(p->coll_scratch, v, sizeof(v), hipMemcpyHostToDevice, p->stream));
  RF_NCCL(g_rccl.AllReduce(p->coll_scratch, p->coll_scratch, 2, ncclFloat64, ncclMin, p->comm, p->stream));
  RF_HIP(hipMemcpyAsync(v, p->coll_scratch, sizeof(v), hipMemcpyDeviceToHost, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  *all = v[0] > 0.5;
  return 0;
}
}  // namespace

// COLLECTIVE over the plan's communicator.  enable != 0: every rank publishes the IPC handles of its two receive buffers (one
// ncclAllGather), maps its peers' buffers (hipIpcOpenMemHandle, peer access enabled lazily), proves the mapping with a marker that every
// rank stores into every peer's buffer from a kernel and every rank then finds in its own, and only if ALL ranks succeeded switches
// the plan to the direct exchange: *enabled = 1.  Any failure anywhere (no IPC between these devices, a shape whose y-pass tiles
// straddle x planes ...) leaves every rank on the grouped ncclSend / ncclRecv exchange, *enabled = 0, return value 0 -- never a
// job where some ranks store directly and others wait in a receive.  enable == 0: back to the RCCL exchange (also collective).
int rf_comm_enable_direct(rf_plan* p, int enable, int* enabled) {
  RF_REQUIRE(p && enabled, "null argument");
  *enabled = 0;
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->nranks > 1 ? p->comm != nullptr : p->force_slab, "rf_comm_enable_direct needs rf_comm_init first (or, on one rank, RF_FLAG_FORCE_SLAB_PATH)");
  RF_HIP(hipSetDevice(p->device));
  drop_graphs(p);
  if (!enable) return direct_reset(p);
  if (p->direct) { *enabled = 1; return 0; }
  if (int rc = direct_reset(p)) return rc;
  const int P = p->nranks;
  bool ok = direct_shape_ok(p) && !p->replicate;
  if (ok && ensure_batch_buffers(p) != 0) ok = false;
  if (P == 1) {                                  // the forced slab path of a single rank: its own buffers are all there is
    if (!ok) return 0;
    p->peer_R[0].assign(1, p->R); p->peer_R[1].assign(1, p->R2);
    if (int rc = rebuild_peer_tab(p)) return rc;
    p->direct = true; *enabled = 1;
    return 0;
  }
  // 1. handles of R and R2, gathered from every rank
  constexpr size_t REC = DIRECT_REC;
  std::vector<unsigned char> rec(REC * P, 0);
  unsigned char* mine = rec.data() + REC * p->rank;
  ok = ok && direct_fill_record(p, mine);
  unsigned char* dev = nullptr;
  RF_HIP(hipMalloc((void**)&dev, REC * P));
  auto done = [&](int rc) { (void)hipFree(dev); return rc; };
  if (hipMemcpyAsync(dev + REC * p->rank, mine, REC, hipMemcpyHostToDevice, p->stream) != hipSuccess) return done(fail(2, "hipMemcpyAsync of the IPC handles failed"));
  if (g_rccl.AllGather(dev + REC * p->rank, dev, REC, ncclUint8, p->comm, p->stream) != ncclSuccess) return done(fail(5, "ncclAllGather of the IPC handles failed"));
  if (hipMemcpyAsync(rec.data(), dev, REC * P, hipMemcpyDeviceToHost, p->stream) != hipSuccess || hipStreamSynchronize(p->stream) != hipSuccess)
    return done(fail(2, "reading back the gathered IPC handles failed"));
  // 2. map the peers' buffers
  std::vector<void*> R0, R1;
  ok = ok && direct_map_peers(p, rec.data(), R0, R1);
  bool all = false;
  if (int rc = agree(p, ok, &all)) return done(rc);
  if (!all) { (void)hipFree(dev); return direct_reset(p); }
  // 3. prove it: rank g stores the marker (tag + g) into slot g of every rank's two buffers, from a kernel, as the y pass will;
  //    behind a barrier every rank must find all P markers in its own buffers (which are scratch between realisations)
  p->peer_R[0] = R0; p->peer_R[1] = R1;
  void** tabs = nullptr;
  if (hipMalloc((void**)&tabs, 2 * P * sizeof(void*)) != hipSuccess) return done(fail(2, "hipMalloc failed"));
  std::vector<void*> both(R0); both.insert(both.end(), R1.begin(), R1.end());
  const unsigned long long tag = 0x5246444952000000ull;          // "RFDIR"
  bool good = hipMemsetAsync(p->R, 0, 8 * P, p->stream) == hipSuccess && hipMemsetAsync(p->R2, 0, 8 * P, p->stream) == hipSuccess &&
              hipMemcpyAsync(tabs, both.data(), 2 * P * sizeof(void*), hipMemcpyHostToDevice, p->stream) == hipSuccess;
  if (int rc = direct_barrier(p, p->stream)) { (void)hipFree(tabs); return done(rc); }          // every rank has cleared its slots
  good = good && launch_peer_mark(tabs, 2 * P, p->rank, tag + (unsigned)p->rank, p->stream) == hipSuccess;
  if (int rc = direct_barrier(p, p->stream)) { (void)hipFree(tabs); return done(rc); }          // every rank's markers are on their way ... and have landed
  std::vector<unsigned long long> got(2 * P, 0);
  good = good && hipMemcpyAsync(got.data(), p->R, 8 * P, hipMemcpyDeviceToHost, p->stream) == hipSuccess &&
         hipMemcpyAsync(got.data() + P, p->R2, 8 * P, hipMemcpyDeviceToHost, p->stream) == hipSuccess && hipStreamSynchronize(p->stream) == hipSuccess;
  for (int g = 0; g < P && good; ++g) good = got[g] == tag + (unsigned)g && got[P + g] == tag + (unsigned)g;
  (void)hipFree(tabs);
  (void)hipGetLastError();
  if (int rc = agree(p, good, &all)) return done(rc);
  (void)hipFree(dev);
  if (!all) return direct_reset(p);
  if (int rc = rebuild_peer_tab(p)) { (void)direct_reset(p); return rc; }
  p->direct = true;
  *enabled = 1;
  return 0;
}

int rf_comm_direct_enabled(rf_plan* p, int* enabled) {
  RF_REQUIRE(p && enabled, "null argument");
  *enabled = p->direct ? 1 : 0;
  return 0;
}

// the same between n virtual ranks living on one device (their buffers are plain device pointers to one another); enable = 0 unlinks
int rf_slab_link_direct(rf_plan** plans, int n, int enable) {
  RF_REQUIRE(plans && n >= 1, "null argument");
  for (int g = 0; g < n; ++g) {
    RF_REQUIRE(plans[g] && plans[g]->nranks == n && plans[g]->rank == g, "plans must be ranks 0..n-1 of one n-rank job");
    RF_REQUIRE(plans[g]->device == plans[0]->device && plans[g]->comm == nullptr, "virtual ranks live on one device and have no communicator");
    RF_REQUIRE(!enable || (direct_shape_ok(plans[g]) && !plans[g]->replicate), "this plan's y-pass tiles would straddle x planes (or it replicates its generation): no direct exchange");
  }
  RF_HIP(hipSetDevice(plans[0]->device));
  for (int g = 0; g < n; ++g) {
    if (int rc = direct_reset(plans[g])) return rc;
    if (enable) if (int rc = ensure_batch_buffers(plans[g])) return rc;
  }
  if (!enable) return 0;
  for (int g = 0; g < n; ++g) {
    rf_plan* p = plans[g];
    p->peer_R[0].resize(n); p->peer_R[1].resize(n);
    for (int h = 0; h < n; ++h) { p->peer_R[0][h] = plans[h]->R; p->peer_R[1][h] = plans[h]->R2; }
    if (int rc = rebuild_peer_tab(p)) return rc;
    p->direct = true;
    p->standin_wg = 0;
  }
  return 0;
}

// The IPC hand-off of rf_comm_enable_direct with the TRANSPORT LEFT TO THE CALLER: ranks of one job that live in different processes
// and have no communicator (a test's two processes on one GPU, a host framework with its own rendezvous).  export: this rank's record
// (RF_DIRECT_RECORD_BYTES; all zero when the plan cannot take part); import: the records of all ranks, in rank order -- maps the
// peers' buffers and switches the storing y pass on (*enabled = 0 and nothing mapped when any rank's record says no).  The barriers
// between the storing y pass and the z pass are the caller's too: rf_slab_forward, rf_sync, the caller's barrier, rf_slab_backward.
int rf_slab_direct_export(rf_plan* p, void* record, int nbytes) {
  RF_REQUIRE(p && record, "null argument");
  RF_REQUIRE(nbytes == (int)DIRECT_REC, "record size: RF_DIRECT_RECORD_BYTES");
  RF_REQUIRE(!p->unpacked && p->nranks > 1 && p->comm == nullptr, "rf_slab_direct_export applies to a rank of a multi-rank plan without a communicator");
  RF_HIP(hipSetDevice(p->device));
  direct_fill_record(p, (unsigned char*)record);
  return 0;
}
int rf_slab_direct_import(rf_plan* p, const void* records, int nranks, int* enabled) {
  RF_REQUIRE(p && records && enabled, "null argument");
  *enabled = 0;
  RF_REQUIRE(!p->unpacked && p->nranks > 1 && p->comm == nullptr && nranks == p->nranks, "rf_slab_direct_import: records of all ranks of this communicator-less plan");
  RF_HIP(hipSetDevice(p->device));
  drop_graphs(p);
  if (int rc = direct_reset(p)) return rc;
  if (!(direct_shape_ok(p) && !p->replicate) || ensure_batch_buffers(p) != 0) return 0;
  std::vector<void*> R0, R1;
  if (!direct_map_peers(p, (const unsigned char*)records, R0, R1)) return direct_reset(p);
  p->peer_R[0] = R0; p->peer_R[1] = R1;
  if (int rc = rebuild_peer_tab(p)) { (void)direct_reset(p); return rc; }
  p->direct = true;
  p->standin_wg = 0;
  *enabled = 1;
  return 0;
}

// ONE virtual rank through the schedule of the direct mode (the counterpart of rf_slab_set_exchange_standin): its y pass stores block h
// into segment h of its OWN receive buffers -- the store pattern and volume of the real thing, minus the links; the result is not a field.
// overlap: the batch's storing y pass on the exchange stream beside the neighbouring realisations' x / z passes (1) or everything on one stream (0)
int rf_slab_set_direct_standin(rf_plan* p, int on, int overlap) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked && !p->generic, "this call applies to packed plans on the tiled kernels");
  RF_REQUIRE(p->nranks > 1 && p->comm == nullptr, "the direct stand-in is for a rank of a multi-rank plan without a communicator");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = direct_reset(p)) return rc;
  p->direct_standin = false;
  if (!on) return 0;
  RF_REQUIRE(direct_shape_ok(p) && !p->replicate, "this plan's y-pass tiles would straddle x planes (or it replicates its generation): no direct exchange");
  if (int rc = ensure_batch_buffers(p)) return rc;
  const long long blk = (long long)p->nxl * p->ny * p->nzl * (long long)p->csize;
  p->peer_R[0].resize(p->nranks); p->peer_R[1].resize(p->nranks);
  for (int h = 0; h < p->nranks; ++h) {
    p->peer_R[0][h] = (char*)p->R + (long long)(h - p->rank) * blk;
    p->peer_R[1][h] = (char*)p->R2 + (long long)(h - p->rank) * blk;
  }
  if (int rc = rebuild_peer_tab(p)) return rc;
  p->direct = true;
  p->direct_standin = true;
  p->direct_overlap = overlap ? 1 : 0;
  p->standin_wg = 0;
  return 0;
}

int rf_slab_backward(rf_plan* p) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->nranks > 1, "rf_slab_* are for multi-rank plans");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = queue_z_slab(p, p->R, p->W, p->stats, p->stream)) return rc;
  p->stats_slot = 0;
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

// A rank of an n-rank job WITHOUT a communicator (a virtual rank): `workgroups` > 0 lets rf_realise / rf_realise_batch run the real
// schedule of a multi-GPU rank -- forward half, exchange on the exchange stream under the next forward half, gathering z pass -- with
// the all-to-all replaced by a copy kernel of that many 256-thread workgroups that reads the blocks the rank would send and writes
// the segments it would receive (RCCL's footprint in local HBM and on the compute units, without the links).  0 = off.
int rf_slab_set_exchange_standin(rf_plan* p, int workgroups) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked && !p->generic, "this call applies to packed plans on the tiled kernels");
  RF_REQUIRE(p->nranks > 1 && p->comm == nullptr, "the exchange stand-in is for a rank of a multi-rank plan without a communicator");
  RF_REQUIRE(workgroups >= 0 && workgroups <= 4096, "workgroups must be in [0, 4096]");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipStreamSynchronize(p->stream));
  p->standin_wg = workgroups;
  p->standin_read_pct = p->standin_write_pct = 100;
  return 0;
}

// the same with the two directions of the exchange's local traffic taken apart: the copy kernel reads read_percent and writes
// write_percent of every block (100 / 100 = rf_slab_set_exchange_standin)
int rf_slab_set_exchange_standin_ex(rf_plan* p, int workgroups, int read_percent, int write_percent) {
  RF_REQUIRE(read_percent >= 0 && read_percent <= 100 && write_percent >= 0 && write_percent <= 100, "percentages must be in [0, 100]");
  if (int rc = rf_slab_set_exchange_standin(p, workgroups)) return rc;
  p->standin_read_pct = read_percent;
  p->standin_write_pct = write_percent;
  return 0;
}

int rf_slab_stats(rf_plan* p, double* sum, double* sumsq) {
  RF_REQUIRE(p && sum && sumsq, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  double st[2];
  RF_HIP(hipMemcpyAsync(st, p->stats, sizeof(st), hipMemcpyDeviceToHost, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  *sum = st[0]; *sumsq = st[1];
  return 0;
}

}  // extern "C"
