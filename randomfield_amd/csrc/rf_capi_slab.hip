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
  RF_HIP(hipMemsetAsync(p->coll_scratch, 0, 2 * sizeof(double), p->stream));
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
