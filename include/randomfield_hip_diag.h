/*
 * randomfield_hip_diag.h -- diagnostics of librandomfield_hip.so: NOT for consumers.
 *
 * The entry points here exist for tests/, bench.py and tools/: per-kernel event times, knobs of the launch structure that the
 * library otherwise chooses by itself, the slab pipeline of a multi-GPU plan in separate steps ("virtual ranks": ranks 0..n-1 of
 * one job living on ONE device, the transport replaced by device copies), and read-back of internal buffers.  They may change
 * with any minor version (rf_version(), randomfield_hip.h) and rf_abi_features() & RF_FEATURE_DIAGNOSTICS says whether a build has
 * them at all.  None of them has a reference counterpart: the reference's FFT is one library call (transform.py:303-315), its
 * stream one RandomState (random.py:24-28), and it has no distributed code.
 */
#ifndef RANDOMFIELD_HIP_DIAG_H
#define RANDOMFIELD_HIP_DIAG_H

#include "randomfield_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- timing and launch structure ----------------------------------------- */
/* GPU time of each kernel of the last rf_realise, 5 floats: x pass (main kernel), y pass, z pass, reduce,
 * and the small x-pass launch that repairs the kz = 0 tiles (0 when the x pass is a single launch) */
int rf_kernel_ms(rf_plan* plan, float* ms5);
/* The z pass of slab s and the y pass of slab s + 1 in ONE launch (the next slab's tiles fill the compute units the draining pass leaves
 * idle; float32 plans whose y pass is the 1024-point one and whose rows hold 512 complex: the 1024^3 pipeline).  mode 0: never;
 * 1 (default): untimed calls -- graph-captured batches, rf_realise_batch_reference; 2: timed calls too, with an event behind every
 * launch: rf_kernel_ms then reports [1] = the first y launch + all merged launches, [2] = the last z launch, and rf_merged_yz_ms the
 * merged launches' summed duration and number.  (No reference counterpart: launch structure of transform.py:303-315's one call.) */
int rf_set_merged_yz(rf_plan* plan, int mode);
int rf_merged_yz_ms(rf_plan* plan, float* sum_ms, int* launches);
/* (when the y and z passes run slab by slab -- RF_FLAG_YZ_SLAB_PLANES -- ms5[1] and ms5[2] are the sums over their launches)
 * How the y / z passes of this plan are launched: *nslab launches each, over *planes x planes (1 and nx: whole-grid passes).
 * No reference counterpart: the reference's FFT is one library call (transform.py:303-315). */
int rf_yz_slabs(rf_plan* plan, int* nslab, int* planes);

/* ---- internal buffers ------------------------------------------------------ */
/* copy deviates [first, first+count) of the device noise buffer to the host (tests) */
int rf_download_noise(rf_plan* plan, double* host, unsigned long long first, unsigned long long count);

/* ---- virtual ranks: the multi-GPU pipeline step by step on one device ------ */
/* The slab pipeline in separate steps, for tests and custom exchanges: forward = generation + x and y
 * passes on this rank's kz slab; backward = z pass on this rank's x slab + local (sum, sumsq).
 * rf_slab_exchange_local performs the all-to-all between n "virtual ranks" that live on ONE device
 * (plain device copies, no RCCL): it checks layouts and kernels where only one GPU is available. */
int rf_slab_forward(rf_plan* plan, uint64_t seed, int mode, const double* noise_host);
/* the same forward half fed like rf_realise_potential (generate.py:200-217: the rank's planes of delta(k)/k^2 are
 * kept in its potential buffer) or like rf_execute_c2r (from the rank's k buffer, e.g. after rf_load_potential) */
enum { RF_SLAB_GENERATE = 0, RF_SLAB_GENERATE_SAVE_POTENTIAL = 1, RF_SLAB_FROM_KSPACE = 2 };
int rf_slab_forward_ex(rf_plan* plan, uint64_t seed, int mode, const double* noise_host, int source);
int rf_slab_exchange_local(rf_plan** plans, int n);
int rf_slab_backward(rf_plan* plan);
int rf_slab_stats(rf_plan* plan, double* sum, double* sumsq);
/* the multi-rank forward transform (rf_execute_r2c) in the same separate steps: rows = z pass on the x slab + cut into send
 * blocks; the reverse all-to-all between virtual ranks; cols = forward y and x passes on the kz slab + the k-space side array */
int rf_slab_r2c_rows(rf_plan* plan);
int rf_slab_exchange_local_reverse(rf_plan** plans, int n);
int rf_slab_r2c_cols(rf_plan* plan);
/* One virtual rank through the REAL schedule of a multi-GPU rank: with workgroups > 0, rf_realise and rf_realise_batch on a rank of an
 * n-rank plan that has no communicator run forward half / exchange on the exchange stream under the next forward half / gathering z
 * pass, the all-to-all replaced by a copy kernel of `workgroups` 256-thread workgroups that reads the n - 1 blocks the rank would send
 * and writes the n - 1 segments it would receive: RCCL's footprint in local HBM and on the compute units, without the links.  What it
 * measures: how much the overlapped passes lose to the exchange's local traffic (bench.py other_configs).  The
 * received segments hold the rank's own data for other x slabs, so the result is not a field.  0 = off. */
int rf_slab_set_exchange_standin(rf_plan* plan, int workgroups);
/* ... with the two directions of that traffic taken apart: the copy kernel reads read_percent and writes write_percent of every block
 * (100 / 100 = the call above; 100 / 0: the send side's reads alone; 0 / 100: the receive side's writes alone; 50 / 50: half the volume) */
int rf_slab_set_exchange_standin_ex(rf_plan* plan, int workgroups, int read_percent, int write_percent);
/* rf_comm_enable_direct (randomfield_hip.h) between n virtual ranks living on one device: every plan's y pass (rf_slab_forward) then
 * stores into the receive buffers of the others -- plain device pointers here, IPC mappings on a real job -- and rf_slab_exchange_local
 * is not called at all: forward on every rank, then backward on every rank.  enable = 0 unlinks.  Same fields, bit for bit. */
int rf_slab_link_direct(rf_plan** plans, int n, int enable);
/* The same hand-off between ranks that live in DIFFERENT processes and have no communicator, the transport left to the caller (the
 * two-process test on one GPU; a host with its own rendezvous): export writes this rank's record (two IPC handles + a flag; all zero when
 * the plan cannot take part), import takes the records of all ranks in rank order, maps the peers' buffers and switches the storing y
 * pass on (*enabled = 0 and nothing mapped when any record says no).  The barrier between the storing y pass and the z pass is the
 * caller's: rf_slab_forward, rf_sync, barrier, rf_slab_backward. */
#define RF_DIRECT_RECORD_BYTES 192
int rf_slab_direct_export(rf_plan* plan, void* record, int nbytes);
int rf_slab_direct_import(rf_plan* plan, const void* records, int nranks, int* enabled);
/* ONE virtual rank through the schedule of the direct exchange (the counterpart of rf_slab_set_exchange_standin): rf_realise /
 * rf_realise_batch run x pass, storing y pass, gathering z pass with the stores of block h going to segment h of the rank's OWN
 * receive buffers -- pattern and volume of the real stores, without the links; the result is not a field (rf_download_real refuses).
 * overlap = 1: in batches the storing y pass runs on the exchange stream beside the x / z passes of the neighbouring realisations (what a
 * real job does, where that pass is link-bound); 0: everything on one stream.  on = 0 switches the stand-in off. */
int rf_slab_set_direct_standin(rf_plan* plan, int on, int overlap);
/* rf_mt_share_exchange (randomfield_hip.h) between n virtual ranks living on one device */
int rf_mt_share_exchange_local(rf_plan** plans, int n);

#ifdef __cplusplus
}
#endif
#endif /* RANDOMFIELD_HIP_DIAG_H */
