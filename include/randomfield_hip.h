/*
 * randomfield_hip.h -- C-ABI of the MI355X (gfx950) Gaussian-random-field hot path.
 *
 * Drop-in boundary for the Fourier-space sampling path of dkirkby/randomfield
 * (SURVEY.md section 8b).  The reference has no FFI of its own -- the path is
 * plain Python over numpy/scipy/pyFFTW -- so each entry point below cites the
 * reference Python interface (randomfield/<file>:<line>) whose work it takes
 * over.  The Python host (randomfield_amd/) binds these with ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success and a non-zero status on
 * failure (message via rf_last_error(), thread-local).  No exceptions, torch
 * types or callbacks cross the boundary: plain pointers and sizes only.  A plan
 * is used from one host thread at a time.  Host arrays use the reference's
 * layouts: k-space (nx, ny, nz/2+1) complex C-order (transform.py:192), real
 * space (nx, ny, nz) dense or (nx, ny, nz+2) padded (transform.py:227-235).
 * All work is queued on the plan's HIP stream; rf_sync() waits for it.
 *
 * THIS header is the consumer surface.  Per-kernel timing, launch-structure knobs and the "virtual rank" entry points that
 * exist for tests, bench.py and the profiling tools are declared in randomfield_hip_diag.h and are not part of what a
 * consumer should bind.
 */
#ifndef RANDOMFIELD_HIP_H
#define RANDOMFIELD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rf_plan rf_plan;

enum { RF_F32 = 0, RF_F64 = 1 };                 /* complex64/float32 or complex128/float64 plan */
enum { RF_NOISE_NATIVE = 0, RF_NOISE_EXTERNAL = 1, RF_NOISE_RESIDENT = 2 };
enum { RF_LAYOUT_DENSE = 0, RF_LAYOUT_PADDED = 1 };

/* ---- library ---------------------------------------------------------- */
/* ABI version of THIS header: RF_ABI_MAJOR changes when an entry point's meaning or signature changes or one is removed, RF_ABI_MINOR on
 * every addition.  rf_version() returns the RF_ABI_VERSION the loaded library was built from: a consumer compares its major with the
 * header it was compiled against and asks rf_abi_features() which groups of entry points the build carries.  (History: rounds 1-4 of
 * this repository returned the constant 1 while the surface grew from ~20 to 70 entry points; 5.0 is the first version that means
 * something: the consumer surface below + the diagnostics of randomfield_hip_diag.h.) */
#define RF_ABI_MAJOR 5
#define RF_ABI_MINOR 3
#define RF_ABI_VERSION ((RF_ABI_MAJOR << 16) | RF_ABI_MINOR)
int rf_version(void);                            /* (major << 16) | minor */
/* bit mask of the groups of entry points this build exports (each bit: every function of the group is present and works as this
 * header says; a 0 bit: calling one returns an error or the symbol is missing) */
enum {
  RF_FEATURE_REALISE = 1 << 0,             /* rf_generate / rf_execute_c2r / rf_realise / rf_realise_batch(_prepare) / rf_moments: rows K..D */
  RF_FEATURE_R2C = 1 << 1,                 /* rf_execute_r2c, rf_upload_real: the reverse plan (transform.py:278-301) */
  RF_FEATURE_C2C = 1 << 2,                 /* rf_plan_create_c2c, rf_upload_c, rf_download_c, rf_execute_c2c: Plan(packed=False) */
  RF_FEATURE_LOGNORMAL = 1 << 3,           /* rf_lognormal, rf_scale_z, rf_affine_z, rf_set_z_tables, rf_realise_lognormal: row L */
  RF_FEATURE_POTENTIAL = 1 << 4,           /* rf_realise_potential, rf_save_potential, rf_load_potential, rf_realise_scaled_potential,
                                              rf_can_regenerate_potential: generate.py:200-217, 333-347 */
  RF_FEATURE_LENSING = 1 << 5,             /* rf_lensing_potential, rf_download_aux: generate.py:352-416 */
  RF_FEATURE_MT19937 = 1 << 6,             /* rf_mt_set_jump, rf_noise_mt19937(_ex), rf_realise_batch_reference, rf_can_batch_reference:
                                              random.py:24-28 replayed on the device */
  RF_FEATURE_MT19937_SHARED = 1 << 7,      /* rf_mt_share_*: that stream shared between kz-slab ranks */
  RF_FEATURE_MULTI_RANK = 1 << 8,          /* nranks > 1 plans, rf_comm_*: kz slabs + one RCCL all-to-all (librccl is dlopen'ed by rf_comm_*) */
  RF_FEATURE_GENERIC_SHAPES = 1 << 9,      /* every even shape (axes of up to 8192 complex64 / 4096 complex128 points, or two factors that fit): rf_shape_supported(_dtype) == 2 */
  RF_FEATURE_EXCHANGE_CHUNKS = 1 << 10,    /* RF_FLAG_EXCHANGE_CHUNKS */
  RF_FEATURE_DIRECT_EXCHANGE = 1 << 12,    /* rf_comm_enable_direct: the y pass stores into the peers' receive buffers (IPC-mapped), no all-to-all kernels */
  RF_FEATURE_DIAGNOSTICS = 1 << 11         /* the entry points of randomfield_hip_diag.h (timing per kernel, launch structure, virtual ranks) */
};
unsigned rf_abi_features(void);
const char* rf_last_error(void);
int rf_device_count(int* count);
/* is this grid shape supported by the HIP kernels?  1: tiled power-of-two kernels (nx, ny in 8..2048, nz in 16..2048);
 * 2: generic mixed-radix kernels (any other even nx, ny, nz -- the reference's own test shapes (4,6,8) and (40,60,80), transform.py:172-177;
 * single GPU, k space materialised.  An axis of up to 8192 points -- 4096 on RF_F64 plans -- is one pass; a longer one must split into two
 * factors within that cap and takes two); 0: unsupported */
int rf_shape_supported(int nx, int ny, int nz);
/* the same for ONE dtype (RF_F32 / RF_F64): what rf_plan_create(nx, ny, nz, dtype, ...) on one rank will accept -- rf_shape_supported answers
 * for complex64 plans on the generic path (one-pass axes up to 8192 points), complex128 lines hold 4096 */
int rf_shape_supported_dtype(int nx, int ny, int nz, int dtype);

/* ---- plan: replaces transform.Plan.__init__ / allocate (transform.py:10-43,170-276)
 * One device buffer of nx*ny*nz reals (== nx*ny*nz/2 complex) is the analogue of
 * the reference's single in-place buffer.  nranks/rank select the kz-slab (k space)
 * / x-slab (real space) owned by this process for multi-GPU plans (1, 0 otherwise). */
int rf_plan_create(rf_plan** plan, int nx, int ny, int nz, int dtype, int device, int nranks, int rank);
int rf_plan_destroy(rf_plan* plan);
int rf_plan_nbytes(rf_plan* plan, size_t* nbytes);          /* transform.py:221,226 nbytes_allocated */
/* plan options.  RF_FLAG_EXACT_GENERATION = 1 makes native-noise float32 realisations use the
 * reference's exact float64 rounding chain for |k| and sigma(k) instead of the fast float32 one.
 * RF_FLAG_FORCE_SLAB_PATH = 2 routes a single-rank plan through the multi-GPU slab pipeline (y pass on the
 * slab, exchange = copy of the own block, gathering z pass, pipelined batches): a test hook.
 * RF_FLAG_REPLICATED_GENERATION = 4 (multi-rank plans, native generator): no all-to-all -- every rank generates all of
 * k space on the fly, runs the full x-FFT and keeps only its own x slab; y and z passes are local.  P-fold redundant
 * x-pass arithmetic instead of the exchange: faster when few GPUs share few xGMI links (2 GPUs: one link).
 * RF_FLAG_TRANSPOSED_INTERMEDIATE = 8 (default OFF; single-GPU plans): the x pass stores its tiles as contiguous chunks into a
 * blocked scratch array of the field's size, the y pass runs in place there and the z pass gathers from it into the field
 * buffer.  Measured on MI355X: 5 % faster at 2048^3 float32, slower at 1024^3; twice the device memory.
 * RF_FLAG_YZ_SLAB_PLANES = 16 (single-GPU plans; the value is a count, not a boolean): the y and z passes run slab by slab of
 * `value` x planes, so that the z pass finds what the y pass has just written in the 256 MiB Infinity Cache; -1 (default) picks
 * slabs of about that size, 0 = whole-grid passes.
 * RF_FLAG_EXCHANGE_CHUNKS = 32 (multi-rank plans in exchange mode; the value is a count, a power of two): the rank's kz slab is
 * generated, x- and y-transformed and SENT as `value` sub-slabs, so that inside ONE realisation (one generate_delta_field call,
 * generate.py:144-230) the all-to-all of sub-slab c runs on a second stream under the forward passes of sub-slab c + 1 instead of
 * behind all of them; the gathering z pass reads ranks x value segments per row.  1 (default) = one exchange of the whole slab. */
enum { RF_FLAG_EXACT_GENERATION = 1, RF_FLAG_FORCE_SLAB_PATH = 2, RF_FLAG_REPLICATED_GENERATION = 4, RF_FLAG_TRANSPOSED_INTERMEDIATE = 8,
       RF_FLAG_YZ_SLAB_PLANES = 16, RF_FLAG_EXCHANGE_CHUNKS = 32 };
int rf_plan_set_flag(rf_plan* plan, int flag, int value);
/* run on a caller-owned HIP stream (hipStream_t passed as void*); NULL restores the plan's own stream */
int rf_plan_set_stream(rf_plan* plan, void* hip_stream);

/* ---- inputs ----------------------------------------------------------- */
/* Per-axis float64 tables k_a(i)^2 computed by the host exactly as
 * powertools.create_ksq_grids (powertools.py:27-37): lengths nx, ny, nz/2+1. */
int rf_set_kgrid(rf_plan* plan, const double* kx2, const double* ky2, const double* kz2);
/* float64 tables x_i = log10 k_i and s_i = N3*sqrt(P_i/(2 Vbox)) computed by the host
 * exactly as powertools.tabulate_sigmas (powertools.py:153-154), n >= 2 rows. */
int rf_set_power(rf_plan* plan, const double* log10k, const double* sigma, int n);

/* ---- rows K,T,R,S: fill_with_log10k + tabulate_sigmas + randomize + symmetrize
 * (powertools.py:40-61,125-164; random.py:12-29; transform.py:114-158).
 * Leaves the symmetrised k-space array in the plan's API-layout k buffer.
 * mode RF_NOISE_NATIVE: counter-based Philox4x32-7 + Box-Muller keyed by (seed, cell).
 * mode RF_NOISE_EXTERNAL: noise_host = 2*nx*ny*(nz/2+1) float64 deviates in the order of
 * RandomState(seed).normal(size=2*M) (random.py:24-28) -- the same-seed parity mode.
 * mode RF_NOISE_RESIDENT: the deviates already in the plan's device buffer (rf_noise_mt19937 or a
 * previous external-noise call). */
int rf_generate(rf_plan* plan, uint64_t seed, int mode, const double* noise_host);

/* ---- on-GPU replay of the reference's noise stream: np.random.RandomState(seed).normal(size=2*M)
 * (random.py:24-28) = MT19937 + legacy polar method, filled into the plan's device noise buffer in the
 * reference's order; afterwards rf_generate / rf_realise with mode RF_NOISE_RESIDENT use it (same-seed
 * parity without drawing 2*M deviates on the host and uploading them).
 * rf_mt_set_jump: the jump polynomials of a radix-R tree over the segments (L = 624*blocks_per_segment words each):
 * row t*(R-1) + (m-1) holds the positions of the set coefficients of t^(m * R^t * L) mod phi(t), m = 1 .. R-1, for
 * as many stages t as the largest grid needs (computed by randomfield_amd/mt19937.py); pos is npolys x stride uint16.
 * rf_noise_mt19937: state624 = the generator's initial state (init_genrand(seed)); *accepted (may be NULL)
 * receives the number of accepted polar attempts that were generated. */
int rf_mt_set_jump(rf_plan* plan, int npolys, const uint16_t* pos, const int* npos, int stride, int blocks_per_segment, int radix);
int rf_noise_mt19937(rf_plan* plan, const uint32_t* state624, unsigned long long* accepted);
/* single = 1 (a request, honoured by float32 plans that have the fast generation pass; others keep float64): keep the
 * deviates as float32 pairs instead -- half the memory traffic of the replay and of
 * the generation pass that reads them; accept / reject stays in float64, f = sqrt(-2 log r2 / r2) is formed in float32
 * (1e-7 relative); the pairs stay in the replay's per-segment runs (8 bytes per polar attempt of the whole stream) and the
 * generation pass locates them.  rf_realise / rf_realise_potential with RF_NOISE_RESIDENT use whichever copy is resident; rf_generate,
 * rf_download_noise and float64 plans need the float64 ones (single = 0, = rf_noise_mt19937). */
int rf_noise_mt19937_ex(rf_plan* plan, const uint32_t* state624, unsigned long long* accepted, int single);
/* n same-seed realisations back to back (random.py:24-28 for each seed; single-GPU complex64 plans on the fast generation path):
 * the replay of seed i + 1 runs on a second stream under the y / z passes of seed i.  states = n x 624 words (MT19937 start
 * states, e.g. numpy's init_genrand / init_by_array of each seed); rms_out (n, optional) = np.std of every field.  The field of
 * the last seed is the plan's current field; every field equals what rf_noise_mt19937_ex(single) + rf_realise(RESIDENT) gives. */
int rf_realise_batch_reference(rf_plan* plan, const uint32_t* states, int n, double* rms_out);
/* does rf_realise_batch_reference serve this plan as it stands (single-GPU complex64 plan on the fast generation path, tables and
 * jump polynomials set, segments long enough for the float32 form)?  With n = 1 it is also the fastest way to ONE same-seed field --
 * replay and passes queued back to back, no host synchronisation between them -- and what Generator(rng='reference') calls
 * (generate.py:191-199 with random.py:24-28 inside).  1 / 0. */
int rf_can_batch_reference(rf_plan* plan);
/* ---- the same stream shared between the ranks of a kz-slab job (random.py:24-28 is ONE sequential stream; the reference is
 * single-process, so there is no reference interface to mirror beyond that definition).  rf_noise_mt19937_ex on a multi-rank
 * plan replays the whole stream on every rank; these calls replay 1/P of it per rank and exchange the deviates once:
 *   rf_mt_share_segments  the stream's segment count and this rank's range [first, first + count)
 *   rf_mt_share_begin     jump to segment `first`, replay the local segments; counts_out[count] = accepted pairs per segment
 *   rf_mt_share_gather    every rank's counts into counts_all[nseg_total], in segment order (an integer all-reduce over RCCL;
 *                         virtual ranks: concatenate on the host instead)
 *   rf_mt_share_pack      scan counts_all, pack the local pairs by destination rank
 *   rf_mt_share_exchange  one all-to-all over RCCL on the plan's stream (rf_comm_init first); rf_mt_share_exchange_local
 *                         (randomfield_hip_diag.h): the same between n virtual ranks living on one device (tests)
 *   rf_mt_share_finish    the plan's resident float64 deviates are this rank's planes of the stream (then RF_NOISE_RESIDENT)
 * single = 1 (complex64 plans): pairs travel as float32 (8 B per cell, the volume of the field's own exchange), rounded as
 * rf_noise_mt19937_ex(single = 1) rounds them; single = 0: the exact float64 deviates, bit for bit those of the replicated replay. */
int rf_mt_share_segments(rf_plan* plan, int* nseg_total, int* seg_first, int* seg_count);
int rf_mt_share_begin(rf_plan* plan, const uint32_t* state624, int single, unsigned long long* counts_out);
int rf_mt_share_gather(rf_plan* plan, unsigned long long* counts_all);
int rf_mt_share_pack(rf_plan* plan, const unsigned long long* counts_all);
int rf_mt_share_exchange(rf_plan* plan);
int rf_mt_share_finish(rf_plan* plan, unsigned long long* accepted);

/* calculate_newtonian_potential (generate.py:333-343) without a stored potential: the inverse transform of scale * delta(k) / k^2
 * with delta(k) regenerated inside the generation pass exactly as rf_realise(seed, mode) produces it (native generator: keyed by
 * (seed, cell); RF_NOISE_RESIDENT: the replayed stream still on the device).  rf_can_regenerate_potential: 1 if the plan and
 * mode allow it (fast generation pass; float32 replayed deviates), else use rf_realise_potential + rf_load_potential.
 * factor_z (optional): plane z of the result times factor_z[z], the light-cone weighting G(z)/(1+z) of generate.py:344-347, applied
 * in the z pass's store (= rf_scale_z on the finished field, without its sweep). */
int rf_can_regenerate_potential(rf_plan* plan, int mode);
int rf_realise_scaled_potential(rf_plan* plan, uint64_t seed, int mode, double scale, const double* factor_z /* nz, or NULL */);

/* ---- row X: Plan.execute (transform.py:303-315) ------------------------- */
int rf_execute_c2r(rf_plan* plan);               /* k buffer -> real field, numpy normalisation 1/(nx ny nz) */
int rf_execute_r2c(rf_plan* plan);               /* real field -> k buffer, unnormalised (transform.py:278-301's reverse plan); on a multi-rank
                                                  * plan: rows on the x slab, the all-to-all in the other direction, columns on the kz slab */

/* ---- unpacked complex-to-complex plans: Plan(packed=False) (transform.py:207-213,266-270; the reference's
 * tests/test_transform.py:180-298).  One device buffer [nx][ny][nz] complex, transformed in place.
 * Power-of-two axes in [8, 2048] run on the tiled kernels, any other even axes on the generic ones (up to 8192 (RF_F32) / 4096 (RF_F64) points in one pass, longer ones split into two factors that fit).
 * Only rf_upload_c / rf_download_c / rf_execute_c2c,
 * rf_sync, rf_elapsed_ms, rf_plan_set_stream, rf_plan_nbytes, rf_device_ptr and rf_plan_destroy apply. */
int rf_plan_create_c2c(rf_plan** plan, int nx, int ny, int nz, int dtype, int device);
int rf_upload_c(rf_plan* plan, const void* host);     /* [nx][ny][nz] complex64 / complex128, C order */
int rf_download_c(rf_plan* plan, void* host);
/* direction = -1: forward, unnormalised (np.fft.fftn); +1: inverse with numpy's 1/(nx ny nz) (np.fft.ifftn) */
int rf_execute_c2c(rf_plan* plan, int direction);

/* ---- fused K,T,R,S,X,D: Generator.generate_delta_field(save_potential=False)
 * (generate.py:191-199,218-219).  Generation is fused into the first FFT pass; the
 * k-space array is never materialised.  rms/mean are available from rf_moments(). */
int rf_realise(rf_plan* plan, uint64_t seed, int mode, const double* noise_host);
/* The reference's DEFAULT call, generate_delta_field(save_potential=True) (generate.py:191-219): as rf_realise, and
 * delta(k) / k^2 (0 at DC; generate.py:200-217) is left in the plan's potential buffer for rf_load_potential (on a kz-slab
 * rank: its own planes, then the Nyquist plane).  With the native generator (float32 and float64 plans) or resident float32
 * deviates (rf_noise_mt19937_ex(single), float32 plans) the potential is a second store stream of the generation pass and
 * delta(k) is never materialised; otherwise (host deviates, exact-generation flag, generic shapes) the call runs
 * rf_generate -> rf_save_potential -> rf_execute_c2r. */
int rf_realise_potential(rf_plan* plan, uint64_t seed, int mode, const double* noise_host);
/* n realisations back to back (native noise); the field of the last seed stays resident; rms_out[i]
 * (may be NULL) = np.std of field i.  Single-GPU plans replay one captured hipGraph.  Multi-GPU plans
 * pipeline instead: realisation i+1's generation / x / y passes run on the compute stream while
 * realisation i's all-to-all is in flight on a second stream (two buffer pairs). */
int rf_realise_batch(rf_plan* plan, const uint64_t* seeds, int n, double* rms_out);
/* capture + instantiate the n-realisation graph now (otherwise the first rf_realise_batch(n) does it) */
int rf_realise_batch_prepare(rf_plan* plan, int n);

/* ---- row D: np.std(delta.flat) (generate.py:219) ------------------------ */
int rf_moments(rf_plan* plan, double* mean, double* std);

/* ---- row L: cosmotools.apply_lognormal_transform (cosmotools.py:206-221) and the
 * per-z scaling delta *= mean_matter_density (generate.py:273).
 * a_z[iz] = sqrt(log t), b_z[iz] = sqrt(t), t = 1 + (sigma*growth[iz])^2 (host, float64);
 * field <- exp(field / sigma * a_z) / b_z with the reference's four roundings. */
int rf_lognormal(rf_plan* plan, const double* a_z, const double* b_z, int nz, double sigma);
/* The same two rows fused into the realisation (single-GPU plans, power-of-two axes): generate_delta_field(save_potential=False)
 * followed by convert_delta_to_density(apply_lognormal_transform=True) (generate.py:191-199,218-219,266-273) as ONE call of five
 * sweeps.  sigma = np.std(delta) (generate.py:219) is obtained before the last pass from the y pass's output (Parseval; the mean
 * is 0 because the DC mode is), the tables are formed on the device and the map runs in the z pass's epilogue as
 * exp(delta * (a_z / sigma)) * (density_z / b_z): within a few ulp of exp's argument of the reference's four statements (1e-15
 * relative for float64 fields; rf_lognormal / rf_scale_z are the rounding-exact, unfused forms).  rf_set_z_tables: growth_z (nz) and, optionally, the mean-density factor (generate.py:273),
 * once per plan.  *sigma_out (optional; blocks) = the rms of the Gaussian field; rf_moments afterwards describes the DENSITY. */
int rf_set_z_tables(rf_plan* plan, const double* growth_z, const double* density_z_or_null, int nz);
int rf_realise_lognormal(rf_plan* plan, uint64_t seed, int mode, const double* noise_or_null, double* sigma_out);
int rf_scale_z(rf_plan* plan, const double* factor_z, int nz);
/* field <- field * mul_z[iz] + add (generate.py:271-272 non-lognormal branch) */
int rf_affine_z(rf_plan* plan, const double* mul_z, int nz, double add);

/* ---- save_potential branch (generate.py:200-217, 333-343) ---------------- */
/* potential(k) = delta(k) / k^2 (0 at DC) from the k buffer into the plan's second k buffer */
int rf_save_potential(rf_plan* plan);
/* k buffer <- scale * potential(k)  (then rf_execute_c2r gives the Newtonian potential) */
int rf_load_potential(rf_plan* plan, double scale);

/* ---- host <-> device (layout conversion to/from the reference's arrays) -- */
int rf_upload_k(rf_plan* plan, const void* host);            /* (nx, ny, nz/2+1) complex */
int rf_download_k(rf_plan* plan, void* host);
int rf_upload_real(rf_plan* plan, const void* host, int layout);
/* rows x0 <= ix < x1 of the real field into host (shape (x1-x0, ny, nz) or (.., nz+2)) */
int rf_download_real(rf_plan* plan, void* host, int layout, int x0, int x1);
/* raw device pointers (real field / k buffer) for zero-copy consumers */
int rf_device_ptr(rf_plan* plan, void** real_field, void** kspace);

/* ---- lensing potential (generate.py:352-416): psi[x][y][e] = Simpson integral over iz in [i_min, e] of
 * -2 (cot_z[iz] - cot_z[e]) * field[x][y][iz] with step `spacing` (scipy.integrate.simps(even='avg') semantics),
 * 0 for e < i_min.  Reads the real field on the device, writes an auxiliary real field (the k buffer's memory:
 * any k-space data there is lost); cot_z = cotK(D) of generate.py:383-395, nz doubles. */
int rf_lensing_potential(rf_plan* plan, const double* cot_z, int nz, double spacing, int i_min);
int rf_download_aux(rf_plan* plan, void* host, int x0, int x1);   /* planes [x0, x1) of the auxiliary field, dense */

/* generate.py:184-189,230 (the reference's calls return HOST arrays): arm a host buffer (layout and size as rf_download_real's, all nx
 * planes; ordinary pageable memory) and the NEXT realisation of a single-GPU plan on the tiled kernels delivers its field there slab by
 * slab -- the device -> host copy of each slab of x planes runs as soon as that slab's z pass has finished, while the GPU works on
 * the following slabs -- and returns when the field is complete on the host.  One shot; rf_host_sink_delivered reports whether the
 * armed call delivered (else use rf_download_real) and disarms.  host = NULL disarms. */
int rf_set_host_sink(rf_plan* plan, void* host, int layout);
int rf_host_sink_delivered(rf_plan* plan, int* delivered);

/* ---- stream control and timing ------------------------------------------ */
int rf_sync(rf_plan* plan);
/* GPU time (hipEvents on the plan's stream) of the last rf_realise / rf_realise_batch /
 * rf_execute_* call, in milliseconds; blocks until that call has finished. */
int rf_elapsed_ms(rf_plan* plan, float* ms);

/* ---- multi-GPU: one process per GPU, RCCL all-to-all between the y and z passes.
 * The 128-byte unique id comes from rank 0 and is distributed by the host
 * (torch.distributed / a file / MPI ...).  No-op requirement for nranks == 1. */
int rf_comm_unique_id(void* id128);
int rf_comm_init(rf_plan* plan, const void* id128);
/* COLLECTIVE (every rank of the communicator calls it with the same `enable`).  The exchange WITHOUT send / receive kernels: the y pass
 * of every rank stores its output tiles straight into the receive buffer of the rank that owns their x planes -- the buffers are
 * mapped into the peers' address spaces once, here (hipIpcGetMemHandle / one ncclAllGather / hipIpcOpenMemHandle) -- and one tiny
 * all-reduce per realisation is the barrier between the peers' stores and the z pass.  Local HBM traffic is then what the passes
 * alone move (the grouped ncclSend / ncclRecv exchange reads every block and writes every segment once more: DESIGN.md section 5).
 * The mapping is proved with markers stored from a kernel before it is used.  *enabled = 1: all ranks switched; 0: some rank could
 * not (no IPC between these devices, a grid whose y-pass tiles straddle x planes ...) and ALL ranks keep the RCCL exchange -- that is
 * not an error (return value 0).  Same fields either way, bit for bit.  enable = 0 switches back.  (No reference counterpart: the
 * reference has no distributed code) */
int rf_comm_enable_direct(rf_plan* plan, int enable, int* enabled);
int rf_comm_direct_enabled(rf_plan* plan, int* enabled);
/* ranks of the plan's communicator as RCCL counts them (ncclCommCount); 0 before rf_comm_init.  What a benchmark line states
 * as "did RCCL see N ranks" (the reference has no distributed code: SURVEY.md section 8e). */
int rf_comm_size(rf_plan* plan, int* nranks);
/* host-side all-reduce of 1 or 2 doubles over the plan's communicator (op 0 = sum, 1 = max), after all
 * queued work of the plan: doubles as a barrier.  With one rank it only synchronises the stream. */
int rf_comm_allreduce_f64(rf_plan* plan, double* inout, int n, int op);

#ifdef __cplusplus
}
#endif
#endif /* RANDOMFIELD_HIP_H */
