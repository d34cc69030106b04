// rf_capi.hip -- the C-ABI of include/randomfield_hip.h: plan object, device buffers, stream / graph orchestration of the HIP kernels
// (plans, inputs, realisations, transforms, host <-> device, timing; the replay of numpy's stream: rf_capi_mt.hip; the communicator
// and the slab pipeline step by step: rf_capi_slab.hip; shared helpers: rf_plan.h).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "rf_plan.h"

using namespace rf;
using namespace rfc;

namespace rfc {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return 0;
  // If an RCCL is already in the process (e.g. the one bundled with PyTorch, which comes with its own HIP /
  // HSA runtime) it must be THAT one: a second librccl would talk to a second, uninitialised HSA runtime.
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }
  if (!h)
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) return fail(4, std::string("cannot load librccl: ") + dlerror());
#define RF_SYM(field, name)                                                        \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));         \
  if (!g_rccl.field) return fail(4, std::string("librccl lacks ") + name);
  RF_SYM(GetUniqueId, "ncclGetUniqueId")
  RF_SYM(CommInitRank, "ncclCommInitRank")
  RF_SYM(CommDestroy, "ncclCommDestroy")
  RF_SYM(CommCount, "ncclCommCount")
  RF_SYM(GroupStart, "ncclGroupStart")
  RF_SYM(GroupEnd, "ncclGroupEnd")
  RF_SYM(Send, "ncclSend")
  RF_SYM(Recv, "ncclRecv")
  RF_SYM(AllReduce, "ncclAllReduce")
  RF_SYM(AllGather, "ncclAllGather")
  RF_SYM(GetErrorString, "ncclGetErrorString")
#undef RF_SYM
  g_rccl.lib = h;
  return 0;
}


void drop_graphs(rf_plan* p) {
  for (auto& kv : p->graphs) {
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
  }
  p->graphs.clear();
}

int ensure_k(rf_plan* p) {
  if (!p->K) RF_HIP(hipMalloc(&p->K, p->k_bytes));
  return 0;
}

// The saved potential is written by a second store stream of the generation pass, row for row next to the field's.  The time of
// that pass is bimodal -- 5.2 or 5.75 ms for the whole default call at 1024^3 -- and tools/pot_offset.py shows what decides it:
// NOT the virtual addresses (45 plans at identical virtual addresses of both arrays and 15 offsets of the potential inside its
// allocation, 256 B ... 4 MiB: either mode at every offset, the same offset in both modes) but the physical pages the driver
// happens to back them with, which user space neither sees nor chooses: the array sits at offset 0 of its allocation.
int ensure_p(rf_plan* p) {
  if (p->P) return 0;
  RF_HIP(hipMalloc(&p->P_base, p->p_bytes));
  p->P = p->P_base;
  return 0;
}

int ensure_g(rf_plan* p) {
  if (!p->G) RF_HIP(hipMalloc(&p->G, p->k_bytes));
  return 0;
}

// May the plan hand the x pass's output to the y and z passes through the blocked intermediate X (rf_fft.h xblock_*_geom)?
// Single-rank tiled plans whose x and y passes use tiles of the same width, whose kz runs are whole tiles and whose z pass
// can gather (whole workgroups per x block and iy, whole segments per thread group).
int xpose_row_block(const rf_plan* p) { return col_gen_row_block(p->f64, p->nx, 64); }      // x rows per block of the transposed intermediate
bool xpose_ok(const rf_plan* p) {
  if (!p->xposed || p->generic || p->unpacked || p->nranks > 1 || p->force_slab) return false;
  const int tcx = col_gen_tile_cols(p->f64, p->nx), tcy = col_tile_cols(p->f64, p->ny);
  return tcx > 0 && tcx == tcy && p->nzl >= tcx && p->nzl % tcx == 0 && row_c2r_xgather_ok(p->f64, (int)p->nzc, tcx, xpose_row_block(p));
}
int ensure_x(rf_plan* p) {
  if (!p->X && xpose_ok(p)) RF_HIP(hipMalloc(&p->X, p->w_bytes));
  return 0;
}

// Any even shape (transform.py:172-177 asks for even axes, nothing more) whose axes either fit one line of the LDS
// (generic_max_axis(dtype): 8192 for complex64, 4096 for complex128) or split into two factors that do (the four-step form of
// rf_generic.h: up to cap^2).  packed: the contiguous axis is transformed at length nz / 2 (c2r / r2c plans), else at nz (c2c plans).
bool generic_dims(int nx, int ny, int nz, int f64, bool packed, GenericDims& d) {
  if (nx < 2 || ny < 2 || nz < 2 || (nx & 1) || (ny & 1) || (nz & 1)) return false;
  const int cap = generic_max_axis(f64);
  d = GenericDims();
  d.nx = nx; d.ny = ny; d.nz = nz; d.csize = f64 ? 16 : 8;
  auto one = [&](long long n, GenericAxis& ax, GenericLong& lg, bool strided) {
    lg = GenericLong();
    if (n > cap) return generic_split(n, cap, lg);
    if (!generic_factor((int)n, ax)) return false;
    // a strided line that fits, but only one or two to a workgroup: the four-step form if the length splits (rf_generic.h)
    if (strided && generic_prefers_split(ax, (int)d.csize) && !generic_split(n, cap, lg)) lg = GenericLong();
    return true;
  };
  // (a packed plan's root table of the contiguous axis has nz entries; nz itself must stay addressable: nz / 2 <= cap^2 is the limit that bites)
  return one(nx, d.ax, d.lx, true) && one(ny, d.ay, d.ly, true) && one(packed ? nz / 2 : nz, d.az, d.lz, false);
}
bool generic_shape(int nx, int ny, int nz, int f64, GenericAxis& ax, GenericAxis& ay, GenericAxis& az_half) {
  GenericDims d;
  if (!generic_dims(nx, ny, nz, f64, true, d)) return false;
  ax = d.ax; ay = d.ay; az_half = d.az;
  return true;
}

GenParams make_gen(rf_plan* p, uint64_t seed, int mode, bool seed_from_dev) {
  GenParams g;
  g.nx = p->nx; g.ny = p->ny; g.nz = p->nz;
  g.kx2 = p->kx2; g.ky2 = p->ky2; g.kz2 = p->kz2;
  g.xt = p->xt; g.st = p->st; g.sl = p->sl; g.bin = p->bin;
  g.nt = p->nt; g.nbins = p->nbins; g.x0 = p->x0; g.inv_dx = p->inv_dx;
  g.noise_mode = (mode == RF_NOISE_RESIDENT) ? (int)RF_NOISE_EXTERNAL : mode; g.seed = seed; g.seed_dev = nullptr; (void)seed_from_dev;   // graph batches point seed_dev at seeds_dev[i]
  g.noise = p->noise;
  g.zpitch = p->nzl + 1; g.zoff = p->kz0;      // side arrays (noise, K, P) hold this rank's planes + the Nyquist plane
  return g;
}

int ensure_noise(rf_plan* p) {
  const size_t n = 2 * (size_t)p->nx * p->ny * (p->nzl + 1);       // this rank's planes + the Nyquist plane
  if (p->noise_cap < n) {
    if (p->noise) RF_HIP(hipFree(p->noise));
    p->noise = nullptr; p->noise_cap = 0; p->noise_resident = false;
    RF_HIP(hipMalloc((void**)&p->noise, n * sizeof(double)));
    p->noise_cap = n;
  }
  return 0;
}

int upload_noise(rf_plan* p, int mode, const double* noise_host) {
  if (mode == RF_NOISE_RESIDENT) {
    RF_REQUIRE(p->noise_resident || p->noise32_resident, "no deviates resident on the device: call rf_noise_mt19937 (or an external-noise run) first");
    return 0;
  }
  if (mode != RF_NOISE_EXTERNAL) return 0;
  RF_REQUIRE(noise_host != nullptr, "external noise mode needs a host noise array");
  if (int rc = ensure_noise(p)) return rc;
  // the host array is the reference's full set, 2 doubles per cell of [nx][ny][nz/2+1]; a kz-slab rank keeps its own
  // planes and the Nyquist plane (one rank: everything, one contiguous copy)
  const size_t cell = 2 * sizeof(double), hp = (size_t)(p->nzc + 1) * cell, dp = (size_t)(p->nzl + 1) * cell, rows = (size_t)p->nx * p->ny;
  if (p->nranks == 1) {
    RF_HIP(hipMemcpyAsync(p->noise, noise_host, rows * hp, hipMemcpyHostToDevice, p->stream));
  } else {
    RF_HIP(hipMemcpy2DAsync(p->noise, dp, (const char*)noise_host + (size_t)p->kz0 * cell, hp, (size_t)p->nzl * cell, rows, hipMemcpyHostToDevice, p->stream));
    RF_HIP(hipMemcpy2DAsync((char*)p->noise + (size_t)p->nzl * cell, dp, (const char*)noise_host + (size_t)p->nzc * cell, hp, cell, rows, hipMemcpyHostToDevice, p->stream));
  }
  p->noise_resident = true;
  p->noise32_resident = false;
  return 0;
}

FastGenParams make_fast(rf_plan* p, uint64_t seed, bool seed_from_dev, const uint64_t* seed_ptr) {
  FastGenParams f;
  f.nx = p->nx; f.ny = p->ny; f.nz = p->nz;
  f.dkx = p->fdkx; f.dky = p->fdky; f.dkz = p->fdkz;
  f.rec = p->frec; f.nbins = p->fnbins; f.u_scale = p->fu_scale; f.u_off = p->fu_off;
  f.seed = seed; f.seed_dev = seed_from_dev ? seed_ptr : nullptr;
  f.noise = nullptr; f.noise32 = nullptr;
  f.rowtab = nullptr; f.seg_cap = 0;
  f.zpitch = p->nzl + 1; f.zoff = p->kz0; f.ppitch = p->ppitch;
  f.pscale = p->emit_pscale; f.emit_potential = p->emit_potential ? 1 : 0;
  return f;
}

// (re)build the fast-path records once both the k grid and the power table are known
int build_fast(rf_plan* p) {
  // captured batch graphs carry the generation parameters (table pointers, bin scalars) by value: they are stale now
  RF_HIP(hipStreamSynchronize(p->stream));
  drop_graphs(p);
  p->have_fast = false;
  if (!p->have_kgrid || !p->have_power || p->generic) return 0;
  if (!col_fastgen_supported(p->f64, p->nx)) return 0;   // e.g. float64, nx = 2048: the exact kernel is used
  double kmax2 = 0, kmin2 = 1e300;
  auto scan = [&](const std::vector<double>& a) { for (double v : a) if (v > 0 && v < kmin2) kmin2 = v; };
  scan(p->h_kx2); scan(p->h_ky2); scan(p->h_kz2);
  auto mx = [](const std::vector<double>& a) { double m = 0; for (double v : a) m = v > m ? v : m; return m; };
  kmax2 = mx(p->h_kx2) + mx(p->h_ky2) + mx(p->h_kz2);
  if (!(kmin2 < 1e300) || !(kmax2 > 0)) return 0;
  std::vector<FastRec> rec;
  double x0, dx;
  if (!build_fast_records(p->h_tab, 0.5 * std::log10(kmin2) - 0.01, 0.5 * std::log10(kmax2) + 0.01, rec, x0, dx))
    return 0;   // knots too dense for the per-bin records: the exact kernel is used instead
  if ((int)rec.size() > FAST_LDS_BINS) return 0;   // the records must fit the kernel's LDS table
  // the fast kernel forms |k|^2 arithmetically, k_axis(i) = dk_axis * signed index: needs uniform fftfreq-style axes
  // (powertools.py:27-37 always makes them so); anything else keeps the exact kernel, which reads the tables
  auto regular = [](const std::vector<double>& a, int n, bool half, float& dk) {
    if (n < 2 || (int)a.size() < 2 || !(a[1] > 0)) return false;
    for (int i = 0; i < (int)a.size(); ++i) {
      const double j = (half || i < n / 2) ? i : i - n;
      const double want = j * j * a[1];
      if (std::fabs(a[i] - want) > 1e-9 * (want + 1e-300)) return false;
    }
    dk = (float)std::sqrt(a[1]);
    return true;
  };
  if (!regular(p->h_kx2, p->nx, false, p->fdkx) || !regular(p->h_ky2, p->ny, false, p->fdky) ||
      !regular(p->h_kz2, p->nz, true, p->fdkz))
    return 0;
  p->fu_scale = (float)(0.5 * std::log10(2.0) / dx);
  p->fu_off = (float)(-x0 / dx);
  if (p->frec) RF_HIP(hipFree(p->frec));
  p->frec = nullptr;
  RF_HIP(hipMalloc((void**)&p->frec, rec.size() * sizeof(FastRec)));
  RF_HIP(hipMemcpy(p->frec, rec.data(), rec.size() * sizeof(FastRec), hipMemcpyHostToDevice));
  p->fnbins = (int)rec.size();
  // side buffer of the repaired kz = 0 slots (one complex per mode (ix, iy): 8 MB at 1024^2 float32), filled and read inside the
  // generation pass by the rank that owns kz = 0 -- every rank in replicated-generation mode
  if (!p->fixbuf) RF_HIP(hipMalloc(&p->fixbuf, (size_t)p->nx * p->ny * p->csize));
  p->have_fast = true;
  return 0;
}

// x pass (generation or API k-space fused into its load) into buffer W on stream sx
// Replicated-generation mode of a multi-rank plan (RF_FLAG_REPLICATED_GENERATION): the native generator is keyed by
// the global cell index and the x pass reads nothing, so a rank can generate ALL of k space on the fly, run the
// full x-FFT and store only its own x slab [rank*nxl, (rank+1)*nxl); the y and z passes are then local and no
// all-to-all is needed (only the 2-double all-reduce of the moments).  It trades P-fold redundant x-pass arithmetic
// for the exchange: a win when the exchange is slower than (P-1) x passes -- 2 GPUs share ONE xGMI link.
// sub-slabs of the exchange (1: the whole kz slab at once); only plans that exchange have them
int slab_chunks(const rf_plan* p) {
  return (p->xchunks > 1 && (p->nranks > 1 || p->force_slab) && !p->replicate && !p->generic && !p->unpacked) ? p->xchunks : 1;
}
size_t chunk_bytes(const rf_plan* p) { return p->w_bytes / (size_t)slab_chunks(p); }

// (kz0c, nzlc >= 0: a sub-slab of this rank's planes instead of all of them -- W then points at the sub-slab's own region)
int queue_x(rf_plan* p, const GenParams& gp, const void* kspace, void* W, hipStream_t sx, bool timed, int kz0c, int nzlc) {
  // resident deviates (the numpy stream replayed by rf_noise_mt19937) take the fast float32 sigma path too; host-supplied
  // deviates (RF_NOISE_EXTERNAL, the parity mode) keep the exact reference dtype chain
  // (float32 copies of the deviates exist for this path only, and it can store the potential too; float64 ones cannot)
  const bool fast_noise = !kspace && gp.noise_mode == NOISE_EXTERNAL && p->resident_fast && p->have_fast && !p->exact_gen &&
                          !p->f64 && !(p->replicate && p->nranks > 1) && (p->noise32_resident || !p->pot_target);
  RF_REQUIRE(kspace || gp.noise_mode != NOISE_EXTERNAL || fast_noise || p->noise_resident,
             "only float32 copies of the deviates are resident: this path (exact chain / float64 plan) needs rf_noise_mt19937's float64 ones");
  const bool fast = (!kspace && gp.noise_mode == NOISE_PHILOX && p->have_fast && !p->exact_gen) || fast_noise;
  const bool rep = p->replicate && p->nranks > 1;
  RF_REQUIRE(!rep || fast, "replicated generation needs the native generator (fast path)");
  const long long nzl = nzlc >= 0 ? nzlc : (rep ? p->nzc : p->nzl);     // kz planes generated by this call (nz/2 on one GPU)
  const int kz0 = kz0c >= 0 ? kz0c : (rep ? 0 : p->kz0);
  // W == p->X: the blocked intermediate [x block][kz tile][ny][rb][TC] -- a tile (all nx rows of TC adjacent kz of one iy) is
  // nx / rb contiguous chunks of rb * TC cells there
  const ColGeom gx = (W == p->X && W != nullptr)
                         ? xblock_x_geom(p->nx, p->ny, nzl, col_gen_tile_cols(p->f64, p->nx), xpose_row_block(p))
                         : ColGeom{(long long)p->ny * nzl, 0, (long long)p->ny * nzl};
  if (timed) { RF_HIP(hipEventRecord(p->ev[5], sx)); p->repair_timed = fast; }   // overwritten by the launcher if it splits
  FastGenParams fgp = make_fast(p, gp.seed, gp.seed_dev != nullptr, gp.seed_dev);
  if (fast_noise && p->noise32_resident) {
    fgp.noise32 = reinterpret_cast<const cplx<float>*>(p->mt_scratch);      // (a later float64 replay reuses the scratch: it clears noise32_resident)
    fgp.rowtab = reinterpret_cast<const RowLoc*>(p->mt_rowtab); fgp.seg_cap = p->seg_cap;
  }
  else if (fast_noise) fgp.noise = gp.noise;
  if (fast)
    RF_HIP(launch_col_fastgen(p->f64, p->nx, W, gx, (long long)p->ny * nzl, fgp,
                              kz0, (int)nzl, p->tw_x, sx, false, timed ? p->ev[5] : nullptr,
                              rep ? p->rank * p->nxl : 0, rep ? (p->rank + 1) * p->nxl : 1 << 30, p->pot_target, p->fixbuf));
  else
    RF_HIP(launch_col_gen(p->f64, p->nx, W, gx, (long long)p->ny * nzl, gp, kspace, kz0, (int)nzl, p->tw_x, sx));
  return 0;
}

// Does this plan exchange by storing its y pass's output straight into the peers' receive buffers (rf_plan.h `direct`)?
bool direct_active(const rf_plan* p) {
  return p->direct && (p->nranks > 1 || p->force_slab) && !p->replicate && !p->generic && !p->unpacked;
}

// The device table of destination bases the storing y pass reads (rf_fft.h DirectColIO): for receive buffer b, sub-slab c and
// destination rank h, cell `off` of block h of the local sub-slab lands at  R_h + (rank * C + c) * blk_c + off  =  tab + h * blk_c + off.
int rebuild_peer_tab(rf_plan* p) {
  const int C = slab_chunks(p), P = p->nranks;
  RF_REQUIRE((int)p->peer_R[0].size() == P && (int)p->peer_R[1].size() == P, "direct exchange: the peers' receive buffers are not known");
  RF_REQUIRE(col_direct_supported(p->f64, p->ny, p->nzl / C),
             "direct exchange: a y-pass tile would straddle two x planes (nz / (2 ranks chunks) is narrower than the tile)");
  const long long blk_c = (long long)p->nxl * p->ny * (p->nzl / C) * (long long)p->csize;
  std::vector<void*> h_tab((size_t)2 * C * P);
  for (int b = 0; b < 2; ++b)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < P; ++h)
        h_tab[((size_t)b * C + c) * P + h] = p->peer_R[b][h] ? (char*)p->peer_R[b][h] + ((long long)p->rank * C + c - h) * blk_c : nullptr;
  RF_HIP(hipStreamSynchronize(p->stream));
  if (p->comm_stream) RF_HIP(hipStreamSynchronize(p->comm_stream));
  if (p->peer_tab) RF_HIP(hipFree(p->peer_tab));
  p->peer_tab = nullptr;
  RF_HIP(hipMalloc((void**)&p->peer_tab, h_tab.size() * sizeof(void*)));
  RF_HIP(hipMemcpy(p->peer_tab, h_tab.data(), h_tab.size() * sizeof(void*), hipMemcpyHostToDevice));
  p->peer_tab_chunks = C;
  return 0;
}

// the barrier between the peers' stores into a receive buffer and its readers (and, the other way, between the readers of a receive
// buffer and the next stores into it): a 2-double all-reduce on its own scratch words.  Virtual ranks (no communicator) are ordered
// by the host that drives them.
int direct_barrier(rf_plan* p, hipStream_t s) {
  if (p->nranks > 1 && p->comm) RF_NCCL(g_rccl.AllReduce(p->coll_scratch + 2, p->coll_scratch + 2, 2, ncclFloat64, ncclSum, p->comm, s));
  return 0;
}

// the y pass of sub-slab c (the whole kz slab when the plan does not chunk) out of W into the peers' receive buffers number `rbuf`
int queue_y_direct(rf_plan* p, const void* W, int rbuf, hipStream_t s, int c) {
  const int C = slab_chunks(p);
  RF_REQUIRE(p->peer_tab && p->peer_tab_chunks == C && c >= 0 && c < C, "direct exchange: the destination table does not match the plan's exchange chunks");
  RF_REQUIRE(p->peer_R[rbuf][p->rank] != nullptr, "direct exchange: the second receive buffer does not exist");
  const long long nzc_ = p->nzl / C;
  const ColGeom gy{nzc_, (long long)p->ny * nzc_, nzc_};
  int shift = 0;
  while ((1 << shift) < p->nxl) ++shift;
  RF_HIP(launch_col_direct(p->f64, p->ny, (const char*)W + (size_t)c * chunk_bytes(p), gy, p->peer_tab + ((size_t)rbuf * C + c) * p->nranks, shift,
                           (long long)p->nx * nzc_, p->tw_y, s));
  return 0;
}

// The forward half of a plan in direct mode: sub-slab by sub-slab the x pass on stream A and the storing y pass on stream Y (Y == A: one
// stream; Y != A: the stores of sub-slab c -- link-bound on a real job -- run beside the x pass of sub-slab c + 1).  The caller has
// put the barrier that frees the receive buffers in front of it on Y.  `timed` (Y == A only): ev[5] / ev[1] / ev[2] as queue_xy.
int direct_forward(rf_plan* p, const GenParams& gp, const void* kspace, void* W, int rbuf, hipStream_t A, hipStream_t Y, bool timed) {
  const int C = slab_chunks(p);
  const long long nzc_ = p->nzl / C;
  if (Y != A)
    while ((int)p->chunk_ev.size() < C + 1) { hipEvent_t e; RF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->chunk_ev.push_back(e); }
  for (int c = 0; c < C; ++c) {
    if (C == 1) { if (int rc = queue_x(p, gp, kspace, W, A, timed)) return rc; }
    else if (int rc = queue_x(p, gp, kspace, (char*)W + (size_t)c * chunk_bytes(p), A, false, p->kz0 + c * (int)nzc_, (int)nzc_)) return rc;
    if (timed && C == 1) RF_HIP(hipEventRecord(p->ev[1], A));
    if (Y != A) { RF_HIP(hipEventRecord(p->chunk_ev[c], A)); RF_HIP(hipStreamWaitEvent(Y, p->chunk_ev[c], 0)); }
    if (int rc = queue_y_direct(p, W, rbuf, Y, c)) return rc;
  }
  if (timed && C > 1) { RF_HIP(hipEventRecord(p->ev[5], A)); p->repair_timed = false; RF_HIP(hipEventRecord(p->ev[1], A)); }
  if (timed) RF_HIP(hipEventRecord(p->ev[2], A));          // (Y != A: the stores are still running on Y -- the x passes' end stands for both, as in the chunked rccl path)
  return 0;
}

// x pass (generation or API k-space fused into its load) + y pass of buffer W on stream s.
// Records ev[1] (after x) and ev[2] (after y) when `timed`.
// the forward half of ONE sub-slab c of a plan that exchanges in chunks: x pass + y pass on its nzl / xchunks planes (region c of W)
int queue_xy_chunk(rf_plan* p, const GenParams& gp, const void* kspace, void* W, hipStream_t s, int c) {
  const int C = slab_chunks(p);
  const long long nzc_ = p->nzl / C;
  char* Wc = (char*)W + (size_t)c * chunk_bytes(p);
  if (int rc = queue_x(p, gp, kspace, Wc, s, false, p->kz0 + c * (int)nzc_, (int)nzc_)) return rc;
  const ColGeom gy{nzc_, (long long)p->ny * nzc_, nzc_};
  RF_HIP(launch_col_plain(p->f64, p->ny, +1, Wc, gy, (long long)p->nx * nzc_, p->tw_y, s));
  return 0;
}

// (a plan in direct mode: the y pass stores into the peers' receive buffers `rbuf`, and what follows is the z pass, not an exchange)
int queue_xy(rf_plan* p, const GenParams& gp, const void* kspace, void* W, hipStream_t s, bool timed, int rbuf) {
  if (direct_active(p)) return direct_forward(p, gp, kspace, W, rbuf, s, s, timed);
  if (slab_chunks(p) > 1) {
    for (int c = 0; c < slab_chunks(p); ++c)
      if (int rc = queue_xy_chunk(p, gp, kspace, W, s, c)) return rc;
    if (timed) { RF_HIP(hipEventRecord(p->ev[5], s)); p->repair_timed = false; RF_HIP(hipEventRecord(p->ev[1], s)); RF_HIP(hipEventRecord(p->ev[2], s)); }
    return 0;
  }
  const bool rep = p->replicate && p->nranks > 1;
  const long long nzl = rep ? p->nzc : p->nzl, nxp = rep ? p->nxl : p->nx;      // the local array is [nxp][ny][nzl]
  const ColGeom gy{nzl, (long long)p->ny * nzl, nzl};
  if (int rc = queue_x(p, gp, kspace, W, s, timed)) return rc;
  if (timed) RF_HIP(hipEventRecord(p->ev[1], s));
  RF_HIP(launch_col_plain(p->f64, p->ny, +1, W, gy, nxp * nzl, p->tw_y, s));
  if (timed) RF_HIP(hipEventRecord(p->ev[2], s));
  return 0;
}

// multi-rank: z pass on the local x slab, rows gathered from the P received blocks in R; output
// (dense real [nxl][ny][nz]) into W, then the local (sum, sumsq)
int queue_z_slab(rf_plan* p, const void* R, void* W, double* stats_out, hipStream_t s) {
  const long long nrows = (long long)p->nxl * p->ny;
  const double scale = 1.0 / ((double)p->nx * (double)p->ny * (double)p->nz);
  const long long nzseg = p->nzl / slab_chunks(p);            // planes per received segment: nranks * chunks of them make a row
  RF_HIP(launch_row_c2r_gather(p->f64, p->nzc, R, W, nrows, scale, (int)nzseg, nrows * nzseg, p->tw_z, p->partials, s));
  RF_HIP(launch_reduce_partials(p->partials, p->npartials, stats_out, p->partials + 2 * p->npartials, s));
  p->cur = W;
  p->real_valid = true;
  return 0;
}

// multi-rank: the single all-to-all between the y and z passes.  Rank g sends to rank h the block
// [x in slab h][all y][kz in slab g], which is contiguous in W because x is the slowest axis.
// (chunk >= 0: sub-slab `chunk` only -- RF_FLAG_EXCHANGE_CHUNKS; -1: everything, all chunks of a chunked plan in ONE group;
// plain = true: the unchunked block layout whatever the flag says, as the forward transform's reverse exchange uses it)
int queue_exchange_rccl(rf_plan* p, const void* W, void* R, hipStream_t s, int chunk = -1, bool plain = false) {
  RF_REQUIRE(p->nranks == 1 || p->comm != nullptr || p->standin_wg > 0, "rf_comm_init has not been called on this multi-rank plan");
  const int C = plain ? 1 : slab_chunks(p);
  const size_t blk = (size_t)p->nxl * p->ny * p->nzl * p->csize / (size_t)C, cb = p->w_bytes / (size_t)C;
  const int c0 = chunk >= 0 ? chunk : 0, c1 = chunk >= 0 ? chunk + 1 : C;
  // block h of sub-slab c of W -> segment (this rank, c) of rank h's R
  auto src = [&](int c, int h) { return (const char*)W + (size_t)c * cb + (size_t)h * blk; };
  auto dst = [&](int c, int g) { return (char*)R + ((size_t)g * C + c) * blk; };
  for (int c = c0; c < c1; ++c)
    RF_HIP(hipMemcpyAsync(dst(c, p->rank), src(c, p->rank), blk, hipMemcpyDeviceToDevice, s));
  if (p->nranks == 1) return 0;        // forced slab path of a single-rank plan: the own block is everything
  if (!p->comm) {
    // exchange stand-in of a virtual rank (diagnostics): what this rank's RCCL kernels would do to ITS memory and compute units --
    // read the nranks - 1 blocks it sends, write the nranks - 1 segments it receives -- by a copy kernel of fixed width; the
    // segments hold this rank's own data for other x slabs (not a field: the timing of the overlapped passes is the point)
    RF_REQUIRE(p->nranks <= 17, "the exchange stand-in serves up to 17 ranks");
    for (int c = c0; c < c1; ++c) {
      const void* sp[16]; void* dp[16];
      int nb = 0;
      for (int h = 0; h < p->nranks; ++h) if (h != p->rank) { sp[nb] = src(c, h); dp[nb] = dst(c, h); ++nb; }
      RF_HIP(launch_exchange_standin(sp, dp, nb, blk, p->standin_wg, p->standin_read_pct, p->standin_write_pct, (unsigned*)p->coll_scratch, s));
    }
    return 0;
  }
  // (a failing send / receive must not leave the group open on this communicator: the first error is kept, the group is always closed)
  RF_NCCL(g_rccl.GroupStart());
  ncclResult_t first = ncclSuccess;
  const char* what = "";
  for (int c = c0; c < c1 && first == ncclSuccess; ++c)
    for (int h = 0; h < p->nranks && first == ncclSuccess; ++h) {
      if (h == p->rank) continue;
      if ((first = g_rccl.Send(src(c, h), blk, ncclUint8, h, p->comm, s)) != ncclSuccess) { what = "ncclSend"; break; }
      if ((first = g_rccl.Recv(dst(c, h), blk, ncclUint8, h, p->comm, s)) != ncclSuccess) { what = "ncclRecv"; break; }
    }
  const ncclResult_t end = g_rccl.GroupEnd();
  if (first != ncclSuccess) return fail(5, std::string(what) + " inside the grouped exchange failed: " + g_rccl.GetErrorString(first));
  RF_NCCL(end);
  return 0;
}

int ensure_comm_stream(rf_plan* p) {
  if (!p->comm_stream) RF_HIP(hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking));
  return 0;
}

// second (send, receive) buffer pair, exchange stream and events of the pipelined batches
int ensure_batch_buffers(rf_plan* p) {
  if (!p->W2) {
    RF_HIP(hipMalloc(&p->W2, p->w_bytes));
    RF_HIP(hipMalloc(&p->R2, p->w_bytes));
    if (int rc = ensure_comm_stream(p)) return rc;
    for (auto& e : p->pev) RF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  return 0;
}

// Pipelined batch of a plan in direct mode.  The y pass of realisation i stores into the PEERS' receive buffers i % 2 -- it is the
// exchange, and on a real job it runs at the speed of the links -- so it goes to the exchange stream Y, where it runs beside the z pass of
// realisation i - 1 and the x pass of realisation i + 1 on the compute stream A; B(i), a tiny all-reduce on Y, tells every rank that
// all y(i) and all z(i - 1) are done:
//   A:  x(0) | x(1)          z(0) | x(2)          z(1) | ...
//   Y:  B(-1) y(0) B(0)    | y(1)   B(1)        | y(2)   B(2)        | ...
// y(i) waits for x(i) (event) and follows B(i - 1) on Y: no peer still reads its receive buffer i % 2 (z(i - 2) is inside B(i - 1));
// z(i) waits for B(i): every peer's stores of realisation i have landed; x(i + 2), which overwrites the send buffer y(i) read, follows
// z(i) on A.  The local HBM traffic is that of the passes alone: no send-side reads, no receive-side writes by copy kernels.
// (direct_overlap = 0: everything on A in the order x(i) y(i) z(i - 1) B(i).)
int slab_batch_direct(rf_plan* p, const uint64_t* seeds, int n) {
  RF_REQUIRE(p->nranks == 1 || p->comm || p->direct_standin, "virtual ranks linked for the direct exchange run step by step (rf_slab_forward, rf_slab_backward)");
  void* Wb[2] = {p->W, p->W2};
  void* Rb[2] = {p->R, p->R2};
  hipEvent_t *ev_fwd = p->pev, *ev_bar = p->pev + 2, *ev_z = p->pev + 4;
  const bool two = p->direct_overlap != 0;
  hipStream_t A = p->stream, Y = two ? p->comm_stream : p->stream;
  RF_HIP(hipEventRecord(p->ev[0], A));
  if (two) { RF_HIP(hipEventRecord(ev_z[1], A)); RF_HIP(hipStreamWaitEvent(Y, ev_z[1], 0)); }     // whatever the plan's stream did before
  if (int rc = direct_barrier(p, Y)) return rc;                   // B(-1): nobody is still reading (or sending from) a receive buffer
  for (int i = 0; i <= n; ++i) {
    const int b = i & 1, pb = (i - 1) & 1;
    if (i < n) {
      // (sub-slab by sub-slab when the plan chunks: x(c) on A, the stores of sub-slab c on Y behind an event)
      if (int rc = direct_forward(p, make_gen(p, seeds[i], RF_NOISE_NATIVE, false), nullptr, Wb[b], b, A, Y, false)) return rc;
      (void)ev_fwd;
    }
    if (i >= 1) {
      if (two) RF_HIP(hipStreamWaitEvent(A, ev_bar[pb], 0));
      if (int rc = queue_z_slab(p, Rb[pb], Wb[pb], p->stats + 2 * (i - 1), A)) return rc;
      if (two) RF_HIP(hipEventRecord(ev_z[pb], A));
    }
    if (i < n) {
      if (two && i >= 1) RF_HIP(hipStreamWaitEvent(Y, ev_z[pb], 0));
      if (int rc = direct_barrier(p, Y)) return rc;               // B(i)
      if (two) RF_HIP(hipEventRecord(ev_bar[b], Y));
    }
  }
  if (p->nranks > 1 && p->comm) {                                 // the moments of all n realisations: one all-reduce, on the stream that drives the communicator
    const int lb = (n - 1) & 1;
    if (two) RF_HIP(hipStreamWaitEvent(Y, ev_z[lb], 0));
    RF_NCCL(g_rccl.AllReduce(p->stats, p->stats, 2 * (size_t)n, ncclFloat64, ncclSum, p->comm, Y));
    if (two) { RF_HIP(hipEventRecord(ev_bar[lb], Y)); RF_HIP(hipStreamWaitEvent(A, ev_bar[lb], 0)); }
  }
  RF_HIP(hipEventRecord(p->ev[4], A));
  p->cur = Wb[(n - 1) & 1];
  p->stats_slot = n - 1;
  p->timed = false;
  p->real_valid = !p->direct_standin;
  p->stats_valid = true;
  return 0;
}

// Pipelined batch on the slab path: realisation i+1's generation + x + y passes (compute stream) run
// while realisation i's all-to-all is in flight (exchange stream); two (send, receive) buffer pairs.
//   compute: x,y(0) | x,y(1)   z(0) | x,y(2)   z(1) | ...            (in order on p->stream)
//   exchange:        | exch(0)       | exch(1)       | ...            (in order on p->comm_stream)
// z(i) waits for exch(i); exch(i) waits for x,y(i) and -- because it overwrites R[i%2] -- for z(i-2).
// One all-reduce of all n (sum, sumsq) pairs at the end.
int slab_batch(rf_plan* p, const uint64_t* seeds, int n) {
  if (p->replicate && p->nranks > 1) {          // no exchange to overlap: realisations back to back, one all-reduce at the end
    if (p->stats_cap < n) {
      RF_HIP(hipStreamSynchronize(p->stream));
      if (p->stats) RF_HIP(hipFree(p->stats));
      p->stats = nullptr;
      RF_HIP(hipMalloc((void**)&p->stats, 2 * (size_t)(n + 64) * sizeof(double)));
      p->stats_cap = n + 64;
    }
    const double scale = 1.0 / ((double)p->nx * (double)p->ny * (double)p->nz);
    RF_HIP(hipEventRecord(p->ev[0], p->stream));
    for (int i = 0; i < n; ++i) {
      if (int rc = queue_xy(p, make_gen(p, seeds[i], RF_NOISE_NATIVE, false), nullptr, p->W, p->stream, false)) return rc;
      RF_HIP(launch_row_c2r(p->f64, (int)p->nzc, p->W, (long long)p->nxl * p->ny, scale, p->tw_z, p->partials, p->stream));
      RF_HIP(launch_reduce_partials(p->partials, p->npartials, p->stats + 2 * i, p->partials + 2 * p->npartials, p->stream));
    }
    if (p->comm) RF_NCCL(g_rccl.AllReduce(p->stats, p->stats, 2 * (size_t)n, ncclFloat64, ncclSum, p->comm, p->stream));
    RF_HIP(hipEventRecord(p->ev[4], p->stream));
    p->cur = p->W;
    p->stats_slot = n - 1;
    p->timed = false;
    p->real_valid = true;
    p->stats_valid = true;
    return 0;
  }
  if (int rc = ensure_batch_buffers(p)) return rc;
  if (p->stats_cap < n) {
    RF_HIP(hipStreamSynchronize(p->stream));
    drop_graphs(p);
    if (p->stats) RF_HIP(hipFree(p->stats));
    p->stats = nullptr;
    RF_HIP(hipMalloc((void**)&p->stats, 2 * (size_t)(n + 64) * sizeof(double)));
    p->stats_cap = n + 64;
  }
  if (direct_active(p)) return slab_batch_direct(p, seeds, n);
  void* Wb[2] = {p->W, p->W2};
  void* Rb[2] = {p->R, p->R2};
  hipEvent_t *ev_fwd = p->pev, *ev_exch = p->pev + 2, *ev_z = p->pev + 4;
  hipStream_t A = p->stream, C = p->comm_stream;
  RF_HIP(hipEventRecord(p->ev[0], A));
  for (int i = 0; i <= n; ++i) {
    if (i < n) {
      const int b = i & 1;
      if (int rc = queue_xy(p, make_gen(p, seeds[i], RF_NOISE_NATIVE, false), nullptr, Wb[b], A, false)) return rc;
      RF_HIP(hipEventRecord(ev_fwd[b], A));
      RF_HIP(hipStreamWaitEvent(C, ev_fwd[b], 0));
      if (i >= 2) RF_HIP(hipStreamWaitEvent(C, ev_z[b], 0));          // R[b] still being read by z(i-2)?
      if (int rc = queue_exchange_rccl(p, Wb[b], Rb[b], C)) return rc;
      RF_HIP(hipEventRecord(ev_exch[b], C));
    }
    if (i >= 1) {
      const int pb = (i - 1) & 1;
      RF_HIP(hipStreamWaitEvent(A, ev_exch[pb], 0));
      if (int rc = queue_z_slab(p, Rb[pb], Wb[pb], p->stats + 2 * (i - 1), A)) return rc;
      RF_HIP(hipEventRecord(ev_z[pb], A));
    }
  }
  if (p->nranks > 1 && p->comm) {        // (no communicator: a virtual rank under rf_slab_set_exchange_standin, local moments only)
    // ONE communicator is only ever driven from ONE stream inside a batch: the moments' all-reduce goes to the exchange
    // stream too, behind the last z pass (event), and the compute stream waits for it
    const int lb = (n - 1) & 1;
    RF_HIP(hipStreamWaitEvent(C, ev_z[lb], 0));
    RF_NCCL(g_rccl.AllReduce(p->stats, p->stats, 2 * (size_t)n, ncclFloat64, ncclSum, p->comm, C));
    RF_HIP(hipEventRecord(ev_exch[lb], C));
    RF_HIP(hipStreamWaitEvent(A, ev_exch[lb], 0));
  }
  RF_HIP(hipEventRecord(p->ev[4], A));
  p->cur = Wb[(n - 1) & 1];
  p->stats_slot = n - 1;
  p->timed = false;
  p->real_valid = !(p->nranks > 1 && !p->comm);        // (a stand-in exchange leaves this rank's own data in the segments: not a field)
  p->stats_valid = true;
  return 0;
}

// x planes per slab of the y / z passes of a single-rank plan (0: whole-grid passes).  The y pass of a slab leaves it in the
// 256 MiB Infinity Cache for the z pass that follows at once; slabs much smaller than the cache make the launches too small
// (1024^3 float32, MI355X, 20 realisations per graph: 4.74 ms whole grid, 4.53-4.57 ms with 64-plane = 256 MiB slabs, 4.66 with 56,
// 4.70 with 72 ... 128, 4.74 with 48, 5.6 ms with 16; 2048^3: 44.5 -> 43.7-43.9 ms; 512^3: 0.545 -> 0.537 ms).
int yz_slab_planes(const rf_plan* p) {
  if (p->nranks > 1 || p->force_slab || p->generic || p->unpacked || p->yz_slab == 0) return 0;
  const long long plane = (long long)p->ny * p->nzl * (long long)p->csize;
  long long B = p->yz_slab;
  if (B < 0) {
    if (p->f64 && !p->sink_host) return 0;      // float64 passes: within 1 % of the whole-grid launches at every slab size measured (9.32-9.53 against 9.41 ms); slabs anyway when a host sink wants them one by one
    const long long target = 256LL << 20;              // the Infinity Cache (RF_FLAG_YZ_SLAB_PLANES sets another slab size per plan)
    B = 1;
    while (2 * B * plane <= target) B *= 2;
  }
  if (B >= p->nx) return 0;
  // whole z-pass workgroups per slab (the last slab may be smaller), at the offsets the whole-grid launch would give them
  const long long tiles1 = row_c2r_tiles(p->f64, (int)p->nzc, p->ny);
  if (tiles1 <= 0 || tiles1 * p->nx != p->npartials) return 0;
  return (int)B;
}

// y and z passes of a single-rank plan + the moments into stats_out[0..1], slab by slab when yz_slab_planes() says so.  The x
// pass has left its output either in W (plain layout: both passes in place) or in the blocked intermediate X (y pass in place
// on X, z pass gathering X -> W).
// Host sink (rf_set_host_sink).  The z pass of a slab of x planes has been queued on s: mark that point with an event ...
int sink_mark(rf_plan* p, int slab, hipStream_t s) {
  while ((int)p->sink_ev.size() <= slab) { hipEvent_t e; RF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->sink_ev.push_back(e); }
  RF_HIP(hipEventRecord(p->sink_ev[slab], s));
  return 0;
}
// ... and, once the launches of the NEXT slab are queued too (the GPU has work while the host waits here), copy planes [x0, x0 + nb)
// to the armed host buffer: an ordinary device -> host copy into pageable memory on dl_stream behind the slab's event -- it returns
// when the rows are in the caller's memory.  (The host buffer is deliberately NOT registered with hipHostRegister: round 6 first
// built it that way, and later device -> host copies of the same process then aborted now and then inside the runtime.)
int sink_copy(rf_plan* p, const void* W, long long x0, long long nb, int slab) {
  RF_HIP(hipStreamWaitEvent(p->dl_stream, p->sink_ev[slab], 0));
  const size_t rsize = p->csize / 2, width = (size_t)p->nz * rsize;
  const size_t hpitch = p->sink_layout == RF_LAYOUT_PADDED ? (size_t)(p->nz + 2) * rsize : width;
  RF_HIP(hipMemcpy2DAsync((char*)p->sink_host + (size_t)x0 * p->ny * hpitch, hpitch, (const char*)W + (size_t)x0 * p->ny * width, width, width,
                          (size_t)nb * p->ny, hipMemcpyDeviceToHost, p->dl_stream));
  RF_HIP(hipStreamSynchronize(p->dl_stream));
  return 0;
}
// ... after the last slab: the armed call returns with the field on the host (one shot)
int sink_finish(rf_plan* p) {
  p->sink_host = nullptr;
  p->sink_delivered = true;
  return 0;
}

int queue_yz(rf_plan* p, void* W, hipStream_t s, double* stats_out, bool timed) {
  const long long nzl = p->nzl;
  bool sink = p->sink_host != nullptr && !p->zscale;
  if (sink) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    sink = cap == hipStreamCaptureStatusNone;      // (a captured batch keeps its fields on the device)
  }
  const double scale = 1.0 / ((double)p->nx * (double)p->ny * (double)p->nz);
  const bool xp = p->X && xpose_ok(p);
  const long long rb = xpose_row_block(p), tc = col_tile_cols(p->f64, p->ny);
  const ColGeom gy = xp ? xblock_y_geom(p->nx, p->ny, nzl, tc, rb) : ColGeom{nzl, (long long)p->ny * nzl, nzl};
  long long B = yz_slab_planes(p);
  if (xp && B > 0 && (B % rb || (B & (B - 1)) || p->nx % B)) B = 0;      // a slab of X is whole x blocks, a power of two of them
  if (B <= 0) B = p->nx;
  const int nslab = (int)((p->nx + B - 1) / B);
  const long long plane = (long long)p->ny * nzl * (long long)p->csize, tiles_per_plane = p->npartials / p->nx;
  if (timed) {
    while ((int)p->slab_ev.size() < 2 * nslab) { hipEvent_t e; RF_HIP(hipEventCreate(&e)); p->slab_ev.push_back(e); }
    p->slab_timed = nslab;
  }
  // untimed single-rank float32 realisations at the sizes rf_k_yz.hip serves: the z pass of slab i and the y pass of slab i + 1 share a
  // launch (the next slab's y tiles fill the CUs the draining z pass leaves idle); timed calls keep one launch per pass and slab, so
  // that rf_kernel_ms still means what it says
  // (rf_set_merged_yz(2) merges timed calls too, with an event behind every launch: rf_merged_yz_ms)
  // (the last slab may be smaller -- a slab size that does not divide nx: RF_FLAG_YZ_SLAB_PLANES -- as long as both sizes fit the kernel)
  const long long Blast = p->nx - (long long)(nslab - 1) * B;
  if ((timed ? p->yz_merge >= 2 : p->yz_merge >= 1) && !xp && !p->zscale && nslab > 1 &&
      yz_merged_fits(p->f64, p->ny, (int)p->nzc, gy, B * p->ny, B * nzl) &&
      (Blast == B || yz_merged_fits(p->f64, p->ny, (int)p->nzc, gy, B * p->ny, Blast * nzl))) {
    if (timed) {
      while ((int)p->slab_ev.size() < nslab + 1) { hipEvent_t e; RF_HIP(hipEventCreate(&e)); p->slab_ev.push_back(e); }
      p->slab_merged = nslab;
      p->slab_timed = 0;       // (rf_kernel_ms: the merged form's events apply, the per-pass pairs of slab_ev were not recorded by this call)
    }
    RF_HIP(launch_col_plain(p->f64, p->ny, +1, W, gy, B * nzl, p->tw_y, s));
    if (timed) RF_HIP(hipEventRecord(p->slab_ev[0], s));
    for (int i = 0; i < nslab; ++i) {
      char* Ws = (char*)W + (long long)i * B * plane;
      double* part = p->partials + 2 * (long long)i * B * tiles_per_plane;
      const long long nb = i + 1 < nslab ? B : Blast, nb_next = i + 2 < nslab ? B : Blast;       // planes of slab i / of slab i + 1
      if (i + 1 < nslab)
        RF_HIP(launch_yz_merged(p->f64, p->ny, (int)p->nzc, Ws, nb * p->ny, scale, p->tw_z, part, Ws + B * plane, gy, nb_next * nzl, p->tw_y, s));
      else
        RF_HIP(launch_row_c2r(p->f64, (int)p->nzc, Ws, nb * p->ny, scale, p->tw_z, part, s));
      if (timed) RF_HIP(hipEventRecord(p->slab_ev[i + 1], s));
      if (sink) {
        if (int rc = sink_mark(p, i, s)) return rc;
        if (i > 0) if (int rc = sink_copy(p, W, (long long)(i - 1) * B, B, i - 1)) return rc;       // (slab i - 1, while the GPU runs slab i)
      }
    }
    if (timed) { RF_HIP(hipEventRecord(p->ev[2], s)); RF_HIP(hipEventRecord(p->ev[3], s)); }
    RF_HIP(launch_reduce_partials(p->partials, p->npartials, stats_out, p->partials + 2 * p->npartials, s));
    if (timed) RF_HIP(hipEventRecord(p->ev[4], s));
    if (sink) {
      if (int rc = sink_copy(p, W, (long long)(nslab - 1) * B, Blast, nslab - 1)) return rc;
      return sink_finish(p);
    }
    return 0;
  }
  for (int i = 0; i < nslab; ++i) {
    const long long x0 = (long long)i * B, nb = x0 + B <= p->nx ? B : p->nx - x0;      // planes [x0, x0 + nb)
    char* Ws = (char*)W + x0 * plane;
    char* Xs = xp ? (char*)p->X + x0 * plane : nullptr;        // (x blocks are contiguous and as large as their planes)
    if (xp) RF_HIP(launch_col_xpose(p->f64, p->ny, Xs, gy, Xs, gy, nb * nzl, p->tw_y, s));
    else RF_HIP(launch_col_plain(p->f64, p->ny, +1, Ws, gy, nb * nzl, p->tw_y, s));
    if (timed) RF_HIP(hipEventRecord(p->slab_ev[2 * i], s));
    double* part = p->partials + 2 * x0 * tiles_per_plane;
    if (xp) RF_HIP(launch_row_c2r_xgather(p->f64, (int)p->nzc, Xs, Ws, nb * p->ny, scale, (int)tc, (int)rb, p->ny, p->tw_z, part, s));
    else if (p->zscale) RF_HIP(launch_row_c2r_zscale(p->f64, (int)p->nzc, Ws, nb * p->ny, scale, p->zscale, p->tw_z, part, s));
    else RF_HIP(launch_row_c2r(p->f64, (int)p->nzc, Ws, nb * p->ny, scale, p->tw_z, part, s));
    if (timed) RF_HIP(hipEventRecord(p->slab_ev[2 * i + 1], s));
    if (sink) {
      if (int rc = sink_mark(p, i, s)) return rc;
      if (i > 0) if (int rc = sink_copy(p, W, x0 - B, B, i - 1)) return rc;                          // (slab i - 1, while the GPU runs slab i)
    }
  }
  if (timed) { RF_HIP(hipEventRecord(p->ev[2], s)); RF_HIP(hipEventRecord(p->ev[3], s)); }   // (rf_kernel_ms sums the slab events)
  RF_HIP(launch_reduce_partials(p->partials, p->npartials, stats_out, p->partials + 2 * p->npartials, s));
  if (timed) RF_HIP(hipEventRecord(p->ev[4], s));
  if (sink) {
    const long long xl = (long long)(nslab - 1) * B;
    if (int rc = sink_copy(p, W, xl, p->nx - xl, nslab - 1)) return rc;
    return sink_finish(p);
  }
  return 0;
}

// the whole single-rank pipeline: x pass (generation or API k-space fused into its load; into the transposed intermediate
// when the plan uses it), then queue_yz
int queue_xyz(rf_plan* p, const GenParams& gp, const void* kspace, void* W, hipStream_t s, double* stats_out, bool timed) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  if (cap == hipStreamCaptureStatusNone)           // (no allocation inside a graph capture: batch_prepare() has done it)
    if (int rc = ensure_x(p)) return rc;
  const bool xp = p->X && xpose_ok(p);
  if (int rc = queue_x(p, gp, kspace, xp ? p->X : W, s, timed)) return rc;
  if (timed) RF_HIP(hipEventRecord(p->ev[1], s));
  return queue_yz(p, W, s, stats_out, timed);
}

// the launches behind the sequences of rf_generic.h (generic_c2r_seq / generic_r2c_seq / generic_c2c_seq) on the plan's stream
struct HipGenericOps {
  rf_plan* p;
  hipStream_t s;
  const void* root(int which) const { return which == 0 ? p->tw_x : (which == 1 ? p->tw_y : p->tw_z); }
  int axis(const void* src, void* dst, const GenericAxis& ax, long long stride, long long inner, long long outer, long long nlines, int which, int sign, double scale) {
    RF_HIP(launch_generic_axis(p->f64, src, dst, ax, stride, inner, outer, nlines, root(which), sign, scale, s));
    return 0;
  }
  int lines(const void* src, void* dst, const GenericLines& L, int which) {
    RF_HIP(launch_generic_lines(p->f64, src, dst, L, root(which), s));
    return 0;
  }
  int row_c2r(const void* G, void* W, double scale) {
    RF_HIP(launch_generic_row_c2r(p->f64, G, W, p->gdims.az, (long long)p->nx * p->ny, p->tw_z, scale, p->partials, s));
    return 0;
  }
  int row_r2c(const void* W, void* G) {
    RF_HIP(launch_generic_row_r2c(p->f64, W, G, p->gdims.az, (long long)p->nx * p->ny, p->tw_z, s));
    return 0;
  }
  int untangle(const void* G, void* Z) { RF_HIP(launch_generic_untangle(p->f64, G, Z, (int)p->nzc, (long long)p->nx * p->ny, p->tw_z, s)); return 0; }
  int tangle(const void* Z, void* G) { RF_HIP(launch_generic_tangle(p->f64, Z, G, (int)p->nzc, (long long)p->nx * p->ny, p->tw_z, s)); return 0; }
  int moments(const void* W) { RF_HIP(launch_generic_moments(p->f64, W, (long long)p->nx * p->ny * p->nz, p->partials, p->npartials, s)); return 0; }
  int copy(void* dst, const void* src, size_t bytes) { RF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)); return 0; }
};
bool generic_any_long(const rf_plan* p) { return p->gdims.lx.split() || p->gdims.ly.split() || p->gdims.lz.split(); }
// the second scratch array, for plans with an axis in the four-step form
int ensure_g2(rf_plan* p) {
  if (!p->G2 && generic_any_long(p)) RF_HIP(hipMalloc(&p->G2, p->unpacked ? p->w_bytes : p->k_bytes));
  return 0;
}

// non-power-of-two grid: API-layout half spectrum K -> x pass into G -> y pass -> contiguous c2r pass into W (rf_generic.h
// generic_c2r_seq: axes too long for one line of the LDS take the four-step form through the scratch arrays), (sum, sumsq) into stats_out
int generic_c2r(rf_plan* p, const void* K, double* stats_out) {
  if (int rc = ensure_g(p)) return rc;
  if (int rc = ensure_g2(p)) return rc;
  HipGenericOps ops{p, p->stream};
  const double scale = 1.0 / ((double)p->nx * (double)p->ny * (double)p->nz);
  if (int rc = generic_c2r_seq(ops, p->gdims, K, p->G, p->G2, p->W, scale)) return rc;
  RF_HIP(launch_reduce_partials(p->partials, p->npartials, stats_out, p->partials + 2 * p->npartials, p->stream));
  return 0;
}

// one realisation / transform on the plan's stream into the primary buffer
int queue_c2r(rf_plan* p, const GenParams& gp, const void* kspace) {
  if (p->generic) {
    RF_HIP(hipEventRecord(p->ev[0], p->stream));
    if (!kspace) {                                  // rows K,T,R,S into the API-layout buffer first
      RF_REQUIRE(gp.noise_mode != NOISE_EXTERNAL || p->noise_resident, "no float64 deviates resident on the device");
      if (int rc = ensure_k(p)) return rc;
      RF_HIP(launch_gen_kspace(p->f64, p->K, gp, p->stream));
      p->k_valid = true;
      p->aux_valid = false;
      kspace = p->K;
    }
    if (int rc = generic_c2r(p, kspace, p->stats)) return rc;
    RF_HIP(hipEventRecord(p->ev[4], p->stream));
    p->timed = false;                               // no per-kernel events on this path
    p->cur = p->W;
    p->stats_slot = 0;
    p->real_valid = true;
    p->stats_valid = true;
    return 0;
  }
  p->slab_timed = 0;                            // (set again by queue_yz when this call runs the y / z passes slab by slab, timed)
  p->slab_merged = 0;
  if (p->timed) RF_HIP(hipEventRecord(p->ev[0], p->stream));
  if (p->nranks == 1 && !p->force_slab) {       // one GPU: x pass, then the y / z passes (slab by slab on large grids)
    if (int rc = queue_xyz(p, gp, kspace, p->W, p->stream, p->stats, p->timed)) return rc;
    p->cur = p->W;
    p->stats_slot = 0;
    p->real_valid = true;
    p->stats_valid = true;
    return 0;
  }
  if (direct_active(p)) {
    // ONE realisation of a plan in direct mode: the storing y pass is the exchange.  Unchunked: x, B, y, B, z on the plan's stream.  In
    // sub-slabs: the stores of sub-slab c go to the exchange stream and run beside the x pass of sub-slab c + 1.  The barrier in front
    // keeps the stores away from a peer that still reads (or, in the forward transform's reverse exchange, sends from) its receive
    // buffer; the one behind tells every rank that all stores have landed.  The communicator is driven from ONE stream per call.
    RF_REQUIRE(p->nranks == 1 || p->comm || p->direct_standin, "virtual ranks linked for the direct exchange run step by step (rf_slab_forward, rf_slab_backward)");
    const int C = slab_chunks(p);
    const bool two = p->direct_overlap != 0 && C > 1;
    if (int rc = ensure_comm_stream(p)) return rc;
    while ((int)p->chunk_ev.size() < C + 1) { hipEvent_t e; RF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->chunk_ev.push_back(e); }
    hipStream_t A = p->stream, Y = two ? p->comm_stream : p->stream;
    if (two) { RF_HIP(hipEventRecord(p->chunk_ev[C], A)); RF_HIP(hipStreamWaitEvent(Y, p->chunk_ev[C], 0)); }
    if (int rc = direct_barrier(p, Y)) return rc;
    if (int rc = direct_forward(p, gp, kspace, p->W, 0, A, Y, p->timed)) return rc;
    if (int rc = direct_barrier(p, Y)) return rc;
    if (two) { RF_HIP(hipEventRecord(p->chunk_ev[C], Y)); RF_HIP(hipStreamWaitEvent(A, p->chunk_ev[C], 0)); }
    if (int rc = queue_z_slab(p, p->R, p->W, p->stats, A)) return rc;
    if (p->timed) RF_HIP(hipEventRecord(p->ev[3], A));
    if (p->nranks > 1 && p->comm) {
      if (two) { RF_HIP(hipEventRecord(p->chunk_ev[C], A)); RF_HIP(hipStreamWaitEvent(Y, p->chunk_ev[C], 0)); }
      RF_NCCL(g_rccl.AllReduce(p->stats, p->stats, 2, ncclFloat64, ncclSum, p->comm, Y));
      if (two) { RF_HIP(hipEventRecord(p->chunk_ev[C], Y)); RF_HIP(hipStreamWaitEvent(A, p->chunk_ev[C], 0)); }
    }
    if (p->timed) RF_HIP(hipEventRecord(p->ev[4], A));
    p->stats_slot = 0;
    p->stats_valid = true;
    if (p->direct_standin) p->real_valid = false;       // (the stores went to this rank's own buffers: no field came out)
    return 0;
  }
  if (slab_chunks(p) > 1) {
    // ONE realisation, exchange overlapped with its own forward half: sub-slab c is generated and x / y-transformed on the plan's
    // stream, its grouped send / receive goes to the exchange stream behind an event, the forward half of sub-slab c + 1 follows
    // at once; the gathering z pass waits for the last sub-slab's exchange.  (The communicator is driven from the exchange stream
    // only, the final all-reduce of the moments too.)
    const int C = slab_chunks(p);
    if (int rc = ensure_comm_stream(p)) return rc;
    while ((int)p->chunk_ev.size() < C + 1) { hipEvent_t e; RF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->chunk_ev.push_back(e); }
    hipStream_t A = p->stream, X = p->comm_stream;
    RF_HIP(hipEventRecord(p->chunk_ev[C], A));                 // (whatever used R / the communicator before on the plan's stream)
    RF_HIP(hipStreamWaitEvent(X, p->chunk_ev[C], 0));
    for (int c = 0; c < C; ++c) {
      if (int rc = queue_xy_chunk(p, gp, kspace, p->W, A, c)) return rc;
      RF_HIP(hipEventRecord(p->chunk_ev[c], A));
      RF_HIP(hipStreamWaitEvent(X, p->chunk_ev[c], 0));
      if (int rc = queue_exchange_rccl(p, p->W, p->R, X, c)) return rc;
    }
    if (p->timed) { RF_HIP(hipEventRecord(p->ev[5], A)); p->repair_timed = false; RF_HIP(hipEventRecord(p->ev[1], A)); RF_HIP(hipEventRecord(p->ev[2], A)); }
    RF_HIP(hipEventRecord(p->chunk_ev[C], X));
    RF_HIP(hipStreamWaitEvent(A, p->chunk_ev[C], 0));
    if (int rc = queue_z_slab(p, p->R, p->W, p->stats, A)) return rc;
    if (p->timed) RF_HIP(hipEventRecord(p->ev[3], A));
    if (p->nranks > 1 && p->comm) {
      RF_HIP(hipEventRecord(p->chunk_ev[C], A));
      RF_HIP(hipStreamWaitEvent(X, p->chunk_ev[C], 0));
      RF_NCCL(g_rccl.AllReduce(p->stats, p->stats, 2, ncclFloat64, ncclSum, p->comm, X));
      RF_HIP(hipEventRecord(p->chunk_ev[C], X));
      RF_HIP(hipStreamWaitEvent(A, p->chunk_ev[C], 0));
    }
    if (p->timed) RF_HIP(hipEventRecord(p->ev[4], A));
    p->stats_slot = 0;
    p->stats_valid = true;
    if (p->nranks > 1 && !p->comm) p->real_valid = false;       // (stand-in exchange: not a field)
    return 0;
  }
  if (int rc = queue_xy(p, gp, kspace, p->W, p->stream, p->timed)) return rc;
  if (p->replicate && p->nranks > 1) {          // the local array already is this rank's x slab [nxl][ny][nz/2]
    const double scale = 1.0 / ((double)p->nx * (double)p->ny * (double)p->nz);
    RF_HIP(launch_row_c2r(p->f64, (int)p->nzc, p->W, (long long)p->nxl * p->ny, scale, p->tw_z, p->partials, p->stream));
    RF_HIP(launch_reduce_partials(p->partials, p->npartials, p->stats, p->partials + 2 * p->npartials, p->stream));
    if (p->timed) RF_HIP(hipEventRecord(p->ev[3], p->stream));
    if (p->comm) RF_NCCL(g_rccl.AllReduce(p->stats, p->stats, 2, ncclFloat64, ncclSum, p->comm, p->stream));
    if (p->timed) RF_HIP(hipEventRecord(p->ev[4], p->stream));
    p->cur = p->W;
    p->real_valid = true;
    p->stats_slot = 0;
    p->stats_valid = true;
    return 0;
  }
  if (p->nranks > 1 || p->force_slab) {
    if (int rc = queue_exchange_rccl(p, p->W, p->R, p->stream)) return rc;
    if (int rc = queue_z_slab(p, p->R, p->W, p->stats, p->stream)) return rc;
    if (p->timed) RF_HIP(hipEventRecord(p->ev[3], p->stream));
    // global (sum, sumsq): one 2-double all-reduce
    if (p->nranks > 1 && p->comm) RF_NCCL(g_rccl.AllReduce(p->stats, p->stats, 2, ncclFloat64, ncclSum, p->comm, p->stream));
    if (p->timed) RF_HIP(hipEventRecord(p->ev[4], p->stream));
    p->stats_slot = 0;
    p->stats_valid = true;
    if (p->nranks > 1 && !p->comm) p->real_valid = false;       // (stand-in exchange: not a field)
    return 0;
  }
  return fail(1, "queue_c2r: unreachable");
}

template <typename T> int upload_twiddles(void** dst, int n) {
  auto w = make_twiddles<T>(n);
  hipError_t e = hipMalloc(dst, w.size() * sizeof(cplx<T>));
  if (e != hipSuccess) return fail(2, std::string("hipMalloc twiddles: ") + hipGetErrorString(e));
  e = hipMemcpy(*dst, w.data(), w.size() * sizeof(cplx<T>), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(2, std::string("hipMemcpy twiddles: ") + hipGetErrorString(e));
  return 0;
}

int shape_check(int nx, int ny, int nz, int f64, std::string* why, int nranks = 1) {
  auto bad = [&](const std::string& m) { if (why) *why = m; return 1; };
  if (nx <= 0 || ny <= 0 || nz <= 0) return bad("grid dimensions must be positive");
  if (nz % 4) return bad("nz must be a multiple of 4 (packed layout, transform.py:53-56)");
  if (!col_size_supported(nx) || !col_size_supported(ny))
    return bad("HIP path needs nx, ny in {8,16,...,2048} (powers of two)");
  if (!row_size_supported(nz / 2)) return bad("HIP path needs nz in {16,32,...,2048} (powers of two)");
  const long long nzc = nz / 2;
  if (nranks > 1 && (nx % nranks || nzc % nranks || (nzc / nranks) % 2))
    return bad("nx and nz/2 must be divisible by the number of ranks (and nz/(2*ranks) must be even)");
  const long long nzl = nzc / nranks;
  if (((long long)ny * nzl) % col_tile_cols(f64, nx)) return bad("ny*nz/(2*ranks) is not a multiple of the x-pass tile width");
  if (((long long)nx * nzl) % col_tile_cols(f64, ny)) return bad("nx*nz/(2*ranks) is not a multiple of the y-pass tile width");
  return 0;
}

}  // namespace rfc

extern "C" {

int rf_version(void) { return RF_ABI_VERSION; }

unsigned rf_abi_features(void) {
  return RF_FEATURE_REALISE | RF_FEATURE_R2C | RF_FEATURE_C2C | RF_FEATURE_LOGNORMAL | RF_FEATURE_POTENTIAL | RF_FEATURE_LENSING |
         RF_FEATURE_MT19937 | RF_FEATURE_MT19937_SHARED | RF_FEATURE_MULTI_RANK | RF_FEATURE_GENERIC_SHAPES | RF_FEATURE_EXCHANGE_CHUNKS |
         RF_FEATURE_DIRECT_EXCHANGE | RF_FEATURE_DIAGNOSTICS;
}

const char* rf_last_error(void) { return g_err.c_str(); }

int rf_device_count(int* count) {
  RF_REQUIRE(count != nullptr, "count is null");
  RF_HIP(hipGetDeviceCount(count));
  return 0;
}

// 1: tiled power-of-two kernels; 2: the generic path for other even shapes (rf_generic.h); 0: not supported
int rf_shape_supported(int nx, int ny, int nz) {
  if (shape_check(nx, ny, nz, 0, nullptr) == 0 && shape_check(nx, ny, nz, 1, nullptr) == 0) return 1;
  GenericAxis a, b, c;
  return generic_shape(nx, ny, nz, 0, a, b, c) ? 2 : 0;          // (complex128 plans: axes up to 4096 -- rf_plan_create says so)
}

// the same for one dtype: exactly what rf_plan_create on one rank accepts (complex128 plans: generic axes up to 4096)
int rf_shape_supported_dtype(int nx, int ny, int nz, int dtype) {
  if (dtype != RF_F32 && dtype != RF_F64) return 0;
  if (shape_check(nx, ny, nz, dtype, nullptr) == 0) return 1;
  GenericAxis a, b, c;
  return generic_shape(nx, ny, nz, dtype == RF_F64, a, b, c) ? 2 : 0;
}

int rf_plan_create(rf_plan** out, int nx, int ny, int nz, int dtype, int device, int nranks, int rank) {
  RF_REQUIRE(out != nullptr, "plan pointer is null");
  *out = nullptr;
  RF_REQUIRE(dtype == RF_F32 || dtype == RF_F64, "dtype must be RF_F32 or RF_F64");
  std::string why;
  RF_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "invalid nranks/rank");
  GenericDims gd;
  bool generic = false;
  if (shape_check(nx, ny, nz, dtype, &why, nranks)) {
    generic = nranks == 1 && generic_dims(nx, ny, nz, dtype == RF_F64, true, gd);
    if (!generic)
      return fail(1, "unsupported shape: " + why + (nranks == 1 ? std::string(" (and not an even shape whose axes fit one line of the LDS -- ") + (dtype == RF_F64 ? "4096 points on complex128 plans" : "8192 points on complex64 plans") + " -- or split into two factors that do, either: rf_shape_supported_dtype)" : ""));
  }
  RF_HIP(hipSetDevice(device));
  rf_plan* p = new rf_plan();
  p->generic = generic;
  if (generic) { p->gdims = gd; p->gax = gd.ax; p->gay = gd.ay; p->gaz = gd.az; }
  p->nx = nx; p->ny = ny; p->nz = nz; p->nzc = nz / 2; p->f64 = dtype; p->device = device;
  p->nranks = nranks; p->rank = rank;
  p->csize = dtype ? 16 : 8;
  p->nxl = nx / nranks; p->nzl = p->nzc / nranks; p->kz0 = rank * p->nzl;
  p->w_bytes = (size_t)nx * ny * p->nzl * p->csize;       // == nxl * ny * nzc: the local share of the field
  p->k_bytes = (size_t)nx * ny * (p->nzl + 1) * p->csize;    // side arrays: this rank's planes + the Nyquist plane
  // float32: an even pitch (nzl is even), so that the fused store writes 16-byte aligned cell pairs; on large grids 64 cells
  // beyond the row instead of 2 -- the time of the two strided store streams of rf_realise_potential depends on where the
  // allocator puts the two arrays (tools/frag_probe.py, 1024^3: 5.6 / 6.8 / 6.9 ms in three allocation histories with pitch
  // nz/2 + 2; 5.5 / 6.1 / 6.0 ms with nz/2 + 64), which a row stride further from the field's 4 MiB softens
  p->ppitch = dtype ? (int)p->nzl + 1 : (int)p->nzl + (p->nzl >= 256 ? 64 : 2);
  p->p_bytes = (size_t)nx * ny * p->ppitch * p->csize;
  auto cleanup = [&](int rc) { rf_plan_destroy(p); return rc; };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking)) != hipSuccess)
    return cleanup(fail(2, std::string("hipStreamCreate: ") + hipGetErrorString(e)));
  p->stream = p->own_stream;
  if ((e = hipMalloc(&p->W, p->w_bytes)) != hipSuccess || (nranks > 1 && (e = hipMalloc(&p->R, p->w_bytes)) != hipSuccess))
    return cleanup(fail(2, std::string("hipMalloc field buffer: ") + hipGetErrorString(e)));
  int rc = 0;
  if (dtype) {
    rc = upload_twiddles<double>(&p->tw_x, nx);
    if (!rc) rc = upload_twiddles<double>(&p->tw_y, ny);
    if (!rc) rc = upload_twiddles<double>(&p->tw_z, nz);
  } else {
    rc = upload_twiddles<float>(&p->tw_x, nx);
    if (!rc) rc = upload_twiddles<float>(&p->tw_y, ny);
    if (!rc) rc = upload_twiddles<float>(&p->tw_z, nz);
  }
  if (rc) return cleanup(rc);
  p->npartials = generic ? (gd.lz.split() ? 1024 : generic_row_blocks(dtype, gd.az, (long long)nx * ny))      // (long rows: the moments are a pass of their own, 1024 blocks)
               : nranks > 1 ? row_c2r_tiles(dtype, p->nzc, (long long)p->nxl * ny) : row_c2r_tiles(dtype, p->nzc, (long long)nx * ny);
  p->stats_cap = 64;
  if ((e = hipMalloc((void**)&p->partials, (2 * p->npartials + 512) * sizeof(double))) != hipSuccess ||
      (e = hipMalloc((void**)&p->stats, 2 * p->stats_cap * sizeof(double))) != hipSuccess ||
      (e = hipMalloc((void**)&p->coll_scratch, 4 * sizeof(double))) != hipSuccess ||      // [0..1] host-side all-reduces, [2..3] the direct exchange's barriers
     
      (e = hipMalloc((void**)&p->kx2, nx * sizeof(double))) != hipSuccess ||
      (e = hipMalloc((void**)&p->ky2, ny * sizeof(double))) != hipSuccess ||
      (e = hipMalloc((void**)&p->kz2, (p->nzc + 1) * sizeof(double))) != hipSuccess ||
      (e = hipMalloc((void**)&p->ztab, 2 * nz * sizeof(double))) != hipSuccess)
    return cleanup(fail(2, std::string("hipMalloc workspace: ") + hipGetErrorString(e)));
  for (auto& ev : p->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return cleanup(fail(2, std::string("hipEventCreate: ") + hipGetErrorString(e)));
  if (!generic) {  // function attributes (dynamic LDS above 64 KB) are set here, never inside a graph capture
    GenParams gp0; memset(&gp0, 0, sizeof(gp0)); gp0.nx = nx; gp0.ny = ny; gp0.nz = nz;
    const long long nzc = p->nzc, nzl = p->nzl;
    const ColGeom gx{(long long)ny * nzl, 0, (long long)ny * nzl}, gy{nzl, (long long)ny * nzl, nzl};
    FastGenParams fp0; memset(&fp0, 0, sizeof(fp0)); fp0.nx = nx; fp0.ny = ny; fp0.nz = nz;
    if ((e = launch_col_gen(dtype, nx, p->W, gx, (long long)ny * nzl, gp0, nullptr, 0, (int)nzl, p->tw_x, p->stream, true)) != hipSuccess ||
        (e = launch_col_fastgen(dtype, nx, p->W, gx, (long long)ny * nzl, fp0, 0, (int)nzl, p->tw_x, p->stream, true)) != hipSuccess ||
        (e = launch_col_plain(dtype, ny, +1, p->W, gy, (long long)nx * nzl, p->tw_y, p->stream, true)) != hipSuccess ||
        (e = launch_col_xpose(dtype, ny, p->W, gy, p->W, gy, (long long)nx * nzl, p->tw_y, p->stream, true)) != hipSuccess ||
        (e = launch_col_direct(dtype, ny, p->W, gy, nullptr, 0, (long long)nx * nzl, p->tw_y, p->stream, true)) != hipSuccess ||
        (e = launch_col_plain_acc(dtype, ny, p->W, gy, (long long)nx * nzl, 0, (int)nzl, nullptr, p->tw_y, p->stream, true)) != hipSuccess ||
        (e = launch_row_c2r_lognormal(dtype, (int)nzc, p->W, (long long)p->nxl * ny, 1.0, nullptr, nullptr, p->tw_z, p->partials, p->stream, true)) != hipSuccess ||
        (yz_merged_supported(dtype, ny, (int)nzc) && (e = launch_yz_merged(dtype, ny, (int)nzc, p->W, 8, 1.0, p->tw_z, p->partials, p->W, ColGeom{p->nzl, (long long)ny * p->nzl, p->nzl}, 8, p->tw_y, p->stream, true)) != hipSuccess) ||
        (e = launch_row_c2r(dtype, (int)nzc, p->W, (long long)p->nxl * ny, 1.0, p->tw_z, p->partials, p->stream, true)) != hipSuccess ||
        (e = launch_row_c2r_zscale(dtype, (int)nzc, p->W, (long long)p->nxl * ny, 1.0, nullptr, p->tw_z, p->partials, p->stream, true)) != hipSuccess ||
        (e = launch_row_c2r_xgather(dtype, (int)nzc, p->W, p->W, (long long)p->nxl * ny, 1.0, 8, 8, ny, p->tw_z, p->partials, p->stream, true)) != hipSuccess ||
        (e = launch_row_r2c(dtype, (int)nzc, p->W, (long long)p->nxl * ny, p->tw_z, p->stream, true)) != hipSuccess ||
        (e = launch_col_plain(dtype, ny, -1, p->W, gy, (long long)nx * nzl, p->tw_y, p->stream, true)) != hipSuccess ||
        (e = launch_col_plain(dtype, nx, -1, p->W, gx, (long long)ny * nzl, p->tw_x, p->stream, true)) != hipSuccess ||
        ((e = launch_row_c2r_gather(dtype, (int)nzc, p->W, p->W, (long long)p->nxl * ny, 1.0, (int)nzl, (long long)p->nxl * ny * nzl,
                                                  p->tw_z, p->partials, p->stream, true)) != hipSuccess))
      return cleanup(fail(2, std::string("kernel preparation: ") + hipGetErrorString(e)));
  }
  *out = p;
  return 0;
}

// Unpacked c2c plan (transform.py:207-213,266-270): one buffer [nx][ny][nz] complex, transformed in place.
int rf_plan_create_c2c(rf_plan** out, int nx, int ny, int nz, int dtype, int device) {
  RF_REQUIRE(out != nullptr, "plan pointer is null");
  *out = nullptr;
  RF_REQUIRE(dtype == RF_F32 || dtype == RF_F64, "dtype must be RF_F32 or RF_F64");
  bool generic = !col_size_supported(nx) || !col_size_supported(ny) || !rowc_size_supported(nz);
  if (!generic) {
    const int tcx = col_tile_cols(dtype, nx), tcy = col_tile_cols(dtype, ny);
    generic = ((long long)ny * nz) % tcx || ((long long)nx * nz) % tcy;      // too few columns for a tile
  }
  GenericDims gd;
  if (generic && !generic_dims(nx, ny, nz, dtype == RF_F64, false, gd))
    return fail(1, "unsupported shape for a c2c plan: nx, ny, nz must be even, and every axis must fit one line of the LDS (8192 points complex64 / 4096 complex128) or split into two factors that do");
  RF_HIP(hipSetDevice(device));
  rf_plan* p = new rf_plan();
  p->generic = generic;
  if (generic) { p->gdims = gd; p->gax = gd.ax; p->gay = gd.ay; p->gaz = gd.az; }
  p->nx = nx; p->ny = ny; p->nz = nz; p->nzc = nz / 2; p->f64 = dtype; p->device = device;
  p->nranks = 1; p->rank = 0; p->csize = dtype ? 16 : 8; p->nxl = nx; p->nzl = p->nzc; p->kz0 = 0;
  p->unpacked = true;
  p->w_bytes = (size_t)nx * ny * nz * p->csize;
  p->k_bytes = 0;
  auto cleanup = [&](int rc) { rf_plan_destroy(p); return rc; };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking)) != hipSuccess)
    return cleanup(fail(2, std::string("hipStreamCreate: ") + hipGetErrorString(e)));
  p->stream = p->own_stream;
  if ((e = hipMalloc(&p->W, p->w_bytes)) != hipSuccess) return cleanup(fail(2, std::string("hipMalloc field buffer: ") + hipGetErrorString(e)));
  int rc = dtype ? upload_twiddles<double>(&p->tw_x, nx) : upload_twiddles<float>(&p->tw_x, nx);
  if (!rc) rc = dtype ? upload_twiddles<double>(&p->tw_y, ny) : upload_twiddles<float>(&p->tw_y, ny);
  if (!rc) rc = dtype ? upload_twiddles<double>(&p->tw_z, nz) : upload_twiddles<float>(&p->tw_z, nz);
  if (rc) return cleanup(rc);
  for (auto& ev : p->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return cleanup(fail(2, std::string("hipEventCreate: ") + hipGetErrorString(e)));
  const ColGeom gx{(long long)ny * nz, 0, (long long)ny * nz}, gy{nz, (long long)ny * nz, nz};
  if (!generic && (!col_plain_addressable(dtype, nx, gx) || !col_plain_addressable(dtype, ny, gy)))
    return cleanup(fail(1, "unsupported shape for a c2c plan: an axis shorter than 1024 with rows more than 4 GiB apart "
                           "(32-bit lane offsets); make that axis >= 1024 or the others smaller"));
  for (int dir = -1; dir <= 1 && !generic; dir += 2)
    if ((e = launch_row_c2c(dtype, nz, dir, p->W, (long long)nx * ny, 1.0, p->tw_z, p->stream, true)) != hipSuccess ||
        (e = launch_col_plain(dtype, ny, dir, p->W, gy, (long long)nx * nz, p->tw_y, p->stream, true)) != hipSuccess ||
        (e = launch_col_plain(dtype, nx, dir, p->W, gx, (long long)ny * nz, p->tw_x, p->stream, true)) != hipSuccess)
      return cleanup(fail(2, std::string("kernel preparation: ") + hipGetErrorString(e)));
  *out = p;
  return 0;
}

int rf_upload_c(rf_plan* p, const void* host) {
  RF_REQUIRE(p && host, "null argument");
  RF_REQUIRE(p->unpacked, "rf_upload_c is for plans made by rf_plan_create_c2c");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipMemcpyAsync(p->W, host, p->w_bytes, hipMemcpyHostToDevice, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

int rf_download_c(rf_plan* p, void* host) {
  RF_REQUIRE(p && host, "null argument");
  RF_REQUIRE(p->unpacked, "rf_download_c is for plans made by rf_plan_create_c2c");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipMemcpyAsync(host, p->W, p->w_bytes, hipMemcpyDeviceToHost, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

// direction = -1: forward, unnormalised (np.fft.fftn); +1: inverse with numpy's 1/(nx ny nz) (np.fft.ifftn)
int rf_execute_c2c(rf_plan* p, int direction) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(p->unpacked, "rf_execute_c2c is for plans made by rf_plan_create_c2c");
  RF_REQUIRE(direction == 1 || direction == -1, "direction must be +1 (inverse) or -1 (forward)");
  RF_HIP(hipSetDevice(p->device));
  const long long nz = p->nz;
  const ColGeom gx{(long long)p->ny * nz, 0, (long long)p->ny * nz}, gy{nz, (long long)p->ny * nz, nz};
  const double scale = direction > 0 ? 1.0 / ((double)p->nx * (double)p->ny * (double)p->nz) : 1.0;
  RF_HIP(hipEventRecord(p->ev[0], p->stream));
  if (p->generic) {
    if (generic_any_long(p) && !p->G) RF_HIP(hipMalloc(&p->G, p->w_bytes));        // scratch of the four-step form
    HipGenericOps ops{p, p->stream};
    if (int rc = generic_c2c_seq(ops, p->gdims, p->W, p->G, direction, scale)) return rc;
  } else {
    RF_HIP(launch_col_plain(p->f64, p->nx, direction, p->W, gx, (long long)p->ny * nz, p->tw_x, p->stream));
    RF_HIP(launch_col_plain(p->f64, p->ny, direction, p->W, gy, (long long)p->nx * nz, p->tw_y, p->stream));
    RF_HIP(launch_row_c2c(p->f64, (int)nz, direction, p->W, (long long)p->nx * p->ny, scale, p->tw_z, p->stream));
  }
  RF_HIP(hipEventRecord(p->ev[4], p->stream));
  p->timed = false;
  return 0;
}

int rf_plan_destroy(rf_plan* p) {
  if (!p) return 0;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  drop_graphs(p);
  if (p->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(p->comm);
  if (p->comm_stream) { (void)hipStreamSynchronize(p->comm_stream); (void)hipStreamDestroy(p->comm_stream); }
  for (auto& e : p->pev) if (e) (void)hipEventDestroy(e);
  for (void* m : p->ipc_open) (void)hipIpcCloseMemHandle(m);
  if (p->dl_stream) { (void)hipStreamSynchronize(p->dl_stream); (void)hipStreamDestroy(p->dl_stream); }
  for (auto& e : p->sink_ev) (void)hipEventDestroy(e);
  void* bufs[] = {p->peer_tab, p->W, p->R, p->W2, p->R2, p->K, p->P_base, p->G, p->G2, p->tw_x, p->tw_y, p->tw_z, p->kx2, p->ky2, p->kz2, p->xt, p->st, p->sl, p->bin,
                  p->X, p->lntab, p->ypart, p->noise, p->mt_scratch, p->mt_send, p->mt_recv, p->mt_sbase, p->mt_first, p->mt_pos, p->mt_npos_dev, p->mt_states, p->mt_counts, p->mt_offsets, p->mt_rowtab, p->mt_flags, p->br_tmp, p->fixbuf, p->partials, p->stats, p->seeds_dev, p->ztab, p->frec, p->coll_scratch};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (int i = 0; i < 2; ++i) {
    if (p->seeds_pin[i]) (void)hipHostFree(p->seeds_pin[i]);
    if (p->seeds_ev[i]) (void)hipEventDestroy(p->seeds_ev[i]);
  }
  for (auto& ev : p->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : p->slab_ev) (void)hipEventDestroy(ev);
  for (auto& ev : p->chunk_ev) (void)hipEventDestroy(ev);
  for (auto& ev : p->bev)
    if (ev) (void)hipEventDestroy(ev);
  if (p->aux_stream) (void)hipStreamDestroy(p->aux_stream);
  if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
  delete p;
  return 0;
}

int rf_plan_nbytes(rf_plan* p, size_t* nbytes) {
  RF_REQUIRE(p && nbytes, "null argument");
  *nbytes = p->w_bytes * (1 + (p->R ? 1 : 0) + (p->W2 ? 2 : 0) + (p->X ? 1 : 0)) + (p->K ? p->k_bytes : 0) + (p->P ? p->p_bytes : 0) + ((p->G ? 1 : 0) + (p->G2 ? 1 : 0)) * (p->unpacked ? p->w_bytes : p->k_bytes) +
            p->noise_cap * sizeof(double) + p->mt_scratch_bytes;      // + resident deviates and the replay's scratch runs
  return 0;
}

int rf_plan_set_flag(rf_plan* p, int flag, int value) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(flag == RF_FLAG_EXACT_GENERATION || flag == RF_FLAG_FORCE_SLAB_PATH || flag == RF_FLAG_REPLICATED_GENERATION ||
             flag == RF_FLAG_TRANSPOSED_INTERMEDIATE || flag == RF_FLAG_YZ_SLAB_PLANES || flag == RF_FLAG_EXCHANGE_CHUNKS, "unknown flag");
  RF_HIP(hipStreamSynchronize(p->stream));
  if (p->comm_stream) RF_HIP(hipStreamSynchronize(p->comm_stream));
  if (flag == RF_FLAG_EXCHANGE_CHUNKS) {         // sub-slabs of the exchange: 1 (or 0) = the whole kz slab at once
    const int C = value <= 1 ? 1 : value;
    RF_REQUIRE(!p->generic && !p->unpacked, "RF_FLAG_EXCHANGE_CHUNKS is for packed plans on the tiled kernels");
    RF_REQUIRE((C & (C - 1)) == 0 && p->nzl % C == 0, "the number of exchange chunks must be a power of two that divides nz / (2 ranks)");
    const long long nzc_ = p->nzl / C;
    RF_REQUIRE(C == 1 || (nzc_ % 2 == 0 && ((long long)p->ny * nzc_) % col_tile_cols(p->f64, p->nx) == 0 &&
                          ((long long)p->ny * nzc_) % col_gen_tile_cols(p->f64, p->nx) == 0 && ((long long)p->nx * nzc_) % col_tile_cols(p->f64, p->ny) == 0),
               "too many exchange chunks for this grid: a sub-slab must hold an even number of planes and whole tiles of the x and y passes");
    RF_REQUIRE(!p->direct || col_direct_supported(p->f64, p->ny, nzc_),
               "too many exchange chunks for the direct exchange: a y-pass tile would straddle two x planes");
    p->xchunks = C;
    drop_graphs(p);
    p->real_valid = false;                       // (the buffers' layout between the passes changes; nothing resident survives it)
    if (p->direct) return rebuild_peer_tab(p);
    return 0;
  }
  if (flag == RF_FLAG_YZ_SLAB_PLANES) {          // -1 automatic, 0 whole-grid passes, > 0 x planes per slab
    p->yz_slab = value;
    drop_graphs(p);
    return 0;
  }
  if (flag == RF_FLAG_TRANSPOSED_INTERMEDIATE) {
    p->xposed = value != 0;
    drop_graphs(p);
    if (!p->xposed && p->X) { RF_HIP(hipFree(p->X)); p->X = nullptr; }
    return 0;
  }
  if (flag == RF_FLAG_REPLICATED_GENERATION) {
    RF_REQUIRE(p->nranks > 1, "RF_FLAG_REPLICATED_GENERATION is for multi-rank plans");
    RF_REQUIRE(!value || col_replicate_supported(p->f64, p->nx, p->nranks),
               "replicated generation is not available for this shape (nx too small for the number of ranks, or float64 with nx = 2048)");
    p->replicate = value != 0;
    return 0;
  }
  if (flag == RF_FLAG_FORCE_SLAB_PATH) {
    RF_REQUIRE(p->nranks == 1, "RF_FLAG_FORCE_SLAB_PATH is for single-rank plans");
    RF_REQUIRE(!p->generic, "RF_FLAG_FORCE_SLAB_PATH needs power-of-two axes");
    if (value && !p->R) RF_HIP(hipMalloc(&p->R, p->w_bytes));
    p->force_slab = value != 0;
    drop_graphs(p);
    return 0;
  }
  p->exact_gen = value != 0;
  drop_graphs(p);
  return 0;
}

int rf_plan_set_stream(rf_plan* p, void* hip_stream) {
  RF_REQUIRE(p, "null plan");
  RF_HIP(hipStreamSynchronize(p->stream));
  p->stream = hip_stream ? (hipStream_t)hip_stream : p->own_stream;
  drop_graphs(p);
  return 0;
}

int rf_set_kgrid(rf_plan* p, const double* kx2, const double* ky2, const double* kz2) {
  RF_REQUIRE(p && kx2 && ky2 && kz2, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipMemcpyAsync(p->kx2, kx2, p->nx * sizeof(double), hipMemcpyHostToDevice, p->stream));
  RF_HIP(hipMemcpyAsync(p->ky2, ky2, p->ny * sizeof(double), hipMemcpyHostToDevice, p->stream));
  RF_HIP(hipMemcpyAsync(p->kz2, kz2, (p->nzc + 1) * sizeof(double), hipMemcpyHostToDevice, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  p->h_kx2.assign(kx2, kx2 + p->nx); p->h_ky2.assign(ky2, ky2 + p->ny); p->h_kz2.assign(kz2, kz2 + p->nzc + 1);
  p->have_kgrid = true;
  return build_fast(p);
}

int rf_set_power(rf_plan* p, const double* log10k, const double* sigma, int n) {
  RF_REQUIRE(p && log10k && sigma, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(n >= 2, "power table needs at least 2 rows");
  for (int i = 0; i + 1 < n; ++i) RF_REQUIRE(log10k[i + 1] > log10k[i], "log10k must be strictly increasing");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipStreamSynchronize(p->stream));
  SigmaTableHost t;
  build_sigma_table(log10k, sigma, n, t);
  for (void* b : {(void*)p->xt, (void*)p->st, (void*)p->sl, (void*)p->bin})
    if (b) RF_HIP(hipFree(b));
  p->xt = p->st = p->sl = nullptr; p->bin = nullptr; p->have_power = false;
  RF_HIP(hipMalloc((void**)&p->xt, n * sizeof(double)));
  RF_HIP(hipMalloc((void**)&p->st, n * sizeof(double)));
  RF_HIP(hipMalloc((void**)&p->sl, t.sl.size() * sizeof(double)));
  RF_HIP(hipMalloc((void**)&p->bin, t.bin.size() * sizeof(int)));
  RF_HIP(hipMemcpy(p->xt, t.xt.data(), n * sizeof(double), hipMemcpyHostToDevice));
  RF_HIP(hipMemcpy(p->st, t.st.data(), n * sizeof(double), hipMemcpyHostToDevice));
  RF_HIP(hipMemcpy(p->sl, t.sl.data(), t.sl.size() * sizeof(double), hipMemcpyHostToDevice));
  RF_HIP(hipMemcpy(p->bin, t.bin.data(), t.bin.size() * sizeof(int), hipMemcpyHostToDevice));
  p->nt = n; p->nbins = (int)t.bin.size(); p->x0 = t.x0; p->inv_dx = t.inv_dx;
  p->have_power = true;
  p->h_tab = t;
  return build_fast(p);
}

int rf_generate(rf_plan* p, uint64_t seed, int mode, const double* noise_host) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
  RF_REQUIRE(mode == RF_NOISE_NATIVE || mode == RF_NOISE_EXTERNAL || mode == RF_NOISE_RESIDENT, "invalid noise mode");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = ensure_k(p)) return rc;        // a kz-slab rank holds (and generates) its own planes + the Nyquist plane
  if (int rc = upload_noise(p, mode, noise_host)) return rc;
  RF_REQUIRE(mode != RF_NOISE_RESIDENT || p->noise_resident, "rf_generate needs float64 deviates: only float32 copies are resident");
  RF_HIP(launch_gen_kspace(p->f64, p->K, make_gen(p, seed, mode, false), p->stream));
  if (mode == RF_NOISE_EXTERNAL) RF_HIP(hipStreamSynchronize(p->stream));  // host noise buffer may be released by the caller
  p->k_valid = true;
  p->aux_valid = false;
  return 0;
}

int rf_execute_c2r(rf_plan* p) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->K && p->k_valid, "no k-space data: call rf_generate or rf_upload_k first");
  RF_REQUIRE(!(p->replicate && p->nranks > 1), "replicated-generation plans have no distributed k-space buffer");
  RF_HIP(hipSetDevice(p->device));
  p->timed = true;
  return queue_c2r(p, make_gen(p, 0, RF_NOISE_NATIVE, false), p->K);
}

}  // extern "C"
namespace rfc {
// multi-rank forward transform, x-slab half: z pass on the local rows, in place, then the rows cut into the P send blocks
// [nxl][ny][nzl] of R (block g = the kz planes of rank g) -- the reverse of what the gathering z pass reads
int queue_r2c_slab_rows(rf_plan* p, hipStream_t s) {
  const long long nrows = (long long)p->nxl * p->ny;
  RF_HIP(launch_row_r2c(p->f64, (int)p->nzc, p->W, nrows, p->tw_z, s));
  const size_t seg = (size_t)p->nzl * p->csize, blk = (size_t)nrows * seg;
  for (int g = 0; g < p->nranks; ++g)
    RF_HIP(hipMemcpy2DAsync((char*)p->R + g * blk, seg, (const char*)p->W + g * seg, (size_t)p->nzc * p->csize, seg, (size_t)nrows,
                            hipMemcpyDeviceToDevice, s));
  return 0;
}
// kz-slab half: forward y and x passes on [nx][ny][nzl] (the blocks as they arrived: x is the slowest axis), then the side array
int queue_r2c_slab_cols(rf_plan* p, hipStream_t s) {
  const long long nzl = p->nzl;
  const ColGeom gx{(long long)p->ny * nzl, 0, (long long)p->ny * nzl}, gy{nzl, (long long)p->ny * nzl, nzl};
  RF_HIP(launch_col_plain(p->f64, p->ny, -1, p->W, gy, (long long)p->nx * nzl, p->tw_y, s));
  RF_HIP(launch_col_plain(p->f64, p->nx, -1, p->W, gx, (long long)p->ny * nzl, p->tw_x, s));
  RF_HIP(launch_unpack_kspace(p->f64, p->W, p->K, p->nx, p->ny, (int)nzl, p->kz0, s));
  p->real_valid = false;      // the field buffer now holds packed k space
  p->stats_valid = false;
  p->k_valid = true;
  p->aux_valid = false;
  return 0;
}
}  // namespace rfc
extern "C" {

int rf_execute_r2c(rf_plan* p) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->real_valid && p->cur == p->W, "no real-space field on the device: call rf_upload_real (or a c2r) first");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = ensure_k(p)) return rc;
  if (p->nranks > 1) {
    // transform.py:278-301 on x-slab / kz-slab ranks: rows, the all-to-all in the other direction (block g of R -> rank g,
    // arriving as block h of W: the same grouped send / receive with the buffers swapped), columns
    RF_REQUIRE(!p->generic && !p->replicate, "the multi-rank forward transform runs on the tiled kernels of exchange-mode plans");
    RF_HIP(hipEventRecord(p->ev[0], p->stream));
    if (int rc = queue_r2c_slab_rows(p, p->stream)) return rc;
    if (int rc = queue_exchange_rccl(p, p->R, p->W, p->stream, -1, true)) return rc;
    if (int rc = queue_r2c_slab_cols(p, p->stream)) return rc;
    RF_HIP(hipEventRecord(p->ev[4], p->stream));
    p->timed = false;
    return 0;
  }
  const long long nzc = p->nzc;
  const ColGeom gx{(long long)p->ny * nzc, 0, (long long)p->ny * nzc}, gy{nzc, (long long)p->ny * nzc, nzc};
  RF_HIP(hipEventRecord(p->ev[0], p->stream));
  if (p->generic) {            // rows -> half spectrum in K, then the y and x forward passes (rf_generic.h generic_r2c_seq)
    if (generic_any_long(p)) { if (int rc = ensure_g(p)) return rc; if (int rc = ensure_g2(p)) return rc; }
    HipGenericOps ops{p, p->stream};
    if (int rc = generic_r2c_seq(ops, p->gdims, p->W, p->K, p->G, p->G2)) return rc;
    RF_HIP(hipEventRecord(p->ev[4], p->stream));
    p->timed = false;
    p->k_valid = true;          // the real field in W is untouched on this path
    p->aux_valid = false;
    return 0;
  }
  RF_HIP(launch_row_r2c(p->f64, (int)nzc, p->W, (long long)p->nx * p->ny, p->tw_z, p->stream));     // z, in place
  RF_HIP(launch_col_plain(p->f64, p->ny, -1, p->W, gy, (long long)p->nx * nzc, p->tw_y, p->stream));   // y forward
  RF_HIP(launch_col_plain(p->f64, p->nx, -1, p->W, gx, (long long)p->ny * nzc, p->tw_x, p->stream));   // x forward
  RF_HIP(launch_unpack_kspace(p->f64, p->W, p->K, p->nx, p->ny, (int)nzc, 0, p->stream));
  RF_HIP(hipEventRecord(p->ev[4], p->stream));
  p->timed = false;
  p->real_valid = false;      // the field buffer now holds packed k space
  p->stats_valid = false;
  p->k_valid = true;
  p->aux_valid = false;
  return 0;
}

int rf_realise(rf_plan* p, uint64_t seed, int mode, const double* noise_host) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
  RF_REQUIRE(mode == RF_NOISE_NATIVE || mode == RF_NOISE_EXTERNAL || mode == RF_NOISE_RESIDENT, "invalid noise mode");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = upload_noise(p, mode, noise_host)) return rc;
  p->timed = true;
  p->resident_fast = (mode == RF_NOISE_RESIDENT);
  int rc = queue_c2r(p, make_gen(p, seed, mode, false), nullptr);
  p->resident_fast = false;
  if (rc) return rc;
  if (mode == RF_NOISE_EXTERNAL) RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

// calculate_newtonian_potential (generate.py:333-343) WITHOUT a stored potential: the inverse transform of scale * delta(k) / k^2,
// with delta(k) regenerated inside the x pass exactly as rf_realise(seed, mode) generates it -- the native generator is keyed by
// (seed, cell), the replayed reference stream is still resident -- so generate_delta_field(save_potential=True) need not write
// 4.3 GB per 1024^3 "in case" and the later load + transform reads nothing.  Every cell is rounded as its stored copy and the
// scaled copy of that would be.  Needs the fast generation pass (rf_can_regenerate_potential); returns the real field in place of
// the current one, like rf_load_potential + rf_execute_c2r.
int rf_can_regenerate_potential(rf_plan* p, int mode) {
  if (!p || p->unpacked || p->generic || !p->have_fast || p->exact_gen || (p->replicate && p->nranks > 1)) return 0;
  if (mode == RF_NOISE_NATIVE) return 1;
  if (mode == RF_NOISE_RESIDENT) return (p->noise32_resident && !p->f64) ? 1 : 0;
  return 0;
}

int rf_realise_scaled_potential(rf_plan* p, uint64_t seed, int mode, double scale, const double* factor_z) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
  RF_REQUIRE(rf_can_regenerate_potential(p, mode), "this plan / noise mode stores its potential (rf_realise_potential): nothing to regenerate from");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = upload_noise(p, mode, nullptr)) return rc;
  p->timed = true;
  p->resident_fast = (mode == RF_NOISE_RESIDENT);
  p->emit_potential = true;
  p->emit_pscale = p->f64 ? scale : (double)(float)scale;
  // the light-cone factor per plane z (generate.py:344-347): in the z pass's own store where the plan runs the plain single-rank
  // passes, else by the sweep rf_scale_z would make -- the same two roundings either way
  // (decided from the plan's FLAG: the blocked intermediate X is allocated lazily inside queue_c2r, and its gathering z pass has no
  // per-z factor -- testing p->X here dropped the factor on the first call after RF_FLAG_TRANSPOSED_INTERMEDIATE was set)
  if (int rc = ensure_x(p)) return rc;
  const bool fuse_z = factor_z && p->nranks == 1 && !p->force_slab && !xpose_ok(p);
  if (factor_z) RF_HIP(hipMemcpyAsync(p->ztab, factor_z, (size_t)p->nz * sizeof(double), hipMemcpyHostToDevice, p->stream));
  p->zscale = fuse_z ? p->ztab : nullptr;
  int rc = queue_c2r(p, make_gen(p, seed, mode, false), nullptr);
  p->zscale = nullptr;
  p->emit_potential = false;
  p->resident_fast = false;
  if (rc) return rc;
  if (factor_z && !fuse_z) {
    RF_HIP(launch_affine_z(p->f64, p->cur, (long long)p->nxl * p->ny, p->nz, p->ztab, 0.0, p->stream));
    p->stats_valid = false;
  }
  if (factor_z) RF_HIP(hipStreamSynchronize(p->stream));     // (factor_z is the caller's memory)
  return 0;
}

}  // extern "C"
namespace rfc {
// generate_delta_field(save_potential=True) (generate.py:191-219): the field as rf_realise, plus delta(k) / k^2 in the
// plan's potential buffer.  With the native generator (or resident float32 deviates) the potential is a second store
// stream of the generation pass; every other case runs the unfused sequence generate -> save_potential -> c2r.
// whole = false stops after the y pass (the slab pipeline's forward half, rf_slab_forward_ex)
int potential_forward(rf_plan* p, uint64_t seed, int mode, const double* noise_host, bool whole) {
  RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
  RF_REQUIRE(mode == RF_NOISE_NATIVE || mode == RF_NOISE_EXTERNAL || mode == RF_NOISE_RESIDENT, "invalid noise mode");
  RF_REQUIRE(!(p->replicate && p->nranks > 1), "replicated-generation plans keep no k-space potential: clear RF_FLAG_REPLICATED_GENERATION");
  RF_HIP(hipSetDevice(p->device));
  // fused: delta(k) / k^2 is a second store stream of the generation pass -- native generator (float32 and float64 plans)
  // or resident float32 deviates (float32 plans)
  const bool fused = ((mode == RF_NOISE_NATIVE) || (mode == RF_NOISE_RESIDENT && p->noise32_resident && !p->f64)) && p->have_fast &&
                     !p->exact_gen && !p->generic;
  if (!fused) {
    if (int rc = rf_generate(p, seed, mode, noise_host)) return rc;
    if (int rc = rf_save_potential(p)) return rc;
    if (whole) return rf_execute_c2r(p);
    return queue_xy(p, make_gen(p, 0, RF_NOISE_NATIVE, false), p->K, p->W, p->stream, false);
  }
  if (int rc = ensure_p(p)) return rc;
  p->timed = whole;
  p->pot_target = p->P;
  p->resident_fast = (mode == RF_NOISE_RESIDENT);
  const GenParams gp = make_gen(p, seed, mode, false);
  const int rc = whole ? queue_c2r(p, gp, nullptr) : queue_xy(p, gp, nullptr, p->W, p->stream, false);
  p->pot_target = nullptr;
  p->resident_fast = false;
  return rc;
}
}  // namespace rfc
extern "C" {

int rf_realise_potential(rf_plan* p, uint64_t seed, int mode, const double* noise_host) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  return potential_forward(p, seed, mode, noise_host, true);
}

// issue the n realisations of a batch on the plan's stream (under stream capture)
static int batch_issue(rf_plan* p, int n) {
  for (int i = 0; i < n; ++i) {
    GenParams gp = make_gen(p, 0, RF_NOISE_NATIVE, true);
    gp.seed_dev = p->seeds_dev + i;
    if (int rc = queue_xyz(p, gp, nullptr, p->W, p->stream, p->stats + 2 * i, false)) return rc;
  }
  return 0;
}

// Capture the whole batch as ONE graph: realisation i reads seed[i] from device memory and leaves its
// (sum, sumsq) in stats[2i].  (Two-stream pipelining of consecutive realisations was measured on
// MI355X and gave nothing: an x-pass and a y-pass kernel do not co-execute profitably -- DESIGN.md.)
static int batch_prepare(rf_plan* p, int n) {
  RF_REQUIRE(n >= 1, "need at least one seed");
  RF_REQUIRE(p->nranks == 1, "graph-captured batches are single-GPU; loop rf_realise on multi-GPU plans");
  RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
  RF_HIP(hipSetDevice(p->device));
  if (p->seeds_cap < n || p->stats_cap < n) {
    // device arrays are baked into the captured graphs: grow them (generously) and start over
    RF_HIP(hipStreamSynchronize(p->stream));
    drop_graphs(p);
    const int cap = n > 64 ? n : 64;
    if (p->seeds_dev) RF_HIP(hipFree(p->seeds_dev));
    p->seeds_dev = nullptr;
    RF_HIP(hipMalloc((void**)&p->seeds_dev, cap * sizeof(uint64_t)));
    p->seeds_cap = cap;
    if (p->stats) RF_HIP(hipFree(p->stats));
    p->stats = nullptr;
    RF_HIP(hipMalloc((void**)&p->stats, 2 * (size_t)cap * sizeof(double)));
    p->stats_cap = cap;
  }
  if (p->graphs.count(n)) return 0;
  if (int rc = ensure_x(p)) return rc;
  const bool timed_save = p->timed;
  p->timed = false;
  RF_HIP(hipStreamSynchronize(p->stream));
  rf_plan::BatchGraph bg;
  RF_HIP(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
  const int rc = batch_issue(p, n);
  hipError_t e2 = hipStreamEndCapture(p->stream, &bg.graph);
  p->timed = timed_save;
  if (rc) return rc;
  RF_HIP(e2);
  RF_HIP(hipGraphInstantiate(&bg.exec, bg.graph, nullptr, nullptr, 0));
  p->graphs[n] = bg;
  return 0;
}

int rf_realise_batch_prepare(rf_plan* p, int n) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  if (p->nranks > 1 || p->force_slab || p->generic) return 0;     // slab / generic batches are not graph-captured
  return batch_prepare(p, n);
}

int rf_realise_batch(rf_plan* p, const uint64_t* seeds, int n, double* rms_out) {
  RF_REQUIRE(p && seeds, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(n >= 1, "need at least one seed");
  if (p->nranks > 1 || p->force_slab) {
    RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
    RF_HIP(hipSetDevice(p->device));
    if (int rc = slab_batch(p, seeds, n)) return rc;
    if (rms_out) {
      std::vector<double> st(2 * (size_t)n);
      RF_HIP(hipMemcpyAsync(st.data(), p->stats, st.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
      RF_HIP(hipStreamSynchronize(p->stream));
      const double cnt = (double)p->nx * p->ny * p->nz;
      for (int i = 0; i < n; ++i) {
        const double m = st[2 * i] / cnt, v = st[2 * i + 1] / cnt - m * m;
        rms_out[i] = v > 0 ? std::sqrt(v) : 0.0;
      }
    }
    return 0;
  }
  if (p->generic) {             // no fused generation, no graph: realisations one after the other
    RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
    RF_HIP(hipSetDevice(p->device));
    if (p->stats_cap < n) {
      RF_HIP(hipStreamSynchronize(p->stream));
      if (p->stats) RF_HIP(hipFree(p->stats));
      p->stats = nullptr;
      RF_HIP(hipMalloc((void**)&p->stats, 2 * (size_t)(n + 64) * sizeof(double)));
      p->stats_cap = n + 64;
    }
    if (int rc = ensure_k(p)) return rc;
    RF_HIP(hipEventRecord(p->ev[0], p->stream));
    for (int i = 0; i < n; ++i) {
      RF_HIP(launch_gen_kspace(p->f64, p->K, make_gen(p, seeds[i], RF_NOISE_NATIVE, false), p->stream));
      if (int rc = generic_c2r(p, p->K, p->stats + 2 * i)) return rc;
    }
    RF_HIP(hipEventRecord(p->ev[4], p->stream));
    p->cur = p->W; p->timed = false; p->real_valid = true; p->stats_valid = true; p->k_valid = true; p->aux_valid = false;
    p->stats_slot = n - 1;
    if (rms_out) {
      std::vector<double> st(2 * (size_t)n);
      RF_HIP(hipMemcpyAsync(st.data(), p->stats, st.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
      RF_HIP(hipStreamSynchronize(p->stream));
      const double cnt = (double)p->nx * p->ny * p->nz;
      for (int i = 0; i < n; ++i) {
        const double m = st[2 * i] / cnt, v = st[2 * i + 1] / cnt - m * m;
        rms_out[i] = v > 0 ? std::sqrt(v) : 0.0;
      }
    }
    return 0;
  }
  if (int rc = batch_prepare(p, n)) return rc;
  if (p->seeds_pin_cap < n) {
    RF_HIP(hipStreamSynchronize(p->stream));
    for (int i = 0; i < 2; ++i) {
      if (p->seeds_pin[i]) RF_HIP(hipHostFree(p->seeds_pin[i]));
      p->seeds_pin[i] = nullptr;
      RF_HIP(hipHostMalloc((void**)&p->seeds_pin[i], (size_t)(n + 64) * sizeof(uint64_t), hipHostMallocDefault));
      if (!p->seeds_ev[i]) RF_HIP(hipEventCreateWithFlags(&p->seeds_ev[i], hipEventDisableTiming));
    }
    p->seeds_pin_cap = n + 64;
  }
  const int slot = p->seeds_turn;
  p->seeds_turn ^= 1;
  RF_HIP(hipEventSynchronize(p->seeds_ev[slot]));        // returns at once for an event that was never recorded
  std::memcpy(p->seeds_pin[slot], seeds, n * sizeof(uint64_t));
  RF_HIP(hipMemcpyAsync(p->seeds_dev, p->seeds_pin[slot], n * sizeof(uint64_t), hipMemcpyHostToDevice, p->stream));
  RF_HIP(hipEventRecord(p->seeds_ev[slot], p->stream));
  RF_HIP(hipEventRecord(p->ev[0], p->stream));
  RF_HIP(hipGraphLaunch(p->graphs[n].exec, p->stream));
  RF_HIP(hipEventRecord(p->ev[4], p->stream));
  p->cur = p->W;
  p->timed = false;
  p->real_valid = true;
  p->stats_valid = true;
  p->stats_slot = n - 1;
  if (rms_out) {
    std::vector<double> st(2 * (size_t)n);
    RF_HIP(hipMemcpyAsync(st.data(), p->stats, st.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    RF_HIP(hipStreamSynchronize(p->stream));
    const double cnt = (double)p->nx * p->ny * p->nz;
    for (int i = 0; i < n; ++i) {
      const double m = st[2 * i] / cnt;
      const double v = st[2 * i + 1] / cnt - m * m;
      rms_out[i] = v > 0 ? std::sqrt(v) : 0.0;
    }
  }
  return 0;
}

int rf_moments(rf_plan* p, double* mean, double* std_out) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->stats_valid, "no realisation has been computed");
  RF_HIP(hipSetDevice(p->device));
  double st[2];
  RF_HIP(hipMemcpyAsync(st, p->stats + 2 * p->stats_slot, sizeof(st), hipMemcpyDeviceToHost, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  const double cnt = (double)p->nx * p->ny * p->nz;
  const double m = st[0] / cnt;
  const double v = st[1] / cnt - m * m;
  if (mean) *mean = m;
  if (std_out) *std_out = v > 0 ? std::sqrt(v) : 0.0;
  return 0;
}

int rf_lognormal(rf_plan* p, const double* a_z, const double* b_z, int nz, double sigma) {
  RF_REQUIRE(p && a_z && b_z, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(nz == p->nz, "table length must equal nz");
  RF_REQUIRE(p->real_valid, "no real-space field on the device");
  RF_REQUIRE(sigma > 0, "sigma must be positive");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipMemcpyAsync(p->ztab, a_z, nz * sizeof(double), hipMemcpyHostToDevice, p->stream));
  RF_HIP(hipMemcpyAsync(p->ztab + nz, b_z, nz * sizeof(double), hipMemcpyHostToDevice, p->stream));
  RF_HIP(launch_lognormal(p->f64, p->cur, (long long)p->nxl * p->ny, nz, p->ztab, p->ztab + nz, sigma, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));  // host tables may go away
  p->stats_valid = false;
  return 0;
}

int rf_set_z_tables(rf_plan* p, const double* growth_z, const double* density_z, int nz) {
  RF_REQUIRE(p && growth_z, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(nz == p->nz, "table length must equal nz");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipStreamSynchronize(p->stream));
  if (!p->lntab) RF_HIP(hipMalloc((void**)&p->lntab, (4 * (size_t)nz + 8) * sizeof(double)));
  RF_HIP(hipMemcpy(p->lntab, growth_z, nz * sizeof(double), hipMemcpyHostToDevice));
  if (density_z) RF_HIP(hipMemcpy(p->lntab + nz, density_z, nz * sizeof(double), hipMemcpyHostToDevice));
  p->ln_tables = true;
  p->ln_density = density_z != nullptr;
  return 0;
}

// rows K,T,R,S + the c2r transform + the lognormal map, with sigma = the field's rms taken from the y pass (Parseval) so that the
// map runs in the z pass's epilogue: generate_delta_field(save_potential=False) followed by convert_delta_to_density()
// (generate.py:191-199,218-219 and 266-273, cosmotools.py:206-221) in 5 sweeps instead of 7 and without the host round trip.
int rf_realise_lognormal(rf_plan* p, uint64_t seed, int mode, const double* noise_host, double* sigma_out) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->have_kgrid && p->have_power, "rf_set_kgrid and rf_set_power must be called first");
  RF_REQUIRE(p->ln_tables, "rf_set_z_tables must be called first");
  RF_REQUIRE(p->nranks == 1 && !p->force_slab && !p->generic, "rf_realise_lognormal is for single-GPU plans with power-of-two axes");
  RF_REQUIRE(mode == RF_NOISE_NATIVE || mode == RF_NOISE_EXTERNAL || mode == RF_NOISE_RESIDENT, "unknown noise mode");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = upload_noise(p, mode, noise_host)) return rc;
  p->resident_fast = (mode == RF_NOISE_RESIDENT);
  const long long nzl = p->nzl, ntiles = (long long)p->nx * nzl / col_tile_cols(p->f64, p->ny);
  if (p->nypart < ntiles) {
    if (p->ypart) RF_HIP(hipFree(p->ypart));
    p->ypart = nullptr; p->nypart = 0;
    RF_HIP(hipMalloc((void**)&p->ypart, ntiles * sizeof(double)));
    p->nypart = ntiles;
  }
  hipStream_t s = p->stream;
  const GenParams gp = make_gen(p, seed, mode, false);
  const ColGeom gy{nzl, (long long)p->ny * nzl, nzl};
  const double n3 = (double)p->nx * (double)p->ny * (double)p->nz, scale = 1.0 / n3;
  double *growth = p->lntab, *dens = p->lntab + p->nz, *A = p->lntab + 2 * p->nz, *B = p->lntab + 3 * p->nz, *sig = p->lntab + 4 * p->nz;
  RF_HIP(hipEventRecord(p->ev[0], s));
  void* Xsave = p->X;
  p->X = nullptr;                                    // plain layout: the accumulating y pass runs in place on W
  p->slab_timed = 0;
  p->slab_merged = 0;
  int rc = queue_x(p, gp, nullptr, p->W, s, true);   // (timed: rf_kernel_ms reports x, y + tables, z + map, reduce of this call too)
  p->X = Xsave;
  p->resident_fast = false;
  if (rc) return rc;
  RF_HIP(hipEventRecord(p->ev[1], s));
  RF_HIP(launch_col_plain_acc(p->f64, p->ny, p->W, gy, (long long)p->nx * nzl, p->kz0, (int)nzl, p->ypart, p->tw_y, s));
  // rms = sqrt(S / (nx ny)) / N3  (rf_fft.h AccColIO)
  RF_HIP(launch_lognormal_tables(p->ypart, ntiles, 1.0 / ((double)p->nx * (double)p->ny * n3 * n3), growth, p->ln_density ? dens : nullptr, p->nz,
                                 p->f64 ? 0 : 1, p->f64 ? lognormal_ap_unit<double>(scale) : 1.0, sig, A, B, s));
  RF_HIP(hipEventRecord(p->ev[2], s));
  RF_HIP(launch_row_c2r_lognormal(p->f64, (int)p->nzc, p->W, (long long)p->nx * p->ny, scale, A, B, p->tw_z, p->partials, s));
  RF_HIP(hipEventRecord(p->ev[3], s));
  RF_HIP(launch_reduce_partials(p->partials, p->npartials, p->stats, p->partials + 2 * p->npartials, s));
  RF_HIP(hipEventRecord(p->ev[4], s));
  p->timed = true;
  p->cur = p->W;
  p->stats_slot = 0;
  p->real_valid = true;
  p->stats_valid = true;                             // (the moments of the DENSITY field now)
  p->k_valid = false;
  if (sigma_out) {
    RF_HIP(hipMemcpyAsync(sigma_out, sig, sizeof(double), hipMemcpyDeviceToHost, s));
    RF_HIP(hipStreamSynchronize(s));
  }
  return 0;
}

int rf_affine_z(rf_plan* p, const double* mul_z, int nz, double add) {
  RF_REQUIRE(p && mul_z, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(nz == p->nz, "table length must equal nz");
  RF_REQUIRE(p->real_valid, "no real-space field on the device");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipMemcpyAsync(p->ztab, mul_z, nz * sizeof(double), hipMemcpyHostToDevice, p->stream));
  RF_HIP(launch_affine_z(p->f64, p->cur, (long long)p->nxl * p->ny, nz, p->ztab, add, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  p->stats_valid = false;
  return 0;
}

int rf_scale_z(rf_plan* p, const double* factor_z, int nz) { return rf_affine_z(p, factor_z, nz, 0.0); }

// Lensing potential psi of the real field on the device (the Newtonian potential on the light cone) into the
// auxiliary buffer: generate.py:352-416 with cot_z = cotK(D) (generate.py:383-395) and D = spacing * iz.
int rf_lensing_potential(rf_plan* p, const double* cot_z, int nz, double spacing, int i_min) {
  RF_REQUIRE(p && cot_z, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(nz == p->nz, "table length must equal nz");
  RF_REQUIRE(i_min >= 0 && i_min < nz, "invalid i_min");
  RF_REQUIRE(spacing > 0, "spacing must be positive");
  RF_REQUIRE(p->real_valid, "no real-space field on the device");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = ensure_k(p)) return rc;            // (nx ny (nz/2+1)) complex >= (nx ny nz) real
  RF_HIP(hipMemcpyAsync(p->ztab, cot_z, nz * sizeof(double), hipMemcpyHostToDevice, p->stream));
  RF_HIP(launch_lensing(p->f64, p->cur, p->K, (long long)p->nxl * p->ny, nz, p->ztab, spacing, i_min, p->stream));   // rows are local: x slab
  RF_HIP(hipStreamSynchronize(p->stream));
  p->k_valid = false;
  p->aux_valid = true;
  return 0;
}

// planes [x0, x1) of the auxiliary real field (dense [nx][ny][nz])
int rf_download_aux(rf_plan* p, void* host, int x0, int x1) {
  RF_REQUIRE(p && host, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->K && p->aux_valid, "no auxiliary field on the device");
  RF_REQUIRE(0 <= x0 && x0 < x1 && x1 <= p->nxl, "invalid x range (multi-GPU plans hold nx/ranks local planes)");
  RF_HIP(hipSetDevice(p->device));
  const size_t plane = (size_t)p->ny * p->nz * (p->csize / 2);
  RF_HIP(hipMemcpyAsync(host, (const char*)p->K + (size_t)x0 * plane, (size_t)(x1 - x0) * plane, hipMemcpyDeviceToHost, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

int rf_save_potential(rf_plan* p) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->K && p->k_valid, "no k-space data");
  RF_REQUIRE(p->have_kgrid, "rf_set_kgrid must be called first");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = ensure_p(p)) return rc;
  RF_HIP(launch_save_potential(p->f64, p->K, p->P, p->nx, p->ny, p->nz, p->kx2, p->ky2, p->kz2, p->nzl + 1, p->kz0, p->ppitch, p->stream));
  return 0;
}

int rf_load_potential(rf_plan* p, double scale) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->P, "no saved potential");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = ensure_k(p)) return rc;
  RF_HIP(launch_scale_copy(p->f64, p->P, p->K, (long long)p->nx * p->ny * (p->nzl + 1), (int)p->nzl + 1, p->ppitch, scale, p->stream));
  p->k_valid = true;
  p->aux_valid = false;
  return 0;
}

int rf_upload_k(rf_plan* p, const void* host) {
  RF_REQUIRE(p && host, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_HIP(hipSetDevice(p->device));
  if (int rc = ensure_k(p)) return rc;
  RF_HIP(hipMemcpyAsync(p->K, host, p->k_bytes, hipMemcpyHostToDevice, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  p->k_valid = true;
  p->aux_valid = false;
  return 0;
}

int rf_download_k(rf_plan* p, void* host) {
  RF_REQUIRE(p && host, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->K && p->k_valid, "no k-space data");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipMemcpyAsync(host, p->K, p->k_bytes, hipMemcpyDeviceToHost, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

int rf_upload_real(rf_plan* p, const void* host, int layout) {
  RF_REQUIRE(p && host, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(!p->replicate || p->nranks == 1, "replicated-generation plans keep no exchange buffers");
  RF_HIP(hipSetDevice(p->device));
  const size_t rsize = p->csize / 2;
  const size_t width = (size_t)p->nz * rsize;
  const size_t hpitch = layout == RF_LAYOUT_PADDED ? (size_t)(p->nz + 2) * rsize : width;
  // (a multi-rank plan takes its own nx / ranks planes)
  RF_HIP(hipMemcpy2DAsync(p->W, width, host, hpitch, width, (size_t)p->nxl * p->ny, hipMemcpyHostToDevice, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  p->cur = p->W;
  p->real_valid = true;
  p->stats_valid = false;
  return 0;
}

int rf_download_real(rf_plan* p, void* host, int layout, int x0, int x1) {
  RF_REQUIRE(p && host, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->real_valid, "no real-space field on the device");
  RF_REQUIRE(0 <= x0 && x0 < x1 && x1 <= p->nxl, "invalid x range (multi-GPU plans hold nx/ranks local planes)");
  RF_HIP(hipSetDevice(p->device));
  const size_t rsize = p->csize / 2;
  const size_t width = (size_t)p->nz * rsize;
  const size_t hpitch = layout == RF_LAYOUT_PADDED ? (size_t)(p->nz + 2) * rsize : width;
  const char* src = (const char*)p->cur + (size_t)x0 * p->ny * width;
  RF_HIP(hipMemcpy2DAsync(host, hpitch, src, width, width, (size_t)(x1 - x0) * p->ny, hipMemcpyDeviceToHost, p->stream));
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

// generate.py:184-189,230: the reference's calls RETURN a host array, and at 1024^3 the device -> host copy (75 ms over PCIe) is 15 x the
// realisation.  Armed with a host buffer, the NEXT realisation of a single-GPU plan (rf_realise, rf_realise_potential,
// rf_realise_batch_reference ...: everything whose y / z passes run slab by slab on the plan's stream) copies every slab of x planes to
// the host as soon as its z pass has finished, while the GPU runs the following slabs, and returns when the whole field is in `host`
// (layout as rf_download_real): the copy starts ~1.5 ms into a 1024^3 realisation instead of after it.  Ordinary pageable memory.
// One shot: rf_host_sink_delivered says whether the armed call delivered (then rf_download_real is not needed) and disarms.
// host = NULL disarms.
int rf_set_host_sink(rf_plan* p, void* host, int layout) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(!p->unpacked && !p->generic && p->nranks == 1 && !p->force_slab, "a host sink serves single-GPU packed plans on the tiled kernels");
  RF_REQUIRE(layout == RF_LAYOUT_DENSE || layout == RF_LAYOUT_PADDED, "invalid layout");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipStreamSynchronize(p->stream));
  p->sink_delivered = false;
  if (p->f64 && (p->sink_host == nullptr) != (host == nullptr)) drop_graphs(p);      // (a float64 plan runs its y / z passes slab by slab only for a sink)
  p->sink_host = host;
  p->sink_layout = layout;
  if (host && !p->dl_stream) RF_HIP(hipStreamCreateWithFlags(&p->dl_stream, hipStreamNonBlocking));
  return 0;
}

int rf_host_sink_delivered(rf_plan* p, int* delivered) {
  RF_REQUIRE(p && delivered, "null argument");
  *delivered = p->sink_delivered ? 1 : 0;
  if (p->sink_host && p->f64) { RF_HIP(hipStreamSynchronize(p->stream)); drop_graphs(p); }
  p->sink_host = nullptr;
  p->sink_delivered = false;
  return 0;
}

int rf_device_ptr(rf_plan* p, void** real_field, void** kspace) {
  RF_REQUIRE(p, "null plan");
  if (real_field) *real_field = p->cur ? p->cur : p->W;
  if (kspace) *kspace = p->K;
  return 0;
}

int rf_sync(rf_plan* p) {
  RF_REQUIRE(p, "null plan");
  RF_HIP(hipSetDevice(p->device));
  RF_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

int rf_elapsed_ms(rf_plan* p, float* ms) {
  RF_REQUIRE(p && ms, "null argument");
  RF_HIP(hipEventSynchronize(p->ev[4]));
  RF_HIP(hipEventElapsedTime(ms, p->ev[0], p->ev[4]));
  return 0;
}

int rf_kernel_ms(rf_plan* p, float* ms5) {
  RF_REQUIRE(p && ms5, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  RF_REQUIRE(p->timed, "per-kernel times are recorded by rf_realise / rf_execute_c2r only");
  RF_HIP(hipEventSynchronize(p->ev[4]));
  for (int i = 0; i < 4; ++i) RF_HIP(hipEventElapsedTime(&ms5[i], p->ev[i], p->ev[i + 1]));
  if (p->slab_merged == 0 && p->slab_timed > 0) {   // y / z passes ran slab by slab, one launch per pass: [1], [2] = the sums over their launches
    float ys = 0, zs = 0, t = 0;
    for (int i = 0; i < p->slab_timed; ++i) {
      RF_HIP(hipEventElapsedTime(&t, i == 0 ? p->ev[1] : p->slab_ev[2 * i - 1], p->slab_ev[2 * i]));
      ys += t;
      RF_HIP(hipEventElapsedTime(&t, p->slab_ev[2 * i], p->slab_ev[2 * i + 1]));
      zs += t;
    }
    ms5[1] = ys;
    ms5[2] = zs;
    RF_HIP(hipEventElapsedTime(&ms5[3], p->slab_ev[2 * p->slab_timed - 1], p->ev[4]));
  }
  if (p->slab_merged > 0) {                // merged launches (rf_set_merged_yz(2)): [1] = the first y launch + every merged launch, [2] = the last z launch
    float t = 0, ys = 0;
    RF_HIP(hipEventElapsedTime(&ys, p->ev[1], p->slab_ev[0]));
    for (int i = 1; i < p->slab_merged; ++i) {
      RF_HIP(hipEventElapsedTime(&t, p->slab_ev[i - 1], p->slab_ev[i]));
      ys += t;
    }
    ms5[1] = ys;
    RF_HIP(hipEventElapsedTime(&ms5[2], p->slab_ev[p->slab_merged - 1], p->slab_ev[p->slab_merged]));
    RF_HIP(hipEventElapsedTime(&ms5[3], p->slab_ev[p->slab_merged], p->ev[4]));
  }
  // the x pass of the fast generation is two launches: the few tiles that hold slot kz = 0 (with the Hermitian
  // repair), then all the others; report them separately so that [0] is the main kernel alone
  ms5[4] = 0.0f;
  if (p->repair_timed) {
    RF_HIP(hipEventElapsedTime(&ms5[4], p->ev[0], p->ev[5]));
    RF_HIP(hipEventElapsedTime(&ms5[0], p->ev[5], p->ev[1]));
  }
  return 0;
}

// 0: one launch per pass and slab always; 1 (default): untimed calls (graph-captured batches, rf_realise_batch_reference) put the z pass
// of slab s and the y pass of slab s + 1 into one launch where rf_k_yz.hip serves the shape; 2: timed calls (rf_realise ...) too
int rf_set_merged_yz(rf_plan* p, int mode) {
  RF_REQUIRE(p, "null plan");
  RF_REQUIRE(mode >= 0 && mode <= 2, "mode is 0, 1 or 2");
  if (p->yz_merge != mode) {
    RF_HIP(hipSetDevice(p->device));
    RF_HIP(hipStreamSynchronize(p->stream));
    drop_graphs(p);                          // (captured batches carry their launches)
    p->yz_merge = mode;
  }
  return 0;
}

// the merged launches of the last timed call under rf_set_merged_yz(2): their summed duration (HIP events on the plan's stream) and number
int rf_merged_yz_ms(rf_plan* p, float* sum_ms, int* launches) {
  RF_REQUIRE(p && sum_ms && launches, "null argument");
  RF_REQUIRE(p->timed && p->slab_merged > 1, "the last call was not a timed call with merged y / z launches (rf_set_merged_yz(2), a shape rf_k_yz.hip serves)");
  RF_HIP(hipEventSynchronize(p->ev[4]));
  float t = 0, sum = 0;
  for (int i = 1; i < p->slab_merged; ++i) {
    RF_HIP(hipEventElapsedTime(&t, p->slab_ev[i - 1], p->slab_ev[i]));
    sum += t;
  }
  *sum_ms = sum;
  *launches = p->slab_merged - 1;
  return 0;
}

int rf_yz_slabs(rf_plan* p, int* nslab, int* planes) {
  RF_REQUIRE(p && nslab && planes, "null argument");
  RF_REQUIRE(!p->unpacked, "this call does not apply to an unpacked c2c plan");
  long long B = yz_slab_planes(p);
  if (p->X && xpose_ok(p) && B > 0 && (B % xpose_row_block(p) || (B & (B - 1)) || p->nx % B)) B = 0;
  *planes = B > 0 ? (int)B : p->nxl;
  *nslab = B > 0 ? (int)((p->nx + B - 1) / B) : 1;
  return 0;
}

}  // extern "C"
