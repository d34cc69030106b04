#!/usr/bin/env python3
"""
bench.py -- Mcells/s for N^3 delta(x) realisations on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--edge 1024] [--no-cpu-baseline]

One "step" = one realisation seed -> delta(x) resident in HBM + rms
(generate_delta_field(save_potential=False) semantics: rows K,T,R,S fused into
the c2r FFT, then the moments).  Native counter-based RNG, float32, the shipped
500-row P(k) table.  The K timed steps are replayed from one captured hipGraph
(BASELINE config 3) and bracketed by barrier + device sync; rank 0 prints ONE
JSON line.  For N > 1 the driver launches this under torch.distributed.run, one
rank per GPU.

Extra objects on the JSON line:
  roofline     -- dominant kernel: algorithmic bytes per launch / its mean duration
                  (HIP events on the plan's stream, eager launches inside this run)
  cpu_baseline -- the oracle (numpy restatement of the reference path) timed on
                  this host, 1 thread, on the SAME grid (512^3 only if 1024^3 does
                  not fit the host), rank 0, N = 1 only
  other_configs -- the other BASELINE configurations / API paths, timed in this run
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def grid_for(ngpus, n):
    """Weak scaling: per-GPU work fixed at n^3 cells (1 -> n^3 ... 8 -> (2n)^3).  The z axis is doubled first, then y,
    then x: the strided x / y passes of length 2048 need the whole LDS of a CU for one tile and run ~1.3x slower
    per cell than at 1024, so the smaller jobs keep them at 1024."""
    shape = [n, n, n]
    axis, g = 2, ngpus
    while g > 1:
        shape[axis] *= 2
        axis = (axis - 1) % 3
        g //= 2
    return tuple(shape)


def cpu_baseline(power, spacing, sample_n):
    """The oracle (numpy restatement of the reference path, one thread) on the workload's own grid; only if that does
    not fit the host's memory, on a 512^3 sample of it (stated in `sample`)."""
    from oracle import cpu_ref                     # checker / baseline only
    note = ""
    while True:
        try:
            t0 = time.perf_counter()
            delta, rms = cpu_ref.generate_delta_field(sample_n, sample_n, sample_n, spacing, power["k"], power["Pk"], seed=123)
            dt = time.perf_counter() - t0
            break
        except MemoryError:
            if sample_n <= 512:
                raise
            note = " (MemoryError at %d^3 on this host: 512^3 fallback)" % sample_n
            sample_n = 512
    del delta
    return {"value": round(sample_n ** 3 / dt / 1e6, 3), "unit": "Mcells/s", "cores": 1, "kind": "port",
            "sample": "%d^3 float32 realisation, default P(k), seed 123, numpy %s (%.1f s; os.cpu_count()=%d)%s"
                      % (sample_n, np.__version__, dt, os.cpu_count() or 0, note),
            "rms": float(rms)}


def _timed(fn, sync, reps=3, warm=1):
    """median wall time of fn() bracketed by device syncs, after `warm` warm-up calls (a freshly allocated multi-GB
    buffer -- the saved potential -- is still being mapped during its first few sweeps)"""
    for _ in range(warm):
        fn()
    sync()
    ts = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def other_configs(power, spacing, device, only=None):
    """The other BASELINE.json configurations and API paths on one GPU (wall time around each call, inputs resident,
    eager launches): they are parity-test cases, not the bench line, but their rates belong beside it.  Every entry carries
    `kernel_ms` (HIP-event intervals around the passes of one more call of the same kind, launch gaps included) and a
    `roofline` object for its slowest pass: algorithmic bytes of that pass / its time."""
    from randomfield_amd import _hip, cosmotools, powertools, Generator
    out = {}

    def plan_for(n, dtype):
        plan = _hip.DevicePlan(n, n, n, dtype, device=device)
        plan.set_kgrid(*powertools.ksq_axes(n, n, n, spacing))
        plan.set_power(*powertools.sigma_table(power, (n, n, n), spacing))
        return plan

    def passes(dev, n, itemsize, x_sweeps=1.0, extra=None):
        """kernel_ms of the device plan's last timed call + the roofline object of its slowest pass.  x_sweeps: sweeps of the
        packed array the generation pass moves (1 = write only; 2 = it also reads deviates or writes the potential)."""
        ms = [float(v) for v in dev.kernel_ms()]
        sweep = itemsize * float(n) * n * (n // 2 + 1)
        names = ("x pass (generation + FFT)", "y pass (FFT in place)", "z pass (c2r + moments)")
        t = [ms[0] + ms[4], ms[1], ms[2]]
        byts = [x_sweeps * sweep, 2 * sweep, 2 * sweep]
        k = int(np.argmax(t))
        d = {"kernel_ms": {"x": round(ms[0], 4), "x_kz0_tiles": round(ms[4], 4), "y": round(ms[1], 4), "z": round(ms[2], 4), "reduce": round(ms[3], 4)},
             "roofline": {"bound": "hbm", "kernel": names[k], "ms": round(t[k], 4), "algorithmic_bytes": byts[k],
                          "achieved": round(byts[k] / (t[k] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(byts[k] / (t[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        if extra:
            d["kernel_ms"].update(extra)
        return d

    def entry(n, t, bytes_per_cell, **extra):
        d = {"ms": round(t * 1e3, 3), "Mcells_s": round(n ** 3 / t / 1e6, 1),
             "frac_of_hbm_peak": round(bytes_per_cell * n ** 3 / t / 1e9 / HBM_PEAK_GBS, 4)}
        d.update(extra)
        return d

    seeds = iter(range(5000, 6000))
    if only == "float64":
        # config 4's dtype on one GPU, and config 5: float64 + lognormal map (rows D and L: moments, then the per-z map)
        plan = plan_for(1024, np.complex128)
        t = _timed(lambda: plan.realise(seed=next(seeds)), plan.sync)
        out["1024^3 f64"] = entry(1024, t, 40 * (1 + 2 / 1024), **passes(plan, 1024, 16))
        growth = np.exp(-0.5 * np.arange(1024) / 1024)

        def f64_lognormal():
            plan.realise(seed=next(seeds))
            mean, std = plan.moments()
            a_z, b_z = cosmotools.lognormal_tables(growth, std, 1024)
            plan.lognormal(a_z, b_z, std)
        t_unfused = _timed(f64_lognormal, plan.sync)
        # the same configuration fused (rf_realise_lognormal): sigma from the y pass (Parseval), the map in the z pass's epilogue
        plan.set_z_tables(growth)
        t = _timed(lambda: plan.realise_lognormal(seed=next(seeds), want_sigma=False), plan.sync)
        out["1024^3 f64 + lognormal"] = entry(1024, t, 40 * (1 + 2 / 1024),
                                              note="fused (rf_realise_lognormal): 5 sweeps = 40 (1 + 2/nz) B/cell, sigma by Parseval from the y pass, "
                                                   "map in the z pass's epilogue",
                                              unfused=entry(1024, t_unfused, 56 * (1 + 2 / 1024),
                                                            note="rf_realise + rf_moments (host round trip) + rf_lognormal: 56 (1 + 2/nz) B/cell"),
                                              **passes(plan, 1024, 16))
        plan.close()
        return out
    # config 1: 512^3 float32, single realisation: one call, once through eager launches and once as a replayed one-realisation graph
    plan = plan_for(512, np.complex64)
    t = _timed(lambda: plan.realise(seed=next(seeds)), plan.sync, reps=5)
    kp = passes(plan, 512, 8)
    plan.realise_batch_prepare(1)
    tg = _timed(lambda: plan.realise_batch([next(seeds)], want_rms=False), plan.sync, reps=5)
    out["512^3 f32 single realisation"] = entry(512, min(t, tg), 20 * (1 + 2 / 512), ms_eager=round(t * 1e3, 3), ms_graph=round(tg * 1e3, 3), **kp)
    plan.close()
    # the same-seed path: numpy's MT19937 + polar stream replayed on the GPU (kept as float32 pairs, as Generator does for
    # complex64 plans), then the pipeline with the generation pass reading those deviates
    plan = plan_for(1024, np.complex64)
    state = {"rng": 0.0}

    def reference_rng():
        t0 = time.perf_counter()
        plan.reference_noise(next(seeds), single=True)
        plan.sync()
        state["rng"] = time.perf_counter() - t0
        plan.realise(noise="resident")
    t = _timed(reference_rng, plan.sync)
    out["1024^3 f32 rng='reference' (same field as the reference for the same seed)"] = entry(
        1024, t, 20 * (1 + 2 / 1024), ms_mt19937_replay=round(state["rng"] * 1e3, 3),
        **passes(plan, 1024, 8, x_sweeps=2.0, extra={"mt19937_replay (jump tree + polar pass + scan, wall)": round(state["rng"] * 1e3, 3)}))
    # ten of them back to back: the replay of seed i + 1 on a second stream under the y / z passes of seed i
    batch = [next(seeds) for _ in range(10)]
    plan.realise_batch_reference(batch[:2], want_rms=False)
    t = _timed(lambda: plan.realise_batch_reference(batch, want_rms=False), plan.sync, reps=3, warm=0) / len(batch)
    out["1024^3 f32 rng='reference', 10 back to back (rf_realise_batch_reference)"] = entry(
        1024, t, 20 * (1 + 2 / 1024), note="per realisation; includes the host-side seeding of the ten MT19937 states")
    plan.close()
    # the reference API's default call: Generator.generate_delta_field(save_potential=True), field kept on the device
    # (default: delta(k)/k**2 is not written but formed again inside calculate_newtonian_potential's generation pass -- from the
    # seed, or from the replayed deviates still on the device; store_potential=True writes it at generation time as the reference
    # does.  Both are timed, each also together with the call that uses the potential.)
    for rng in ("native", "reference"):
        for store in (False, True):
            gen = Generator(1024, 1024, 1024, spacing, power=power, rng=rng, store_potential=store)
            dev = gen.plan_c2r.device
            how = "potential stored at generation time (store_potential=True)" if store else "potential regenerated on demand (default)"
            key = "1024^3 f32 Generator.generate_delta_field(save_potential=True), rng='%s'%s" % (rng, ", store_potential=True" if store else "")
            t = _timed(lambda: gen.generate_delta_field(seed=next(seeds), save_potential=True, download=False), dev.sync, reps=5, warm=3)
            # (per-pass events: one eager call of what the Generator issued -- its native-generator call is a graph replay)
            noise_arg = "resident" if rng == "reference" else None
            (dev.realise_potential if store else dev.realise)(seed=next(seeds), noise=noise_arg)
            kp = passes(dev, 1024, 8, x_sweeps=(2.0 if rng == "reference" else 1.0) + (1.0 if store else 0.0))

            def both():
                gen.generate_delta_field(seed=next(seeds), save_potential=True, download=False)
                gen.calculate_newtonian_potential(light_cone=False, scale=-1.5, download=False)
            t2 = _timed(both, dev.sync, reps=3, warm=1)
            out[key] = entry(1024, t, (28 if store else 20) * (1 + 2 / 1024), note=how,
                             ms_with_calculate_newtonian_potential=round(t2 * 1e3, 3), **kp)
            if not store:
                t = _timed(lambda: gen.generate_delta_field(seed=next(seeds), save_potential=False, download=False), dev.sync)
                dev.realise(seed=next(seeds), noise=noise_arg)
                out["1024^3 f32 Generator.generate_delta_field(save_potential=False), rng='%s'" % rng] = entry(
                    1024, t, 20 * (1 + 2 / 1024), **passes(dev, 1024, 8, x_sweeps=2.0 if rng == "reference" else 1.0))
            dev.close()
            del gen
    # what a drop-in user of the reference API sees: generate_delta_field(seed) RETURNS a numpy array (generate.py:184-189,230).  PCIe-inclusive
    # wall time per call, field delivered into the plan's host buffer -- slab by slab behind the z pass (rf_set_host_sink: the default of
    # download=True) against the realisation followed by the download.  Never `value`: the bench line is device-resident.
    host = {}
    for rng in ("native", "reference"):
        gen = Generator(1024, 1024, 1024, spacing, power=power, rng=rng)
        dev = gen.plan_c2r.device
        t_sink = _timed(lambda: gen.generate_delta_field(seed=next(seeds), save_potential=False), dev.sync, reps=3, warm=2)

        def serial():
            gen.generate_delta_field(seed=next(seeds), save_potential=False, download=False)
            gen.download_field()
        t_serial = _timed(serial, dev.sync, reps=3, warm=1)
        host["rng='%s'" % rng] = {"ms_delivered_behind_the_z_pass": round(t_sink * 1e3, 2), "ms_realisation_then_download": round(t_serial * 1e3, 2),
                                  "GBs_over_pcie": round(4.0 * 1024 ** 3 / t_sink / 1e9, 1)}
        dev.close()
        del gen
    out["1024^3 f32 Generator.generate_delta_field(seed) -> numpy array on the host (PCIe-inclusive; not the bench line)"] = host
    # the float64 configurations in a process of their own: where the allocator puts a 17 GB plan after the plans above have come
    # and gone costs its strided passes 3 - 5 % (and after IT has been freed, later plans' store streams 20 %: DESIGN_HISTORY.md)
    # -- a fresh process is what a user of that configuration has
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-config", "float64", "--child-device", str(device)],
                           stdout=subprocess.PIPE, stderr=sys.stderr, timeout=600, check=True)
        out.update(json.loads(r.stdout.decode().strip().splitlines()[-1]))
    except Exception as e:                      # (no second process possible: in this one, and say so)
        sub = other_configs(power, spacing, device, only="float64")
        for v in sub.values():
            v["note_process"] = "measured in the bench process (child process failed: %s)" % e
        out.update(sub)
    # a grid that is NOT a power of two (transform.py:172-177 accepts any even shape): the generic mixed-radix kernels;
    # k space is materialised there -- 7 sweeps of the half spectrum instead of 5 -- and no roofline claim is made for the path
    plan = plan_for(1000, np.complex64)
    t = _timed(lambda: plan.realise(seed=next(seeds)), plan.sync, reps=3, warm=1)
    out["1000^3 f32, not a power of two (generic kernels)"] = entry(1000, t, 28 * (1 + 2 / 1000), tiled=bool(plan.tiled))
    plan.close()
    # config 4's grid on ONE GPU (34 GB): its per-GPU kernels at full axis length -- the length-2048 strided passes run as two
    # 1024-point transforms per tile; the 8-GPU job itself is `bench.py --gpus 8`
    plan = plan_for(2048, np.complex64)
    t = _timed(lambda: plan.realise(seed=next(seeds)), plan.sync, reps=3, warm=1)
    kp = passes(plan, 2048, 8)
    plan.realise_batch_prepare(1)                    # ... and as a replayed one-realisation graph (257 launches without their gaps)
    tg = _timed(lambda: plan.realise_batch([next(seeds)], want_rms=False), plan.sync, reps=3, warm=1)
    out["2048^3 f32 on one GPU"] = entry(2048, min(t, tg), 20 * (1 + 2 / 2048), ms_eager=round(t * 1e3, 3), ms_graph=round(tg * 1e3, 3), **kp)
    plan.close()
    # ... and what ONE rank of that job computes per realisation (virtual rank of 8 on this GPU: the kernels and layouts of
    # config 4, the all-to-all left out): forward = generation + x + y on the rank's 128 kz planes, backward = the gathering z pass
    per_rank, standin = {}, {}
    sweep2048 = 8.0 * 2048 * 2048 * (2048 // 2 + 1)
    for r in (0, 3):
        p = _hip.DevicePlan(2048, 2048, 2048, np.complex64, device=device, nranks=8, rank=r)
        p.set_kgrid(*powertools.ksq_axes(2048, 2048, 2048, spacing))
        p.set_power(*powertools.sigma_table(power, (2048, 2048, 2048), spacing))
        p.slab_forward(seed=1)
        p.slab_backward()
        fw, bw = [], []
        for i in range(4):
            p.sync()
            t0 = time.perf_counter()
            p.slab_forward(seed=2 + i)
            t1 = time.perf_counter()
            p.slab_backward()
            t2 = time.perf_counter()
            fw.append(t1 - t0)
            bw.append(t2 - t1)
        f_ms, b_ms = float(np.median(fw)) * 1e3, float(np.median(bw)) * 1e3
        per_rank["rank %d" % r] = {"forward_ms": round(f_ms, 3), "backward_ms": round(b_ms, 3)}
        # ... and the same rank through the REAL pipelined schedule of the multi-GPU batch (forward half of realisation i + 1 on the
        # compute stream under the exchange of realisation i on the exchange stream, then the gathering z pass), the all-to-all
        # replaced by a copy kernel of fixed width -- RCCL's channel footprint -- that reads the 7 blocks the rank would send and
        # writes the 7 segments it would receive (rf_slab_set_exchange_standin): what the exchange's LOCAL traffic and compute
        # units cost the passes it overlaps.  The links themselves are not in it.
        ent = {"forward_plus_backward_ms": round(f_ms + b_ms, 3),
               "standin_bytes_read": 7.0 / 8.0 * sweep2048 / 8.0, "standin_bytes_written": 7.0 / 8.0 * sweep2048 / 8.0}
        nreal = 8
        for w in (16, 32, 128):                    # (16 / 32 = RCCL's channel footprint; 128 = a copy that is not its own bottleneck: the asymptote)
            p.set_exchange_standin(w)
            p.realise_batch(np.arange(3, dtype=np.uint64), want_rms=False)
            p.sync()
            ts = []
            for k in range(3):
                t0 = time.perf_counter()
                p.realise_batch(np.arange(100 * k, 100 * k + nreal, dtype=np.uint64), want_rms=False)
                p.sync()
                ts.append((time.perf_counter() - t0) / nreal * 1e3)
            t0 = time.perf_counter()
            p.realise(seed=5)                      # one realisation: forward, stand-in, backward in sequence
            p.sync()
            seq = (time.perf_counter() - t0) * 1e3
            t_p = float(np.median(ts))
            ent["%d workgroups" % w] = {"pipelined_ms_per_realisation": round(t_p, 3),
                                        "slowdown_vs_forward_plus_backward": round(t_p / (f_ms + b_ms), 4),
                                        "standin_alone_ms": round(seq - f_ms - b_ms, 3),
                                        "standin_alone_GBs_each_way": round(7.0 / 8.0 * sweep2048 / 8.0 / max(seq - f_ms - b_ms, 1e-3) / 1e6, 1)}
        p.set_exchange_standin(0)
        # ... and through the schedule of the DIRECT exchange (rf_slab_set_direct_standin): the y pass stores block h into segment h of the
        # rank's own receive buffers -- the store pattern and volume of the real thing (7/8 of them would cross the links) --, no copy
        # kernel at all: what the scattered-destination, out-of-place y pass costs on this side of the links
        for overlap in (False, True):
            p.set_direct_standin(True, overlap=overlap)
            p.realise_batch(np.arange(3, dtype=np.uint64), want_rms=False)
            p.sync()
            ts = []
            for k in range(3):
                t0 = time.perf_counter()
                p.realise_batch(np.arange(100 * k, 100 * k + nreal, dtype=np.uint64), want_rms=False)
                p.sync()
                ts.append((time.perf_counter() - t0) / nreal * 1e3)
            t_p = float(np.median(ts))
            ent["direct exchange stand-in, %s" % ("storing y pass on the exchange stream" if overlap else "one stream")] = {
                "pipelined_ms_per_realisation": round(t_p, 3), "slowdown_vs_forward_plus_backward": round(t_p / (f_ms + b_ms), 4)}
        p.set_direct_standin(False)
        standin["rank %d" % r] = ent
        p.close()
    out["2048^3 / 8 kz slabs, per-rank compute on this GPU (virtual ranks, no exchange)"] = per_rank
    out["2048^3 / 8 kz slabs, one virtual rank through the pipelined batch with an exchange stand-in"] = standin
    return out


def main_multi(args, rank, world, local_rank, shape, power, spacing):
    """N > 1: one process per GPU, kz-slab / x-slab decomposition with one RCCL all-to-all per realisation.
    No torch in these processes: the unique id travels through a file, barriers and the max over ranks
    through RCCL (randomfield_amd/slab.py)."""
    from randomfield_amd import powertools, slab
    nx, ny, nz = shape
    dplan = slab.DistributedPlan(nx, ny, nz, np.complex64, device=local_rank, rank=rank, world=world, exchange="rccl")      # (the modes are calibrated below)
    plan = dplan.plan
    if world == 1:        # --force-multi on one GPU (tests): the slab pipeline with a one-rank communicator, the exchange = a copy
        from randomfield_amd import _hip
        plan.set_force_slab_path(True)
        plan.comm_init(_hip.DevicePlan.comm_unique_id())
    plan.set_kgrid(*powertools.ksq_axes(nx, ny, nz, spacing))
    plan.set_power(*powertools.sigma_table(power, (nx, ny, nz), spacing))
    # Three ways to run the slab decomposition, all giving the same field:
    #   "direct"    kz slabs; the y pass of every rank stores its output straight into the IPC-mapped receive buffers of its peers, one
    #               tiny all-reduce per realisation as the barrier -- no send / receive kernels, no extra sweep of local memory;
    #   "rccl"      kz slabs + ONE grouped ncclSend / ncclRecv all-to-all per realisation, pipelined behind the next realisation's generation;
    #   "replicate" no exchange: every rank generates all of k space and keeps its x slab (P-fold redundant x-pass arithmetic).
    # xGMI is point to point, so which is fastest depends on how many links the job spans (2 GPUs share ONE link) and on what the links
    # deliver: every candidate is timed on a few realisations and the fastest one runs.  A candidate must first reproduce the rccl mode's
    # rms on the same seeds (they are all-reduced: every rank sees the same numbers and takes the same decision).
    mode = os.environ.get("RANDOMFIELD_MULTI_MODE", "auto")            # auto | direct | rccl (= exchange) | replicate
    if mode == "exchange":
        mode = "rccl"
    calib, rejected = {}, {}
    with dplan.deadline("first exchange"):            # a rank that died leaves the others blocked in the grouped send / receive: bounded
        plan.realise(seed=998)
        plan.sync()
        dplan.barrier()

    def set_mode(m):
        """switch the plan (collectively); False when this job cannot run that way"""
        try:
            if m == "replicate":
                plan.enable_direct_exchange(False)
                plan.set_replicated_generation(True)
                return True
            if world > 1:
                plan.set_replicated_generation(False)
            return plan.enable_direct_exchange(True) if m == "direct" else not plan.enable_direct_exchange(False)
        except RuntimeError:              # a shape without that instantiation
            return False

    def trial(label, reference_rms=None):
        """two realisations with their rms (warm-up + the equality check), then three timed ones: ms per realisation, max over ranks"""
        rms = plan.realise_batch(np.arange(7000, 7002, dtype=np.uint64), want_rms=True)
        plan.sync()
        dplan.barrier()
        if reference_rms is not None and not np.allclose(rms, reference_rms, rtol=1e-6, atol=0):
            rejected[label] = "rms %r differs from the rccl exchange's %r on the same seeds" % ([float(v) for v in rms], [float(v) for v in reference_rms])
            return rms
        t0 = time.perf_counter()
        plan.realise_batch(np.arange(7100, 7103, dtype=np.uint64), want_rms=False)
        plan.sync()
        dplan.barrier()
        calib[label] = float(dplan.allreduce([(time.perf_counter() - t0) / 3], op="max")[0]) * 1e3
        return rms

    rms_rccl = None
    if mode == "auto":                                # (one rank through --force-multi: the two kz-slab modes, the exchange = the own block)
        for m in ("rccl", "direct", "replicate") if world > 1 else ("rccl", "direct"):
            if not set_mode(m):
                rejected[m] = "not available for this job (shape, or a rank could not map its peers' receive buffers)"
                continue
            with dplan.deadline("calibration of the %s mode" % m):      # (a hang names its mode and ends the job instead of blocking until the driver's limit)
                r = trial(m, rms_rccl)
            if m == "rccl":
                rms_rccl = r
        mode = min(calib, key=calib.get)
    elif mode not in ("direct", "rccl", "replicate"):
        mode = "rccl"
    if not set_mode(mode):
        raise RuntimeError("RANDOMFIELD_MULTI_MODE=%s is not available for this job" % mode)
    plan.realise(seed=999)                             # eager single step: per-phase event times
    plan.sync()
    kern = np.array(plan.kernel_ms())                  # x, y, exchange+z, all-reduce
    # kz-slab modes: the rank's kz slab as ONE block per peer, or as 4 sub-slabs (RF_FLAG_EXCHANGE_CHUNKS; the gathering z pass then reads
    # 4 x shorter segments, which one GPU's virtual ranks found 5 - 9 % faster).  RANDOMFIELD_EXCHANGE_CHUNKS = an
    # integer is honoured as given; 'auto' (default) times both layouts and keeps 4 only if it is faster AND reproduces the rms of 1.
    from randomfield_amd.slab import exchange_chunks_setting
    chunks, want_chunks = 1, exchange_chunks_setting()
    if mode in ("direct", "rccl"):
        if want_chunks == "auto":
            rms1 = None
            for c in (1, 4):
                try:
                    plan.set_exchange_chunks(c)
                except RuntimeError:
                    continue
                r = trial("%s, %d sub-slab%s" % (mode, c, "" if c == 1 else "s"), rms1)
                if c == 1:
                    rms1 = r
            best = min((k for k in calib if k.startswith(mode + ", ")), key=calib.get, default=None)
            chunks = 4 if best and best.startswith(mode + ", 4") else 1
        else:
            chunks = int(want_chunks)
        plan.set_exchange_chunks(chunks)
    plan.realise_batch(np.arange(1000, 1000 + max(args.warmup, 1), dtype=np.uint64), want_rms=False)
    plan.sync()
    dplan.barrier()
    t0 = time.perf_counter()
    # K realisations back to back, software-pipelined: realisation i+1's generation / x / y passes run
    # while realisation i's all-to-all is in flight
    plan.realise_batch(np.arange(123, 123 + args.steps, dtype=np.uint64), want_rms=False)
    plan.sync()
    dplan.barrier()
    wall = time.perf_counter() - t0
    wall = float(dplan.allreduce([wall], op="max")[0])
    mean, std = plan.moments()
    # After the timed region: the last timed realisation once more through the OTHER kz-slab exchange (direct <-> rccl).  The two must give
    # the same field (the same cells in the same places), so the same rms to the last bits the all-reduce order leaves: evidence on the
    # line itself that the mode that was timed moved every tile to where it belongs on this host.
    cross = None
    if mode in ("direct", "rccl"):
        other = "rccl" if mode == "direct" else "direct"
        try:
            if other not in calib:          # (only a mode that came through its calibration a moment ago: nothing here may cost the line)
                cross = {"other_mode": other, "agree": None, "note": "the other mode was not calibrated in this job (rejected, or the mode was forced)"}
            elif set_mode(other):
                with dplan.deadline("cross-check through the %s exchange" % other):
                    r = plan.realise_batch(np.array([123 + args.steps - 1], dtype=np.uint64), want_rms=True)
                    plan.sync()
                    dplan.barrier()
                cross = {"other_mode": other, "rms_timed_mode": float(std), "rms_other_mode": float(r[0]),
                         "agree": bool(np.isclose(float(r[0]), float(std), rtol=1e-6, atol=0))}
            else:
                cross = {"other_mode": other, "agree": None, "note": "the other mode is not available for this job"}
        finally:
            set_mode(mode)
            plan.set_exchange_chunks(chunks)
    rccl_ranks = plan.comm_size()                      # what RCCL itself says the communicator spans (ncclCommCount)
    # the N = 1 equivalent in the SAME job: every rank times the per-GPU cube (edge^3, what `--gpus 1` runs: `steps` realisations
    # replayed from one hipGraph) alone on its own GPU, no communicator involved; slowest and fastest rank reported
    from randomfield_amd import _hip
    e = args.edge
    single = _hip.DevicePlan(e, e, e, np.complex64, device=local_rank)
    single.set_kgrid(*powertools.ksq_axes(e, e, e, spacing))
    single.set_power(*powertools.sigma_table(power, (e, e, e), spacing))
    single.realise_batch_prepare(args.steps)
    single.realise_batch(np.arange(1000, 1000 + max(args.warmup, 1), dtype=np.uint64), want_rms=False)
    single.sync()
    dplan.barrier()
    t0 = time.perf_counter()
    single.realise_batch(np.arange(123, 123 + args.steps, dtype=np.uint64), want_rms=False)
    single.sync()
    t_single = (time.perf_counter() - t0) / args.steps * 1e3
    single.close()
    t_single_max = float(dplan.allreduce([t_single], op="max")[0])
    t_single_min = -float(dplan.allreduce([-t_single], op="max")[0])
    # a labelled comparator, NOT this job: every GPU its own realisation of the WHOLE grid, no exchange (what a user with an ensemble
    # of independent realisations gets from the node; it needs the whole grid to fit one GPU)
    t_full = float("inf")
    if world > 1:
        try:
            full = _hip.DevicePlan(nx, ny, nz, np.complex64, device=local_rank)
            full.set_kgrid(*powertools.ksq_axes(nx, ny, nz, spacing))
            full.set_power(*powertools.sigma_table(power, (nx, ny, nz), spacing))
            full.realise_batch_prepare(1)
            full.realise_batch(np.array([1], dtype=np.uint64), want_rms=False)
            full.sync()
            t0 = time.perf_counter()
            for i in range(3):
                full.realise_batch(np.array([2 + i], dtype=np.uint64), want_rms=False)
            full.sync()
            t_full = (time.perf_counter() - t0) / 3
            full.close()
        except RuntimeError:
            pass                                      # (the whole grid does not fit one GPU: no such comparator)
        t_full = float(dplan.allreduce([min(t_full, 1e9)], op="max")[0])
    cells = float(nx) * ny * nz
    sweep = 8.0 * nx * ny * (nz // 2 + 1)
    xgmi_bytes = (world - 1) / world * sweep / world if mode in ("direct", "rccl") else 0.0     # egress per GPU of the exchange
    out = {
        "metric": "Mcells/s for N^3 delta(x) realisation",
        "value": round(cells * args.steps / wall / 1e6, 1), "unit": "Mcells/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%dx%d float32 delta(x) realisations, x-slab decomposition over %d GPUs (%s), "
                               "native Philox4x32-7 RNG, shipped 500-row P(k)"
                               % (nx, ny, nz, world,
                                  {"rccl": "kz-slab generation + ONE RCCL all-to-all per realisation, overlapped with the next realisation's generation",
                                   "direct": "kz-slab generation; the y pass stores into the peers' IPC-mapped receive buffers (the exchange), "
                                             "one tiny all-reduce per realisation as the barrier",
                                   "replicate": "replicated generation: every rank generates all of k space and keeps its x slab, no all-to-all"}[mode]),
                   "grid": [nx, ny, nz], "rms_last": round(std, 6), "multi_gpu_mode": mode, "exchange_sub_slabs": chunks, "rccl_ranks": rccl_ranks,
                   "launcher": "bench.py's own child ranks" if os.environ.get("RANDOMFIELD_LAUNCH_NONCE", "").startswith("bench-") else "external (torch.distributed.run)",
                   "mode_calibration_ms_per_step": {k: round(v, 3) for k, v in calib.items()},
                   "modes_rejected": rejected, "exchange_cross_check": cross},
        "single_gpu_equivalent": {"ms_per_step": round(t_single_max, 4), "ms_per_step_fastest_rank": round(t_single_min, 4),
                                  "grid": [e, e, e], "Mcells_s": round(float(e) ** 3 / t_single_max / 1e3, 1),
                                  "speedup_of_this_job": round(cells * args.steps / wall / (float(e) ** 3 / (t_single_max * 1e-3)), 3),
                                  "note": "every rank alone on its own GPU in this same job (%d^3, %d realisations from one hipGraph, "
                                          "as `--gpus 1`); slowest rank quoted" % (e, args.steps),
                                  # NOT config 4 and not `value`: N independent realisations of the whole grid, one per GPU, no exchange
                                  "ensemble_Mcells_s": (round(world * cells / t_full / 1e6, 1) if world > 1 and t_full < 1e8 else None),
                                  "ensemble_note": "comparator only: every GPU generates its OWN %dx%dx%d realisation (no slab decomposition, no "
                                                   "exchange; slowest rank): what an ensemble of independent realisations gets from the node" % (nx, ny, nz)},
        "pipeline": {"algorithmic_GBs": round(5 * sweep * args.steps / wall / 1e9, 1),
                     "frac_of_hbm_peak": round(5 * sweep * args.steps / wall / 1e9 / (HBM_PEAK_GBS * world), 4),
                     "kernel_ms_rank0_unpipelined_step": {"x": round(float(kern[0]), 4), "y": round(float(kern[1]), 4),
                                                   "exchange+z": round(float(kern[2]), 4),
                                                   "allreduce": round(float(kern[3]), 4)},
                     "xgmi_egress_bytes_per_gpu": xgmi_bytes,
                     # the exchange's own roofline: xGMI is point to point, one link per peer.  AMD's 153.6 GB/s per link is the sum of
                     # both directions (7 links = 1075 GB/s aggregate), i.e. 76.8 GB/s per link and direction; the other reading is kept
                     # beside it.  `achieved` = what every link must have carried per direction if the step time were all exchange.
                     "xgmi": ({"links_used": world - 1, "bytes_per_link_per_direction": xgmi_bytes / (world - 1),
                               "min_exchange_ms_at_76.8_GBs_per_direction": round(xgmi_bytes / (world - 1) / 76.8e9 * 1e3, 3),
                               "min_exchange_ms_at_153.6_GBs_per_direction": round(xgmi_bytes / (world - 1) / 153.6e9 * 1e3, 3),
                               "GBs_per_link_per_direction_if_step_were_all_exchange": round(xgmi_bytes / (world - 1) / (wall / args.steps) / 1e9, 1)}
                              if world > 1 and xgmi_bytes else None)},
        "roofline": {"bound": "hbm", "kernel": "whole pipeline (5 sweeps) over %d GPUs" % world,
                     "achieved": round(5 * sweep * args.steps / wall / 1e9, 1), "peak": HBM_PEAK_GBS * world,
                     "unit": "GB/s", "frac": round(5 * sweep * args.steps / wall / 1e9 / (HBM_PEAK_GBS * world), 4),
                     "traffic": None},
    }
    dplan.barrier()
    plan.close()
    if rank == 0:
        print(json.dumps(out))


def build_line(shape, steps, warmup, gpus, wall, gpu_ms, std, kern, merged_ms, merged_n, nslab, slab_planes, traffic_path=None):
    """The N = 1 JSON line from the measured numbers -- no GPU call in here, so that tests/test_bench_contract.py can run it with stubbed
    timings: wall = seconds of the `steps` timed realisations, gpu_ms = the same by HIP events, std = rms of the last field,
    kern = [x main kernel, y, z, reduce, x launch over the kz = 0 tiles] in ms (eager realisations), merged_ms / merged_n = mean
    duration and launches per realisation of the merged z + y launch (None / 0 where the shape has none), nslab / slab_planes = how
    the y / z passes are launched."""
    nx, ny, nz = shape
    cells = float(nx) * ny * nz
    sweep = 8.0 * nx * ny * (nz // 2 + 1)           # bytes of one sweep of the packed complex64 array
    # kern = [x main kernel, y, z, reduce, x launch over the kz = 0 tiles]; the x pass writes one sweep, of which the
    # main kernel writes all tiles but one per ky row
    names = ["x pass (generation + FFT, write only; its three launches: the side buffer of repaired kz = 0 slots, the kz = 0 tiles, all others)",
             "y pass (FFT in place)", "z pass (c2r + moments, in place)"]
    # the x pass is three launches (side-buffer fill, the tiles that hold slot kz = 0, then all others): the PASS is what gets compared
    pass_ms = np.array([kern[0] + kern[4], kern[1], kern[2]])
    alg = [sweep, 2 * sweep, 2 * sweep]
    dom = int(np.argmax(pass_ms))
    achieved = alg[dom] / (pass_ms[dom] * 1e-3) / 1e9          # = bytes per launch / mean launch duration (both divided by the launches)
    # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (profiles/), if they
    # were taken on this grid; bench.py itself cannot run the profiler
    # (per LAUNCH, like `achieved`: the y and z passes are nslab launches each when they run slab by slab; the x pass is its
    # two launches together)
    launches = [1, nslab, nslab]
    traffic, traffic_source, pass_traffic = None, None, {}
    keys = [["FastGenColIOT<0,", "FastGenColIOT<1,", "FastGenColIOT<3,", "fix_fill_kernel"], ["PlainColIO", "XposeColIO", "Pair2ColIO"], ["row_c2r_kernel"]]
    try:
        tj = json.load(open(traffic_path or os.path.join(ROOT, "profiles", "traffic_latest.json")))
    except Exception as e:                               # no committed profile: say so instead of a silent null
        tj, traffic_source = None, "profiles/traffic_latest.json not readable (%s)" % e
    if tj is not None:
        if (nx, ny, nz) == tuple(tj.get("grid", (1024, 1024, 1024))) and gpus == 1:
            for i, k in enumerate(("x", "y", "z")):
                tot = sum(v["total"] for name, v in tj["kernels"].items() if any(key in name for key in keys[i]))
                pass_traffic[k] = tot if tot > 0 else None
            traffic = pass_traffic[("x", "y", "z")[dom]]
            src = tj.get("source") or tj.get("note") or "profiles/traffic_latest.json"
            traffic_source = ("NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed as %s" % src if traffic
                              else "no kernel of %s matched %r" % (src, keys[dom]))
            if traffic and tj.get("yz_slabs") not in (None, nslab):
                traffic_source += " (taken with %s y / z launches per realisation, this run has %d)" % (tj.get("yz_slabs"), nslab)
        else:
            traffic_source = "the committed PMC passes are for a %s grid on one GPU, not this run's" % (tj.get("grid", [1024, 1024, 1024]),)
    out = {
        "metric": "Mcells/s for N^3 delta(x) realisation",
        "value": round(cells * steps / wall / 1e6, 1),
        "unit": "Mcells/s",
        "n_gpus": gpus, "steps": steps, "warmup": warmup,
        "ms_per_step": round(wall / steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%dx%d float32 delta(x) realisations back to back (hipGraph replay), "
                               "native Philox4x32-7 RNG, shipped 500-row P(k), spacing 2.5 Mpc/h" % (nx, ny, nz),
                   "grid": [nx, ny, nz], "rms_last": round(std, 6)},
        "gpu_ms_per_step_events": round(gpu_ms / steps, 4),
        "pipeline": {"algorithmic_GBs": round(5 * sweep * steps / wall / 1e9, 1),
                     "frac_of_hbm_peak": round(5 * sweep * steps / wall / 1e9 / HBM_PEAK_GBS, 4),
                     "kernel_ms": {"x": round(float(kern[0]), 4), "y": round(float(kern[1]), 4),
                                   "z": round(float(kern[2]), 4), "reduce": round(float(kern[3]), 4),
                                   "x_kz0_tiles": round(float(kern[4]), 4)},
                     "kernel_ms_note": "HIP-event intervals around each pass of EAGER realisations in this process (they include the "
                                       "launch gaps, and y / z are the sums over their %d launches of %d x planes): their sum is "
                                       "larger than ms_per_step, which is the graph replay" % (nslab, slab_planes),
                     "yz_slabs": nslab,
                     "launches_per_realisation": {"x": 3, "y": nslab, "z": nslab, "reduce": 1},
                     "traffic_bytes_per_launch": pass_traffic,
                     "pass_frac_of_hbm_peak": {k: round(alg[i] / (pass_ms[i] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                               for i, k in enumerate(("x", "y", "z"))}},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": alg[dom] / launches[dom], "avg_ms": round(float(pass_ms[dom]) / launches[dom], 5),
                     "launches_per_realisation": launches[dom],
                     "whole_pipeline_frac": round(5 * sweep * steps / wall / 1e9 / HBM_PEAK_GBS, 4)},
    }
    # An estimate of the bytes that really cross the HBM pins: `achieved` / `frac` price ALGORITHMIC bytes, but when
    # the y and z passes run slab by slab the slab goes from the y pass to the z pass through the 256 MiB Infinity Cache and the z pass
    # overwrites it in place, so per realisation only the x pass's write (S), the y pass's read (S) and the z pass's result (S) reach
    # HBM: 3 of the 5 sweeps.  No counter on this chip separates Infinity-Cache hits from HBM accesses (FETCH_SIZE / WRITE_SIZE are L2-side
    # and count the hits: MI355X_MICROARCH.md), hence an estimate: it assumes a perfect hand-off (every line the y pass writes is still
    # in the cache when the z pass reads it and is overwritten before it is evicted).
    handoff = nslab > 1
    hbm_per_pass = [sweep, sweep if handoff else 2 * sweep, sweep if handoff else 2 * sweep]
    out["roofline"]["hbm_bytes_est"] = hbm_per_pass[dom] / launches[dom]
    out["roofline"]["frac_hbm_est"] = round(hbm_per_pass[dom] / (pass_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    out["pipeline"]["hbm_bytes_est_per_realisation"] = float(sum(hbm_per_pass))
    out["pipeline"]["frac_hbm_est"] = round(sum(hbm_per_pass) * steps / wall / 1e9 / HBM_PEAK_GBS, 4)
    hbm_note = ("ESTIMATE, not a counter: algorithmic bytes minus what the slab hand-off keeps in the 256 MiB Infinity Cache (the z pass reads "
                "what the y pass has just written, and overwrites it before it is evicted): x writes S, y reads S, z's result S = 3 of "
                "the 5 sweeps per realisation reach HBM" if handoff else
                "no slab hand-off on this grid (whole-grid passes): the estimate equals the algorithmic bytes")
    out["roofline"]["hbm_bytes_est_note"] = hbm_note
    if merged_ms:
        # the kernel the timed region spends most of its time in: (nslab - 1) merged launches per realisation, each the z pass of one
        # slab (read + write) and the y pass of the next (read + write): 4 sweeps of a slab
        alg_m = 4 * sweep / nslab
        ach_m = alg_m / (merged_ms * 1e-3) / 1e9
        tm = None
        if tj is not None and (nx, ny, nz) == tuple(tj.get("grid", (1024, 1024, 1024))):
            tot = sum(v["total"] for name, v in tj["kernels"].items() if "yz_merged_kernel" in name)
            tm = tot if tot > 0 else None
        out["roofline"] = {"bound": "hbm", "kernel": "yz_merged_kernel: z pass (c2r + moments) of slab s + y pass (FFT in place) of slab s + 1 in one launch",
                           "achieved": round(ach_m, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_m / HBM_PEAK_GBS, 4),
                           "traffic": tm,
                           "traffic_source": (traffic_source if tm else "no yz_merged_kernel entry in profiles/traffic_latest.json"),
                           "algorithmic_bytes_per_launch": alg_m, "avg_ms": round(merged_ms, 5), "launches_per_realisation": merged_n,
                           "note": "HIP events behind every launch of eager realisations (rf_set_merged_yz(2)); the same passes one launch "
                                   "each: pipeline.kernel_ms / pass_frac_of_hbm_peak",
                           "unmerged_dominant_pass": {"kernel": names[dom], "frac": round(achieved / HBM_PEAK_GBS, 4),
                                                      "avg_ms": round(float(pass_ms[dom]) / launches[dom], 5)},
                           "whole_pipeline_frac": round(5 * sweep * steps / wall / 1e9 / HBM_PEAK_GBS, 4),
                           # of the launch's four slab sweeps the y half's read and the z half's result cross the HBM pins
                           "hbm_bytes_est": 2 * sweep / nslab,
                           "frac_hbm_est": round(2 * sweep / nslab / (merged_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "hbm_bytes_est_note": hbm_note}
        out["pipeline"]["launches_per_realisation"] = {"x": 3, "y": 1, "z + y merged": merged_n, "z": 1, "reduce": 1}
    out["config"]["parity"] = ("native stream = this repo's own definition (no reference counterpart): the kernel instantiations "
                               "of this run are value-checked against the oracle's float64 restatement at 1e-5 * rms in "
                               "tests/test_gpu_parity.py::test_native_generation_bench_instantiations_against_oracle; the "
                               "same-seed path (rng='reference') is in other_configs")
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` outside any launcher: start N fresh ranks of this script (one per GPU, RANK = LOCAL_RANK = r)
    with the environment `torch.distributed.run` would give them, relay rank 0's standard output (the ONE JSON line) and
    return the worst exit status.  Nothing in this process has loaded HIP: the ranks are children, never an exec.  A rank that
    fails takes the job down: the others are given RANDOMFIELD_COLLECTIVE_TIMEOUT seconds (their own watchdog,
    slab.Deadline) and are then terminated by pid."""
    import socket
    import subprocess
    with socket.socket() as sock:                    # a free port for MASTER_PORT (names the rendezvous; nothing listens on it)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    base = dict(os.environ)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                 "RANDOMFIELD_LAUNCH_NONCE": "bench-%d-%d" % (os.getpid(), time.time_ns()),
                 "HSA_ENABLE_IPC_MODE_LEGACY": base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))      # only rank 0 speaks on stdout
    grace = float(os.environ.get("RANDOMFIELD_COLLECTIVE_TIMEOUT", "180")) + 30.0
    out0, failed_at, codes = b"", None, [None] * n
    import threading
    box = {}
    reader = threading.Thread(target=lambda: box.setdefault("out", procs[0].stdout.read()))
    reader.daemon = True
    reader.start()
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed_at is None:
                    failed_at = time.time()
                    sys.stderr.write("bench.py: rank %d exited with status %d; waiting up to %.0f s for the others\n" % (r, codes[r], grace))
        if failed_at is not None and time.time() - failed_at > grace:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()                    # exactly the pids started above
        time.sleep(0.05)
    reader.join(10)
    out0 = box.get("out", b"")
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    worst = max((abs(c) for c in codes), default=0)
    if worst:
        sys.stderr.write("bench.py: exit statuses of the %d ranks: %r\n" % (n, codes))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--edge", type=int, default=1024, help="per-GPU cube edge")
    ap.add_argument("--cpu-sample", type=int, default=0, help="cube edge of the CPU baseline sample (0 = the workload's own edge)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--force-multi", action="store_true", help="debug: run the N>1 code path with one rank")
    ap.add_argument("--child-config", default=None, help=argparse.SUPPRESS)      # (other_configs' float64 part in a process of its own)
    ap.add_argument("--child-device", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: no launcher has prepared RANK / WORLD_SIZE, so this process becomes the launcher.
        # It starts the N ranks as CHILDREN (never an exec) before anything here has touched HIP, and relays rank 0's line.
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RANDOMFIELD_BENCH_STUB_CHILD"):
        # test hook (tests/test_bench_launcher.py, no GPU): a rank that only reports how it was started
        print(json.dumps({"stub": True, "rank": rank, "world": world, "local_rank": local_rank, "gpus": args.gpus, "steps": args.steps,
                          "warmup": args.warmup, "master": [os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")],
                          "nonce": os.environ.get("RANDOMFIELD_LAUNCH_NONCE"), "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
        sys.stdout.flush()
        sys.exit(int(os.environ.get("RANDOMFIELD_BENCH_STUB_RC_RANK%d" % rank, "0")))
    if world != args.gpus:
        args.gpus = world                            # the launcher's world size is what runs

    from randomfield_amd import _hip, powertools
    _hip.require_gpu()                              # no GPU / no library -> loud failure, never a CPU run
    if args.child_config:
        print(json.dumps(other_configs(powertools.load_default_power(), 2.5, args.child_device, only=args.child_config)))
        return

    spacing = 2.5
    nx, ny, nz = grid_for(args.gpus, args.edge)
    power = powertools.load_default_power()
    if world > 1 or args.force_multi:
        return main_multi(args, rank, world, local_rank, (nx, ny, nz), power, spacing)
    plan = _hip.DevicePlan(nx, ny, nz, np.complex64, device=local_rank)
    plan.set_kgrid(*powertools.ksq_axes(nx, ny, nz, spacing))
    plan.set_power(*powertools.sigma_table(power, (nx, ny, nz), spacing))

    # --- per-kernel timing (HIP events on the plan's stream, eager launches) ---
    kern = np.zeros(5)
    reps = max(3, min(args.steps, 10))
    plan.realise(seed=1)
    plan.sync()
    for i in range(reps):
        plan.realise(seed=100 + i)
        plan.sync()
        kern += np.array(plan.kernel_ms())
    kern /= reps

    # the graph-replayed realisations of the timed region put the z pass of slab s and the y pass of slab s + 1 into ONE launch
    # (rf_k_yz.hip) where the shape allows: that launch, timed with an event behind every launch of eager realisations
    merged_ms, merged_n = None, 0
    try:
        plan.set_merged_yz(2)
        tot = cnt = 0
        for i in range(reps):
            plan.realise(seed=200 + i)
            plan.sync()
            ms, n_l = plan.merged_yz_ms()
            tot += ms
            cnt += n_l
        merged_ms, merged_n = tot / cnt, cnt // reps
    except RuntimeError:
        pass                                        # (a shape rf_k_yz.hip does not serve: one launch per pass and slab, as timed above)
    plan.set_merged_yz(1)

    # --- the timed region: W warm-up + K steps replayed from one hipGraph -----
    plan.realise_batch_prepare(args.steps)           # capture + instantiate outside the timed region
    if args.warmup > 0:
        plan.realise_batch(np.arange(1000, 1000 + args.warmup, dtype=np.uint64), want_rms=False)
    plan.sync()
    seeds = np.arange(123, 123 + args.steps, dtype=np.uint64)
    t0 = time.perf_counter()
    plan.realise_batch(seeds, want_rms=False)
    plan.sync()
    wall = time.perf_counter() - t0
    gpu_ms = plan.elapsed_ms()
    mean, std = plan.moments()

    nslab, slab_planes = plan.yz_slabs()
    out = build_line((nx, ny, nz), args.steps, args.warmup, args.gpus, wall, gpu_ms, std, kern, merged_ms, merged_n, nslab, slab_planes)
    plan.close()
    if rank == 0 and args.gpus == 1 and not args.no_other_configs:
        out["other_configs"] = other_configs(power, spacing, local_rank)
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(power, spacing, args.cpu_sample or args.edge)
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
