"""
Multi-GPU slab decomposition of the hot path: one process per GPU.

k space is split by kz slabs -- with the Nyquist plane packed into slot kz = 0 there are exactly nz/2
planes, so rank r owns planes [r*nzl, (r+1)*nzl), nzl = nz/(2P).  Generation and the x and y passes are
local.  ONE all-to-all then moves block [x in slab h][all y][kz in slab g] from rank g to rank h, after
which rank h owns the x slab [h*nxl, (h+1)*nxl), nxl = nx/P, runs the z pass on rows gathered from the P
received blocks and holds delta[x in its slab].  The rms needs one 2-double all-reduce.

This module holds the layout arithmetic (shared by the C library's design, the tests and bench.py) and
the process plumbing.  The GPU path needs no torch: rank 0's RCCL unique id reaches the other ranks of
the (single-node) job through a small file, and barriers / reductions of timings go through RCCL itself.
All collectives are issued by ``librandomfield_hip.so`` on the plan's HIP stream.  (``torch.distributed``
with gloo is used by the CPU tests only.)
"""
from __future__ import annotations

import os
import time

import numpy as np

__all__ = ["slab_layout", "exchange_blocks", "direct_store_bases", "exchange_chunks_setting", "exchange_mode_setting", "gather_rows", "side_array_planes", "split_side_array",
           "assemble_side_array", "shared_replay_layout", "shared_replay_pack", "exchange_unique_id", "launch_nonce", "init_process_group", "DistributedPlan", "SlabHostPlan", "SlabHostReversePlan", "Deadline"]


def slab_layout(nx, ny, nz, nranks, rank):
    """Sizes and offsets of rank ``rank``'s slabs."""
    nzc = nz // 2
    if nx % nranks or nzc % nranks or (nzc // nranks) % 2:
        raise ValueError("nx and nz/2 must be divisible by the number of ranks (nz/(2*ranks) even)")
    nxl, nzl = nx // nranks, nzc // nranks
    return dict(nxl=nxl, nzl=nzl, x0=rank * nxl, kz0=rank * nzl, nzc=nzc,
                block_elems=nxl * ny * nzl, local_elems=nx * ny * nzl)


def exchange_blocks(local_k, nranks):
    """Split a rank's post-y-pass array [nx][ny][nzl] into the P send blocks [nxl][ny][nzl]
    (block h goes to rank h).  x is the slowest axis, so the blocks are contiguous slices."""
    nx = local_k.shape[0]
    nxl = nx // nranks
    return [local_k[h * nxl:(h + 1) * nxl] for h in range(nranks)]


def direct_store_bases(nx, ny, nz, nranks, rank, chunks=1):
    """The arithmetic of the DIRECT exchange (rf_capi.hip rebuild_peer_tab, rf_fft.h DirectColIO): rank ``rank``'s y pass stores every
    output tile straight into the receive buffer of the rank h = ix // nxl that owns the tile's x plane.  The receive layout is
    [source rank][chunk][nxl][ny][nzl / chunks] -- what the gathering z pass reads -- and a cell's offset inside its block is the same on
    both sides, so the store address is the LOCAL cell offset plus a per-(chunk, destination) base.  Returns ``base[c][h]`` in cells,
    relative to the start of rank h's receive buffer (it may be negative: the local offset of block h starts at h * blk):
    cell ``off`` of sub-slab c's local array [nx][ny][nzl / chunks] goes to  receive_h[base[c][h] + off]."""
    lay = slab_layout(nx, ny, nz, nranks, rank)
    if lay["nzl"] % chunks:
        raise ValueError("the number of exchange chunks must divide nz / (2 ranks)")
    blk = lay["nxl"] * ny * (lay["nzl"] // chunks)
    return [[(rank * chunks + c - h) * blk for h in range(nranks)] for c in range(chunks)]


def exchange_chunks_setting(explicit=None):
    """ONE parser for ``RANDOMFIELD_EXCHANGE_CHUNKS`` (bench.py, Generator): ``'auto'`` or an integer >= 1; ``explicit`` (an argument
    of the caller) wins over the environment.  Returns 'auto' or the integer; anything else raises ValueError with the variable's name."""
    raw = explicit if explicit is not None else os.environ.get("RANDOMFIELD_EXCHANGE_CHUNKS", "auto")
    if raw is None or (isinstance(raw, str) and raw.strip().lower() in ("", "auto")):
        return "auto"
    try:
        n = int(raw)
    except (TypeError, ValueError):
        raise ValueError("RANDOMFIELD_EXCHANGE_CHUNKS / exchange_chunks must be 'auto' or an integer >= 1, got %r" % (raw,))
    if n < 1:
        raise ValueError("RANDOMFIELD_EXCHANGE_CHUNKS / exchange_chunks must be >= 1, got %d" % n)
    return n


def exchange_mode_setting(explicit=None):
    """``RANDOMFIELD_EXCHANGE``: how a multi-rank plan moves its blocks between the y and z passes -- 'direct' (the y pass stores into
    the peers' IPC-mapped receive buffers; an error if some rank cannot), 'rccl' (grouped ncclSend / ncclRecv) or 'auto' (direct if
    EVERY rank can, else rccl: the default)."""
    raw = explicit if explicit is not None else os.environ.get("RANDOMFIELD_EXCHANGE", "auto")
    mode = str(raw).strip().lower() or "auto"
    if mode not in ("auto", "direct", "rccl"):
        raise ValueError("RANDOMFIELD_EXCHANGE / exchange must be 'auto', 'direct' or 'rccl', got %r" % (raw,))
    return mode


def gather_rows(recv_blocks):
    """Assemble the z-pass input [nxl][ny][nz/2] of a rank from its P received blocks
    (block g holds the kz planes of rank g's slab)."""
    return np.concatenate(recv_blocks, axis=2)


def side_array_planes(nz, nranks, rank):
    """kz planes held, in order, by the k-space side arrays (k buffer, saved potential, resident deviates) of a rank:
    its own nz/(2 ranks) planes and then the Nyquist plane nz/2, which rank 0 packs into slot kz = 0 of the
    device-internal layout (the other ranks carry the slot but never read it).  One rank: 0 .. nz/2, the reference's
    own (nx, ny, nz/2 + 1) layout."""
    nzl = nz // 2 // nranks
    return list(range(rank * nzl, (rank + 1) * nzl)) + [nz // 2]


def split_side_array(full, nranks, rank):
    """A rank's share of an (nx, ny, nz/2 + 1) k-space array."""
    nz = 2 * (full.shape[2] - 1)
    return np.ascontiguousarray(full[:, :, side_array_planes(nz, nranks, rank)])


def assemble_side_array(parts):
    """The (nx, ny, nz/2 + 1) array from every rank's share (the Nyquist plane is taken from rank 0)."""
    nzl = parts[0].shape[2] - 1
    return np.concatenate([a[:, :, :nzl] for a in parts] + [parts[0][:, :, nzl:]], axis=2)


def shared_replay_layout(counts, nx, ny, nz, nranks):
    """The arithmetic of the shared replay of the reference's stream (``rf_mt_share_pack`` in rf_capi.hip; random.py:24-28 is
    one sequential stream of cells in (ix, iy, kz) order, kz = 0 .. nz/2).  ``counts[s]`` = accepted pairs of segment s;
    rank r replays segments [r S / P, (r + 1) S / P) and therefore holds the stream cells [cell_begin[r], cell_begin[r + 1]).
    Rank q needs "stream q": of every row its own nz/(2P) planes and the Nyquist plane, in stream order -- rows of
    nzl + 1 pairs, the layout of the rank's side arrays (:func:`side_array_planes`).  Returns the cell ranges and, in pairs,
    ``sendcnt[r][q]`` / ``sendoff[r][q]`` (rank r's send buffer, one dense region per destination) and ``recvoff[q][r]`` (where
    rank r's pairs start in stream q)."""
    counts = np.asarray(counts, np.uint64).astype(object)
    nseg, nzh, nzl = len(counts), nz // 2 + 1, nz // 2 // nranks
    ncells = nx * ny * nzh
    off = [0]
    for c in counts:
        off.append(off[-1] + int(c))
    if off[-1] < ncells:
        raise ValueError("the segments hold fewer accepted pairs than the grid has cells")
    seg_begin = [r * nseg // nranks for r in range(nranks + 1)]
    cb = [min(off[b], ncells) for b in seg_begin]
    cb[-1] = ncells

    def fq(q, c):               # stream-q cells in front of stream cell c
        col, kz = divmod(c, nzh)
        return col * (nzl + 1) + min(max(kz - q * nzl, 0), nzl)

    sendcnt = [[fq(q, cb[r + 1]) - fq(q, cb[r]) for q in range(nranks)] for r in range(nranks)]
    sendoff = [[sum(row[:q]) for q in range(nranks)] for row in sendcnt]
    recvoff = [[fq(q, cb[r]) for r in range(nranks)] for q in range(nranks)]
    return dict(seg_begin=seg_begin, cell_begin=cb, sendcnt=sendcnt, sendoff=sendoff, recvoff=recvoff, nzl=nzl, nzh=nzh,
                stream_pairs=nx * ny * (nzl + 1))


def shared_replay_pack(pairs, first_cell, nz, nranks):
    """Rank-side packing: ``pairs`` (n, 2) are the stream cells first_cell .. first_cell + n; returns one array per destination
    rank (its planes' cells in stream order; a Nyquist cell goes to every rank)."""
    nzh, nzl = nz // 2 + 1, nz // 2 // nranks
    kz = (first_cell + np.arange(len(pairs))) % nzh
    return [pairs[((kz >= q * nzl) & (kz < (q + 1) * nzl)) | (kz == nzh - 1)] for q in range(nranks)]


_plans_made = 0          # DistributedPlans built by this process so far: every rank builds them in the same order
_T_IMPORT = time.time()


def _proc_start_ticks(pid):
    """Start time of process `pid` in clock ticks since boot (field 22 of /proc/<pid>/stat), or 0 when it cannot be read.
    (pid, start ticks) names a process for the lifetime of the machine: pids are reused, the pair is not."""
    try:
        with open("/proc/%d/stat" % pid, "rb") as f:
            stat = f.read().decode("ascii", "replace")
        return int(stat[stat.rindex(")") + 2:].split()[19])
    except (OSError, ValueError, IndexError):
        return 0


def _process_start_time():
    """When THIS process started, seconds since the epoch (boot time has 1 s resolution; never later than the truth by more
    than that).  Falls back to the time this module was imported."""
    try:
        ticks = _proc_start_ticks(os.getpid())
        with open("/proc/stat", "rb") as f:
            for line in f:
                if line.startswith(b"btime"):
                    return int(line.split()[1]) + ticks / float(os.sysconf("SC_CLK_TCK"))
    except (OSError, ValueError):
        pass
    return _T_IMPORT



def launch_nonce():
    """A string every rank of ONE launch agrees on without talking to the others, and that no other launch on this node
    shares: ``RANDOMFIELD_LAUNCH_NONCE`` when the launcher sets it (``bench.py --gpus N`` does, for the ranks it starts itself),
    else the launcher PROCESS -- the ranks of ``torch.distributed.run`` are children of one agent process, so its (pid, start
    time) identifies the launch -- plus MASTER_PORT and the elastic run id."""
    env = os.environ.get("RANDOMFIELD_LAUNCH_NONCE")
    if env:
        return "".join(c if (c.isalnum() or c in "-.") else "_" for c in env)
    ppid = os.getppid()
    return "%d.%d.%s.%s" % (ppid, _proc_start_ticks(ppid), os.environ.get("MASTER_PORT", "0"),
                            "".join(c if c.isalnum() else "_" for c in os.environ.get("TORCHELASTIC_RUN_ID", "none")))


def _rendezvous_path(serial=None):
    """A file name every rank of one launch agrees on for its `serial`-th DistributedPlan: the launch's nonce
    (:func:`launch_nonce`) and the per-process plan counter (two plans of one job never share a file: a fast rank cannot pick
    up the previous plan's id)."""
    serial = _plans_made if serial is None else serial
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "randomfield_uid_%s_%d" % (launch_nonce(), serial))


def exchange_unique_id(rank, world, make_uid, timeout=300.0, path=None, not_before=None):
    """Hand rank 0's RCCL unique id to every rank of a single-node job WITHOUT importing torch
    (a process that loads PyTorch's bundled ROCm runtime next to the system one is asking for trouble).
    Rank 0 removes any stale file of that name, then writes the 128 bytes, followed by the launch's nonce, atomically.  The
    others poll for the file and accept it only if (a) it is 128 bytes + THIS launch's nonce and (b) it was written after this
    process started (mtime >= `not_before`, default: the process's own start time minus the clock's 2 s of slack; not applied
    when the launcher set RANDOMFIELD_LAUNCH_NONCE, which is unique per launch already).  (a) rejects
    the file of any other launch; (b) the one case (a) cannot see -- a launcher without a nonce of its own that starts job
    after job from ONE long-lived parent with the same MASTER_PORT, where a crashed job's file has this job's name: a rank
    that gets here before rank 0 has removed that file would otherwise read a dead communicator's id and sit in
    ncclCommInitRank until the watchdog (:class:`Deadline`) ends it."""
    path = _rendezvous_path() if path is None else path
    if world == 1:
        return make_uid()
    tag = b"|" + launch_nonce().encode("ascii", "replace")
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
        uid = make_uid()
        if len(uid) != 128:
            raise ValueError("an RCCL unique id is 128 bytes, got %d" % len(uid))
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid + tag)
        os.replace(tmp, path)
        return uid
    if not_before is None:
        # a launcher-made nonce (one per launch) already rules out every other launch's file: no mtime test then -- it would only
        # reject a valid id when ranks start staggered by more than the slack, or /tmp's clock is skewed
        not_before = float("-inf") if os.environ.get("RANDOMFIELD_LAUNCH_NONCE") else _process_start_time() - 2.0
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                mtime = os.fstat(f.fileno()).st_mtime
                blob = f.read()
            if len(blob) == 128 + len(tag) and blob[128:] == tag and mtime >= not_before:
                return blob[:128]
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise RuntimeError("rank %d timed out after %.0f s waiting for rank 0's RCCL unique id at %s" % (rank, timeout, path))
        time.sleep(0.02)


class Deadline(object):
    """Host-side watchdog around a collective step that can block for ever when a rank has died (ncclCommInitRank, the
    first grouped send / receive): every rank checks in with a file before the step; if the step has not returned after
    `seconds`, a timer thread prints which ranks never checked in and ends THIS process with a non-zero status
    (os._exit from a thread -- the main thread is inside the blocked C call; no re-exec of a process that holds the GPU)."""

    def __init__(self, what, rank, world, seconds=None, path=None):
        self.what, self.rank, self.world = what, rank, world
        self.seconds = float(os.environ.get("RANDOMFIELD_COLLECTIVE_TIMEOUT", "180")) if seconds is None else seconds
        self.base = (_rendezvous_path() if path is None else path) + "." + "".join(c if c.isalnum() else "_" for c in what)
        self.timer = None

    def _expired(self):
        import sys
        missing = [r for r in range(self.world) if not os.path.exists("%s.rank%d" % (self.base, r))]
        sys.stderr.write("randomfield_amd: rank %d of %d: '%s' did not finish within %.0f s; ranks that never reached it: %s\n"
                         % (self.rank, self.world, self.what, self.seconds, missing if missing else "none (all checked in: a hung collective)"))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        import threading
        if self.world > 1 and self.seconds > 0:
            with open("%s.rank%d" % (self.base, self.rank), "w") as f:
                f.write("%d\n" % os.getpid())
            self.timer = threading.Timer(self.seconds, self._expired)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
            try:
                os.remove("%s.rank%d" % (self.base, self.rank))
            except OSError:
                pass
        return False


def init_process_group():
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run).
    Returns (dist, rank, world, local_rank).  gloo: only tiny host-side messages go through it.
    (Used by the CPU tests; the GPU path needs no torch at all -- see DistributedPlan.)"""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return dist, rank, world, local_rank


def broadcast_bytes(dist, payload, src=0):
    """Broadcast a bytes object from rank ``src`` over torch.distributed (CPU tests)."""
    box = [payload if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


class DistributedPlan(object):
    """One rank's share of a multi-GPU plan: DevicePlan(nranks, rank) + RCCL communicator.

    Reads RANK / WORLD_SIZE / LOCAL_RANK from the environment (as set by ``torch.distributed.run``).
    After construction ``barrier()`` and ``allreduce()`` go through RCCL on the plan's stream.
    ``plan.set_replicated_generation(True)`` switches to the communication-free mode (every rank generates all
    of k space and keeps its x slab; native generator only) -- see DESIGN.md section 5."""

    def __init__(self, nx, ny, nz, dtype=np.complex64, device=None, rank=None, world=None, exchange=None):
        from . import _hip
        mode = exchange_mode_setting(exchange)
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.layout = slab_layout(nx, ny, nz, self.world, self.rank)
        self.plan = _hip.DevicePlan(nx, ny, nz, dtype, device=local_rank if device is None else device,
                                    nranks=self.world, rank=self.rank)
        global _plans_made
        self._path = _rendezvous_path()       # unique per (launch, plan): see _rendezvous_path
        _plans_made += 1
        if self.world > 1:
            uid = exchange_unique_id(self.rank, self.world, _hip.DevicePlan.comm_unique_id, path=self._path)
            with self.deadline("RCCL communicator init"):
                self.plan.comm_init(uid)      # collective: every rank calls it; ends with a tiny all-reduce
                self.plan.barrier()
            # the exchange without send / receive kernels, if every rank can map its peers' receive buffers (collective; all ranks
            # get the same answer): DESIGN.md section 5.  ``exchange`` / RANDOMFIELD_EXCHANGE = 'rccl' keeps the grouped send / receive.
            self.exchange = "rccl"
            if mode != "rccl":
                with self.deadline("direct exchange set-up"):
                    try:
                        got = self.plan.enable_direct_exchange(True)
                    except RuntimeError:
                        if mode == "direct":
                            raise
                        got = False
                if got:
                    self.exchange = "direct"
                elif mode == "direct":
                    raise RuntimeError("exchange='direct' was asked for, but not every rank could map its peers' receive buffers "
                                       "(or the grid's y-pass tiles straddle x planes)")
            if self.rank == 0:
                try:
                    os.remove(self._path)
                except OSError:
                    pass
        else:
            self.exchange = "none"

    def deadline(self, what, seconds=None):
        """``with dist.deadline("first exchange"): ...`` -- a watchdog around a step every rank must reach (:class:`Deadline`)."""
        return Deadline(what, self.rank, self.world, seconds, path=self._path)

    def barrier(self):
        self.plan.barrier()

    def allreduce(self, values, op="sum"):
        return self.plan.allreduce(values, op)


class SlabHostPlan(object):
    """What :class:`randomfield_amd.generate.Generator` needs from ``transform.Plan``, for one rank of a multi-GPU job:
    the global ``shape``, this rank's host window of the real-space field (``data_out_padded`` (nx/ranks, ny, nz+2) and
    its ``data_out`` view, aliasing as transform.py:227-235) and the device plan.  There is no host-side k-space array:
    k space lives on the devices, split by kz planes."""

    def __init__(self, dist, dtype=np.complex64):
        self.dist = dist
        self.device = dist.plan
        nx, ny, nz = dist.plan.nx, dist.plan.ny, dist.plan.nz
        self.shape = (nx, ny, nz)
        self.inverse, self.packed, self.overwrite, self.backend = True, True, True, "hip"
        rt = dist.plan.real_dtype
        self.data_out_padded = np.empty((dist.plan.nx_local, ny, nz + 2), rt)
        self.data_out = self.data_out_padded[:, :, :nz]
        # shape-only stand-in for the (nx, ny, nz/2+1) complex array the reference allocates on the host (zero strides:
        # one element of memory); the shape rules that look at it (transform.py:46-60) still apply
        self.data_in = np.lib.stride_tricks.as_strided(np.zeros(1, dtype), shape=(nx, ny, nz // 2 + 1), strides=(0, 0, 0))
        self.nbytes_allocated = self.data_out_padded.nbytes

    def agree_on(self, value):
        """The same integer on every rank (max over ranks; exact below 2**53)."""
        return int(self.dist.allreduce([float(value)], op="max")[0])

    def execute(self):
        """There is no host-side k space to transform on a slab rank (it lives on the devices, split by kz planes):
        use :class:`randomfield_amd.generate.Generator` or the device plan."""
        raise RuntimeError("A slab rank's c2r plan has no host k-space input: drive it through Generator / plan.device.")

    def create_reverse_plan(self, reuse_output=True, overwrite=True):
        """The forward (r2c) plan over the same device plan (transform.py:278-301, as ``Generator`` builds it at
        generate.py:79-80): its input is this rank's window of the real-space field -- ``reuse_output`` shares our
        ``data_out_padded`` memory -- and its output this rank's share of k space, ``(nx, ny, nz/(2 ranks) + 1)`` complex:
        the rank's own kz planes, then the Nyquist plane (:func:`side_array_planes`; :func:`assemble_side_array` puts the
        ranks' shares together).  ``execute()`` is collective: rows on the x slab, the all-to-all in the other direction,
        columns on the kz slab (rf_execute_r2c)."""
        if reuse_output and not overwrite:
            return SlabHostReversePlan(self, self.data_out, padded=False)
        if reuse_output:
            return SlabHostReversePlan(self, self.data_out_padded, padded=True)
        nxl, ny, nz = self.data_out.shape
        return SlabHostReversePlan(self, np.empty((nxl, ny, nz + 2 if overwrite else nz), self.data_out.dtype), padded=bool(overwrite),
                                   owns_input=True)


class SlabHostReversePlan(object):
    """``transform.Plan(inverse=False, packed=True)`` for one rank of a multi-GPU job (see :meth:`SlabHostPlan.create_reverse_plan`)."""

    def __init__(self, forward_of, real_buffer, padded, owns_input=False):
        self.dist, self.device, self.shape = forward_of.dist, forward_of.device, forward_of.shape
        self.inverse, self.packed, self.overwrite, self.backend = False, True, bool(padded), "hip"
        nz = self.shape[2]
        if padded:
            self.data_in_padded = real_buffer
            self.data_in = real_buffer[:, :, :nz]
        else:
            self.data_in = real_buffer
        self._padded = bool(padded)
        # this rank's share of k space (host memory of its own: an x slab of reals and a kz slab of complex numbers have
        # different shapes, so the reference's in-place aliasing of the two sides has no counterpart here)
        self.data_out = np.empty(self.device.k_shape, self.device.complex_dtype)
        self.nbytes_allocated = self.data_out.nbytes + (real_buffer.nbytes if owns_input else 0)

    def execute(self):
        """Collective over the plan's ranks; returns this rank's share of k space (``data_out``)."""
        dev = self.device
        if self._padded:
            dev.upload_real(self.data_in_padded, padded=True)
        else:
            dev.upload_real(np.ascontiguousarray(self.data_in), padded=False)
        dev.execute_r2c()
        dev.download_k(self.data_out)
        return self.data_out
