"""world_size-2 and world_size-4 gloo tests (CPU): the multi-GPU path's rendezvous, unique-id hand-off and, above all,
the slab / all-to-all block layout.  Each rank computes its kz slab with the oracle, runs the x and y
inverse transforms locally, exchanges the blocks defined in randomfield_amd/slab.py with a real
all_to_all over gloo, gathers its rows and runs the z c2r -- and must end up with exactly its x slab of
the single-process field."""
import os
import socket
import sys

import numpy as np
import pytest

# torch is imported lazily inside the tests: a `-m gpu` run must not load PyTorch's bundled ROCm runtime
# into the process next to the system one (RCCL then fails to initialise)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, shape, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import cpu_ref
    from randomfield_amd import slab
    d, r, w, lr = slab.init_process_group()
    assert (r, w) == (rank, world)
    # unique-id hand-off (a stand-in payload: no RCCL on the CPU)
    uid = slab.broadcast_bytes(d, bytes(range(128)) if rank == 0 else None, src=0)
    assert uid == bytes(range(128))

    nx, ny, nz = shape
    lay = slab.slab_layout(nx, ny, nz, world, rank)
    pw = np.load(os.path.join(ROOT, "tests", "golden", "default_power.npz"))
    noise = cpu_ref.reference_noise(11, nx * ny * (nz // 2 + 1))
    full_k = cpu_ref.generate_kspace(nx, ny, nz, 2.5, pw["k"], pw["Pk"], noise=noise, dtype=np.complex128)
    nzc = nz // 2
    packed = full_k[:, :, :nzc].copy()
    packed[:, :, 0] = full_k[:, :, 0] + 1j * full_k[:, :, nzc]          # device-internal packing of DC/Nyquist planes
    mine = packed[:, :, lay["kz0"]:lay["kz0"] + lay["nzl"]]             # this rank's kz slab

    def pipeline(mine):
        mine = np.fft.ifft(np.fft.ifft(mine, axis=0), axis=1) * (nx * ny)   # x and y passes (unnormalised)
        # THE exchange: grouped point-to-point sends/receives, block h -> rank h (what the library does with
        # ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd; the own block is a local copy)
        send = [torch.from_numpy(np.ascontiguousarray(b).view(np.float64)) for b in slab.exchange_blocks(mine, world)]
        recv = [torch.empty_like(send[0]) for _ in range(world)]
        recv[rank].copy_(send[rank])
        ops = []
        for h in range(world):
            if h != rank:
                ops.append(dist.P2POp(dist.isend, send[h], h))
                ops.append(dist.P2POp(dist.irecv, recv[h], h))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        rows = slab.gather_rows([t.numpy().view(np.complex128) for t in recv])   # [nxl][ny][nz/2]
        assert rows.shape == (lay["nxl"], ny, nzc)
        half = np.empty((lay["nxl"], ny, nzc + 1), np.complex128)            # unpack slot 0 -> DC and Nyquist elements
        half[:, :, :nzc] = rows
        half[:, :, 0] = rows[:, :, 0].real
        half[:, :, nzc] = rows[:, :, 0].imag
        return np.fft.irfft(half, n=nz, axis=2) / (nx * ny)

    delta = pipeline(mine)
    # the reference's default call + Newtonian potential on slab ranks (generate.py:200-217, 333-343): every rank keeps its
    # planes of delta(k)/k**2 -- own planes, then the Nyquist plane (slab.side_array_planes) -- and later transforms
    # scale * potential from that share alone
    share = slab.split_side_array(cpu_ref.potential_kspace(full_k, 2.5), world, rank)
    assert share.shape == (nx, ny, lay["nzl"] + 1)
    again = share[:, :, :lay["nzl"]].copy()
    if rank == 0:
        again[:, :, 0] = share[:, :, 0] + 1j * share[:, :, lay["nzl"]]     # rank 0 packs the Nyquist plane into slot kz = 0
    phi = pipeline(-1.5 * again)
    # global rms through a 2-double all-reduce
    st = torch.tensor([delta.sum(), (delta ** 2).sum()], dtype=torch.float64)
    dist.all_reduce(st)
    # the forward transform on slab ranks (rf_execute_r2c, transform.py:278-301): rows on the x slab, the same blocks the other
    # way (block g of the x-slab rank -> kz-slab rank g), columns on the kz slab, slot kz = 0 untangled into the two real planes
    nzl, nxl = lay["nzl"], lay["nxl"]
    rows = np.fft.rfft(delta, axis=2)
    pk = rows[:, :, :nzc].copy()
    pk[:, :, 0] = rows[:, :, 0].real + 1j * rows[:, :, nzc].real
    send = [torch.from_numpy(np.ascontiguousarray(pk[:, :, g * nzl:(g + 1) * nzl]).view(np.float64)) for g in range(world)]
    recv = [torch.empty_like(send[0]) for _ in range(world)]
    recv[rank].copy_(send[rank])
    ops = []
    for h in range(world):
        if h != rank:
            ops.append(dist.P2POp(dist.isend, send[h], h))
            ops.append(dist.P2POp(dist.irecv, recv[h], h))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    cols = np.fft.fft(np.fft.fft(np.concatenate([t.numpy().view(np.complex128) for t in recv], axis=0), axis=1), axis=0)
    kshare = np.zeros((nx, ny, nzl + 1), np.complex128)
    kshare[:, :, :nzl] = cols
    if rank == 0:
        c = cols[:, :, 0]
        cm = np.conj(np.roll(c[::-1, ::-1], (1, 1), axis=(0, 1)))          # conj C(-kx, -ky)
        kshare[:, :, 0] = 0.5 * (c + cm)
        kshare[:, :, nzl] = (c - cm) / 2j
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), delta=delta, stats=st.numpy(), x0=lay["x0"], phi=phi, share=share, kshare=kshare)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (8, 8, 16)), (4, (16, 8, 32))])
def test_slab_exchange_processes(tmp_path, world, shape):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from oracle import cpu_ref
    mp.spawn(_worker, args=(world, _free_port(), shape, str(tmp_path)), nprocs=world, join=True)
    nx, ny, nz = shape
    pw = np.load(os.path.join(ROOT, "tests", "golden", "default_power.npz"))
    noise = cpu_ref.reference_noise(11, nx * ny * (nz // 2 + 1))
    ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, 2.5, pw["k"], pw["Pk"], noise=noise, dtype=np.complex128)
    from randomfield_amd import slab
    pot = cpu_ref.potential_kspace(cpu_ref.generate_kspace(nx, ny, nz, 2.5, pw["k"], pw["Pk"], noise=noise, dtype=np.complex128), 2.5)
    phi_ref = np.fft.irfftn(-1.5 * pot, s=shape, axes=(0, 1, 2))
    shares = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))["share"] for r in range(world)]
    assert np.array_equal(slab.assemble_side_array(shares), pot)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        x0 = int(g["x0"])
        assert np.allclose(g["delta"], ref[x0:x0 + nx // world], rtol=0, atol=1e-12 * rms)
        assert np.allclose(g["phi"], phi_ref[x0:x0 + nx // world], rtol=0, atol=1e-12 * phi_ref.std())
        n = ref.size
        assert abs(np.sqrt(g["stats"][1] / n - (g["stats"][0] / n) ** 2) - rms) < 1e-12 * rms
    fwd = np.fft.rfftn(ref, axes=(0, 1, 2))
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))["kshare"]
        want = slab.split_side_array(fwd, world, r)
        keep = slice(None) if r == 0 else slice(0, want.shape[2] - 1)          # (only rank 0 fills the Nyquist slot)
        assert np.allclose(got[:, :, keep], want[:, :, keep], rtol=0, atol=1e-10 * np.abs(fwd).max())


def _share_worker(rank, world, port, shape, counts, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import cpu_ref
    from randomfield_amd import slab
    d, r, w, lr = slab.init_process_group()
    nx, ny, nz = shape
    # what the GPU path gathers with an integer all-reduce: every rank's counts, zero outside its own segments
    nseg = len(counts)
    mine = torch.zeros(nseg, dtype=torch.int64)
    a, b = rank * nseg // world, (rank + 1) * nseg // world
    mine[a:b] = torch.tensor(counts[a:b], dtype=torch.int64)
    dist.all_reduce(mine)
    assert mine.tolist() == list(counts)
    lay = slab.shared_replay_layout(mine.numpy(), nx, ny, nz, world)
    # this rank "replays" its share: cells [cell_begin[rank], cell_begin[rank + 1]) of numpy's stream
    stream = cpu_ref.reference_noise(11, nx * ny * (nz // 2 + 1)).reshape(-1, 2)
    c0, c1 = lay["cell_begin"][rank], lay["cell_begin"][rank + 1]
    send = slab.shared_replay_pack(stream[c0:c1], c0, nz, world)
    assert [len(x) for x in send] == lay["sendcnt"][rank]
    recv = np.full((lay["stream_pairs"], 2), np.nan)
    ops, keep = [], []
    for h in range(world):
        n = lay["sendcnt"][h][rank]                 # what rank h sends to me
        lo = lay["recvoff"][rank][h]
        if h == rank:
            recv[lo:lo + n] = send[rank]
            continue
        if len(send[h]):
            keep.append(torch.from_numpy(np.ascontiguousarray(send[h])))
            ops.append(dist.P2POp(dist.isend, keep[-1], h))
        if n:
            buf = torch.empty((n, 2), dtype=torch.float64)
            keep.append((buf, lo, n))
            ops.append(dist.P2POp(dist.irecv, buf, h))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for item in keep:
        if isinstance(item, tuple):
            recv[item[1]:item[1] + item[2]] = item[0].numpy()
    np.save(os.path.join(out_dir, "share%d.npy" % rank), recv)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,nseg", [(2, (8, 8, 16), 5), (4, (4, 6, 32), 11)])
def test_shared_replay_exchange_processes(tmp_path, world, shape, nseg):
    """The shared replay of the reference's stream (rf_mt_share_*): segments dealt to ranks, counts all-reduced, pairs packed by
    destination and exchanged point to point over gloo; every rank ends up with its planes (+ the Nyquist plane) of
    RandomState(seed).normal, i.e. its side array of the single-process stream."""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from oracle import cpu_ref
    from randomfield_amd import slab
    nx, ny, nz = shape
    ncells = nx * ny * (nz // 2 + 1)
    rng = np.random.RandomState(world)
    cuts = np.sort(rng.choice(np.arange(1, ncells + 40), nseg - 1, replace=False))       # uneven segments, a tail nobody needs
    counts = np.diff(np.concatenate([[0], cuts, [ncells + 40]])).tolist()
    mp.spawn(_share_worker, args=(world, _free_port(), shape, counts, str(tmp_path)), nprocs=world, join=True)
    full = cpu_ref.reference_noise(11, ncells).reshape(nx, ny, nz // 2 + 1, 2)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "share%d.npy" % r)).reshape(nx, ny, nz // 2 // world + 1, 2)
        assert np.array_equal(got, slab.split_side_array(full, world, r))


def test_shared_replay_layout_rules():
    from randomfield_amd import slab
    nx, ny, nz, P = 4, 4, 16, 4
    ncells = nx * ny * 9
    lay = slab.shared_replay_layout([50, 30, 40, 44], nx, ny, nz, P)
    assert lay["cell_begin"] == [0, 50, 80, 120, ncells] and lay["seg_begin"] == [0, 1, 2, 3, 4]
    for q in range(P):          # every stream is complete and contiguous: rank r's pairs start where rank r - 1's ended
        ends = [lay["recvoff"][q][r] + lay["sendcnt"][r][q] for r in range(P)]
        assert lay["recvoff"][q][0] == 0 and ends[:-1] == lay["recvoff"][q][1:] and ends[-1] == lay["stream_pairs"]
    with pytest.raises(ValueError):
        slab.shared_replay_layout([10, 10], nx, ny, nz, 2)
    # fewer segments than ranks: the empty-handed ranks send nothing
    lay = slab.shared_replay_layout([ncells + 3], nx, ny, nz, 2)
    assert lay["cell_begin"] == [0, 0, ncells] and sum(lay["sendcnt"][0]) == 0


def test_unique_id_file_handoff(tmp_path, monkeypatch):
    """The torch-free unique-id hand-off used on the GPU path: rank 0 writes, the others read."""
    import threading
    from randomfield_amd import slab
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "12345")
    payload = bytes(range(128))
    got = {}

    def reader(r):
        got[r] = slab.exchange_unique_id(r, 3, lambda: b"wrong" * 30, timeout=20)

    threads = [threading.Thread(target=reader, args=(r,)) for r in (1, 2)]
    for t in threads:
        t.start()
    assert slab.exchange_unique_id(0, 3, lambda: payload) == payload
    for t in threads:
        t.join()
    assert got == {1: payload, 2: payload}
    assert slab.exchange_unique_id(0, 1, lambda: payload) == payload


def test_slab_layout_rules():
    from randomfield_amd import slab
    lay = slab.slab_layout(2048, 2048, 2048, 8, 3)
    assert lay["nxl"] == 256 and lay["nzl"] == 128 and lay["kz0"] == 384 and lay["x0"] == 768
    assert lay["block_elems"] * 8 == 256 * 2048 * 128 * 8 == 536870912          # 537 MB per pair
    assert lay["local_elems"] == 2048 * 2048 * 128
    with pytest.raises(ValueError):
        slab.slab_layout(16, 16, 16, 3, 0)
    with pytest.raises(ValueError):
        slab.slab_layout(16, 16, 16, 8, 0)           # nz/2 = 8 planes over 8 ranks: 1 plane each (odd)


def test_rendezvous_files_are_per_plan_and_stale_ids_are_dropped(tmp_path, monkeypatch):
    """Two DistributedPlans of one job never share a rendezvous file (per-process plan counter), and rank 0 removes a stale
    file of the same name before it writes (a fast rank must not pick up a previous plan's id)."""
    from randomfield_amd import slab
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "23456")
    a, b = slab._rendezvous_path(0), slab._rendezvous_path(1)
    assert a != b and a.endswith("_0") and b.endswith("_1")
    stale = bytes(128)
    with open(a, "wb") as f:
        f.write(stale)
    fresh = bytes(range(128))
    assert slab.exchange_unique_id(0, 2, lambda: fresh, path=a) == fresh
    assert slab.exchange_unique_id(1, 2, lambda: b"", timeout=5, path=a) == fresh


def test_a_rank_never_accepts_another_launch_or_an_older_file(tmp_path, monkeypatch):
    """What exchange_unique_id promises its non-zero ranks: a file of this name is the id of THIS launch only if it carries the
    launch's nonce and was written after the process started.  A crashed launch's file (other nonce), an untagged file and a
    file of this very name left by an earlier job of the same long-lived launcher (same nonce, old mtime) are all ignored until
    rank 0 replaces them."""
    import threading
    import time
    from randomfield_amd import slab
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "34567")
    # the nonce: the launcher's own when it sets one, else (parent pid, parent start, port, run id) -- a new launcher process
    # or port is a new name, so files of other launches are not even looked at
    n0 = slab.launch_nonce()
    assert n0.startswith("%d." % os.getppid()) and ".34567." in n0
    monkeypatch.setenv("MASTER_PORT", "34568")
    assert slab.launch_nonce() != n0
    monkeypatch.setenv("RANDOMFIELD_LAUNCH_NONCE", "job-17/a b")
    assert slab.launch_nonce() == "job-17_a_b" and "job-17_a_b" in slab._rendezvous_path(3)
    path = slab._rendezvous_path(0)
    dead, fresh = bytes([7]) * 128, bytes(range(128))
    got = {}

    def reader():
        got["uid"] = slab.exchange_unique_id(1, 2, lambda: b"", timeout=20, path=path)

    def run_with(stale_bytes, age):
        with open(path, "wb") as f:
            f.write(stale_bytes)
        if age:
            os.utime(path, (time.time() - age, time.time() - age))
        got.clear()
        t = threading.Thread(target=reader)
        t.start()
        time.sleep(0.3)
        assert t.is_alive() and not got, "the stale file was accepted"
        assert slab.exchange_unique_id(0, 2, lambda: fresh, path=path) == fresh
        t.join(10)
        assert got == {"uid": fresh}

    run_with(dead, 0)                                                # no tag at all (the pre-round-4 format)
    run_with(dead + b"|some-other-launch", 0)                        # another launch's nonce
    # this name and nonce, but written an hour before we started: only a launch WITHOUT a launcher-made nonce can meet that (one
    # long-lived parent starting job after job on one port), and only there does the mtime test apply
    monkeypatch.delenv("RANDOMFIELD_LAUNCH_NONCE")
    path = slab._rendezvous_path(0)
    run_with(dead + b"|" + slab.launch_nonce().encode(), 3600.0)
    # with a launcher-made nonce (unique per launch) the tag alone decides: ranks started long after rank 0 wrote the id, or a
    # TMPDIR whose clock is behind, must not wait for the timeout
    monkeypatch.setenv("RANDOMFIELD_LAUNCH_NONCE", "job-18")
    path = slab._rendezvous_path(0)
    with open(path, "wb") as f:
        f.write(fresh + b"|job-18")
    os.utime(path, (time.time() - 3600.0, time.time() - 3600.0))
    assert slab.exchange_unique_id(1, 2, lambda: b"", timeout=5, path=path) == fresh
    # (the guard is the process's own start time: a file written after it is taken at once)
    assert slab._process_start_time() <= time.time() and time.time() - slab._process_start_time() < 3600
    with pytest.raises(ValueError):
        slab.exchange_unique_id(0, 2, lambda: b"short", path=path)


def test_deadline_names_the_missing_rank_and_exits(tmp_path):
    """A blocked collective step ends the process with status 3 and says which rank never arrived (run in a child:
    the watchdog calls os._exit)."""
    import subprocess
    import sys
    code = (
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "os.environ['TMPDIR'] = %r\n"
        "from randomfield_amd import slab\n"
        "open(slab._rendezvous_path(0) + '.first_exchange.rank0', 'w').close()   # rank 0 got there; rank 2 never does\n"
        "with slab.Deadline('first exchange', 1, 3, seconds=1.0, path=slab._rendezvous_path(0)):\n"
        "    time.sleep(30)\n"
        "print('not reached')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "first exchange" in r.stderr and "[2]" in r.stderr and "not reached" not in r.stdout
    # a step that finishes in time: the watchdog is cancelled and the check-in file removed
    from randomfield_amd import slab
    os.environ["TMPDIR"], old = str(tmp_path), os.environ.get("TMPDIR")
    try:
        with slab.Deadline("quick step", 0, 2, seconds=5.0, path=slab._rendezvous_path(7)) as d:
            assert os.path.exists(d.base + ".rank0")
        assert not os.path.exists(d.base + ".rank0")
    finally:
        if old is None:
            del os.environ["TMPDIR"]
        else:
            os.environ["TMPDIR"] = old


def _direct_worker(rank, world, port, shape, chunks, out_dir):
    """The DIRECT exchange between processes (rf_comm_enable_direct), modelled on the CPU: every rank's receive buffer lives in a
    shared-memory segment (the stand-in for a hipMalloc'ed buffer + hipIpcGetMemHandle), its NAME is the handle that one all-gather
    hands to every rank, every rank maps its peers' segments (hipIpcOpenMemHandle), proves the mapping with markers, and then its
    "y pass" stores every output tile at  local cell offset + slab.direct_store_bases()[chunk][destination]  in the destination's
    buffer -- the arithmetic of rf_capi.hip rebuild_peer_tab.  One barrier, then the gathering z pass reads [source][chunk][nxl][ny][nzl/C]."""
    import torch.distributed as dist
    from multiprocessing import shared_memory
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import cpu_ref
    from randomfield_amd import slab
    d, r, w, lr = slab.init_process_group()
    nx, ny, nz = shape
    lay = slab.slab_layout(nx, ny, nz, world, rank)
    nxl, nzl, nzc = lay["nxl"], lay["nzl"], nz // 2
    nzs = nzl // chunks
    cells = nx * ny * nzl                                     # == nxl * ny * nzc: the receive buffer is as large as the local array
    # 1. the receive buffer and its handle
    seg = shared_memory.SharedMemory(create=True, size=cells * 16)
    mine_R = np.ndarray((cells,), np.complex128, buffer=seg.buf)
    mine_R[:] = np.nan
    handles = [None] * world
    dist.all_gather_object(handles, seg.name)
    # 2. map the peers' buffers
    peers, views = [], []
    for h in range(world):
        if h == rank:
            peers.append(seg)
            views.append(mine_R)
        else:
            s = shared_memory.SharedMemory(name=handles[h])
            peers.append(s)
            views.append(np.ndarray((cells,), np.complex128, buffer=s.buf))
    # 3. prove it: marker (1000 + me) into slot `me` of every buffer, barrier, every rank finds all markers in its own
    for h in range(world):
        views[h][rank] = 1000 + rank
    dist.barrier()
    assert [mine_R[g].real for g in range(world)] == [1000 + g for g in range(world)]
    dist.barrier()
    mine_R[:] = np.nan
    dist.barrier()
    # the rank's kz slab after the x and y passes, sub-slab by sub-slab: local array [chunk][nx][ny][nzl / C]
    pw = np.load(os.path.join(ROOT, "tests", "golden", "default_power.npz"))
    noise = cpu_ref.reference_noise(11, nx * ny * (nz // 2 + 1))
    full_k = cpu_ref.generate_kspace(nx, ny, nz, 2.5, pw["k"], pw["Pk"], noise=noise, dtype=np.complex128)
    packed = full_k[:, :, :nzc].copy()
    packed[:, :, 0] = full_k[:, :, 0] + 1j * full_k[:, :, nzc]
    base = slab.direct_store_bases(nx, ny, nz, world, rank, chunks)
    for c in range(chunks):
        kz0 = lay["kz0"] + c * nzs
        sub = np.fft.ifft(np.fft.ifft(packed[:, :, kz0:kz0 + nzs], axis=0), axis=1) * (nx * ny)      # [nx][ny][nzs], x and y passes
        # THE stores of the y pass: a tile = all ny rows of some kz columns of ONE ix; destination h = ix // nxl
        for ix in range(nx):
            h = ix // nxl
            off = (ix * ny + np.arange(ny)[:, None]) * nzs + np.arange(nzs)[None, :]               # local cell offsets of the plane's rows
            views[h][base[c][h] + off] = sub[ix]
    dist.barrier()                                            # the tiny all-reduce: every rank's stores have landed
    assert not np.isnan(mine_R).any()                         # every cell of the receive buffer was written exactly by its owner
    rows = mine_R.reshape(world, chunks, nxl, ny, nzs).transpose(2, 3, 0, 1, 4).reshape(nxl, ny, nzc)
    half = np.empty((nxl, ny, nzc + 1), np.complex128)
    half[:, :, :nzc] = rows
    half[:, :, 0] = rows[:, :, 0].real
    half[:, :, nzc] = rows[:, :, 0].imag
    delta = np.fft.irfft(half, n=nz, axis=2) / (nx * ny)
    np.save(os.path.join(out_dir, "direct%d.npy" % rank), delta)
    dist.barrier()                                            # nobody unmaps while a peer still reads
    del views, mine_R
    for h, s in enumerate(peers):
        s.close()
    dist.barrier()
    seg.unlink()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,chunks", [(2, (8, 8, 16), 1), (2, (8, 4, 32), 2), (4, (16, 8, 64), 2)])
def test_direct_exchange_processes(tmp_path, world, shape, chunks):
    """The hand-off of the direct exchange between PROCESSES: handles gathered, peers' buffers mapped, markers, stores at
    local offset + per-destination base, one barrier -- and every rank's x slab of the field comes out."""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from oracle import cpu_ref
    mp.spawn(_direct_worker, args=(world, _free_port(), shape, chunks, str(tmp_path)), nprocs=world, join=True)
    nx, ny, nz = shape
    pw = np.load(os.path.join(ROOT, "tests", "golden", "default_power.npz"))
    noise = cpu_ref.reference_noise(11, nx * ny * (nz // 2 + 1))
    ref, rms = cpu_ref.generate_delta_field(nx, ny, nz, 2.5, pw["k"], pw["Pk"], noise=noise, dtype=np.complex128)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "direct%d.npy" % r))
        x0 = r * (nx // world)
        assert np.allclose(got, ref[x0:x0 + nx // world], rtol=0, atol=1e-12 * rms)


def test_direct_store_bases_and_settings(monkeypatch):
    from randomfield_amd import slab
    # rank g, destination h, chunk c: base = (g C + c - h) * block cells; a cell of block h lands in segment (g, c) of rank h
    b = slab.direct_store_bases(16, 8, 64, 4, 2, chunks=2)
    blk = 4 * 8 * 4
    assert b[1][3] == (2 * 2 + 1 - 3) * blk and b[0][0] == 4 * blk and len(b) == 2 and len(b[0]) == 4
    for raw, want in ((None, "auto"), ("auto", "auto"), ("", "auto"), ("4", 4), (2, 2)):
        monkeypatch.delenv("RANDOMFIELD_EXCHANGE_CHUNKS", raising=False)
        if isinstance(raw, str):
            monkeypatch.setenv("RANDOMFIELD_EXCHANGE_CHUNKS", raw)
            assert slab.exchange_chunks_setting() == want
        else:
            assert slab.exchange_chunks_setting(raw) == want
    monkeypatch.setenv("RANDOMFIELD_EXCHANGE_CHUNKS", "many")
    with pytest.raises(ValueError):
        slab.exchange_chunks_setting()
    with pytest.raises(ValueError):
        slab.exchange_chunks_setting(0)
    assert slab.exchange_chunks_setting(8) == 8                  # (an explicit argument wins over the environment)
    monkeypatch.setenv("RANDOMFIELD_EXCHANGE", "RCCL")
    assert slab.exchange_mode_setting() == "rccl" and slab.exchange_mode_setting("direct") == "direct"
    with pytest.raises(ValueError):
        slab.exchange_mode_setting("nvlink")
