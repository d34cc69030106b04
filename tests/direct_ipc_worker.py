"""One rank of the two-process test of the direct exchange's IPC hand-off
(tests/test_gpu_parity.py::test_direct_exchange_between_processes_through_ipc_handles).

usage: python direct_ipc_worker.py RANK WORLD DIR c64|c128 NX NY NZ CHUNKS

Every rank is a process of its own on cuda:0, a communicator-less rank of a WORLD-rank plan.  The ranks swap their IPC records through
files in DIR (rf_slab_direct_export / rf_slab_direct_import: the transport is the caller's), then run forward -- the y pass stores
into the OTHER process's receive buffer through the mapping --, a barrier, and the z pass; each leaves its x slab of the field in
DIR/field_RANK.npy for the test to compare with the one-rank field.  Two realisations: the second finds the buffers already in use.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

SPACING = 2.5
SEEDS = (11, 12)


def publish(d, name, data):
    tmp = os.path.join(d, name + ".tmp")
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, os.path.join(d, name))


def wait_for(d, names, timeout=180.0):
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, n)) for n in names):
        if time.time() - t0 > timeout:
            raise RuntimeError("timed out waiting for {0}".format(names))
        time.sleep(0.01)


def barrier(d, tag, rank, world):
    publish(d, "bar_{0}_{1}".format(tag, rank), b"1")
    wait_for(d, ["bar_{0}_{1}".format(tag, r) for r in range(world)])


def main():
    rank, world, d, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    nx, ny, nz, chunks = (int(v) for v in sys.argv[5:9])
    from conftest import golden
    from oracle import cpu_ref
    from randomfield_amd import _hip, powertools
    pw = golden("default_power.npz")
    ct = np.complex64 if kind == "c64" else np.complex128
    p = _hip.DevicePlan(nx, ny, nz, ct, nranks=world, rank=rank)
    p.set_kgrid(*powertools.ksq_axes(nx, ny, nz, SPACING))
    p.set_power(*cpu_ref.sigma_table(pw["k"], pw["Pk"], nx, ny, nz, SPACING))
    p.set_exchange_chunks(chunks)
    publish(d, "rec_{0}".format(rank), p.slab_direct_export())
    wait_for(d, ["rec_{0}".format(r) for r in range(world)])
    records = [open(os.path.join(d, "rec_{0}".format(r)), "rb").read() for r in range(world)]
    on = p.slab_direct_import(records)
    publish(d, "on_{0}".format(rank), b"1" if on else b"0")
    barrier(d, "mapped", rank, world)
    if not on:
        return 3
    out = []
    for i, seed in enumerate(SEEDS):
        p.slab_forward(seed=seed)
        p.sync()
        barrier(d, "stored{0}".format(i), rank, world)       # every rank's tiles have landed in every receive buffer
        p.slab_backward()
        out.append(p.download_real())
        barrier(d, "read{0}".format(i), rank, world)         # nobody stores realisation i + 1 into a buffer still being read
    np.save(os.path.join(d, "field_{0}.npy".format(rank)), np.stack(out))
    s1, s2 = p.slab_stats()
    publish(d, "stats_{0}".format(rank), np.array([s1, s2]).tobytes())
    barrier(d, "done", rank, world)                          # keep the buffers mapped until every rank has finished
    p.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
