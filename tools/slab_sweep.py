#!/usr/bin/env python3
"""GPU sweep of the multi-GPU decomposition with virtual ranks on one device: exchange mode (kz slabs + all-to-all by
device copies) and replicated-generation mode, P = 2, 4, 8, float32 and float64, against the single-rank field."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randomfield_amd import _hip, powertools      # noqa: E402

POWER = powertools.load_default_power()


def plan(shape, ct, nranks=1, rank=0):
    nx, ny, nz = shape
    p = _hip.DevicePlan(nx, ny, nz, ct, nranks=nranks, rank=rank)
    p.set_kgrid(*powertools.ksq_axes(nx, ny, nz, 2.5))
    p.set_power(*powertools.sigma_table(POWER, shape, 2.5))
    return p


def main():
    shapes = [(16, 16, 64), (64, 8, 128), (8, 64, 256), (256, 32, 64), (32, 256, 128), (512, 16, 64), (16, 512, 64),
              (1024, 8, 64), (8, 1024, 128), (2048, 8, 64), (8, 2048, 64), (16, 16, 2048), (128, 128, 128), (64, 64, 1024)]
    n = 0
    for shape in shapes:
        nx, ny, nz = shape
        for ct in (np.complex64, np.complex128):
            one = plan(shape, ct)
            one.realise(seed=21)
            ref = one.download_real()
            std = one.moments()[1]
            one.close()
            for P in (2, 4, 8):
                if nx % P or (nz // 2) % P or (nz // 2 // P) % 2:
                    continue
                for mode in ("exchange", "replicate"):
                    try:
                        ranks = [plan(shape, ct, P, r) for r in range(P)]
                    except RuntimeError as exc:
                        print("   skipped", shape, np.dtype(ct).name, P, str(exc)[-80:])
                        break
                    if mode == "replicate":
                        try:
                            for p in ranks:
                                p.set_replicated_generation(True)
                                p.realise(seed=21)
                        except RuntimeError as exc:        # e.g. slab boundary not a multiple of the last-pass stride
                            print("   replicate not available", shape, np.dtype(ct).name, P, str(exc)[-70:])
                            for p in ranks:
                                p.close()
                            continue
                    else:
                        for p in ranks:
                            p.slab_forward(seed=21)
                        _hip.DevicePlan.slab_exchange_local(ranks)
                        for p in ranks:
                            p.slab_backward()
                    field = np.concatenate([p.download_real() for p in ranks], axis=0)
                    for p in ranks:
                        p.close()
                    err = np.max(np.abs(field - ref)) / std
                    assert err <= 2e-6, (shape, np.dtype(ct).name, P, mode, err)
                    n += 1
        print("ok", shape, flush=True)
    print("slab sweep ok: %d decompositions" % n)


if __name__ == "__main__":
    main()
