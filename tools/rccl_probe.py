"""Probe RCCL on the GPU box: (a) our library's rf_comm_init with 1 rank, (b) torch.distributed nccl with 1 rank."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
which = sys.argv[1]
if which == "ours":
    from randomfield_amd import _hip
    p = _hip.DevicePlan(32, 32, 64)
    uid = _hip.DevicePlan.comm_unique_id()
    print("uid ok", len(uid))
    p.comm_init(uid)
    print("comm_init ok")
else:
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("torch nccl ok", t.cpu().numpy())
