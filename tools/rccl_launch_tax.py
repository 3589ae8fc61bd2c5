"""What a live RCCL communicator costs the launches of the SAME process (found by the one-rank self-check of bench.py, r05):
the C2 frame — ~45 short launches — takes 0.65 ms without a process group and 0.77 ms with one.  Phases in one process:
  0 no process group | 1 init_process_group(nccl, device_id) | 2 after the first all-reduce | 3 after destroy_process_group
usage: python tools/rccl_launch_tax.py [lazy] [first]
  lazy: no device_id — the communicator is created by the first collective
  first: the process group is created BEFORE the sequence (its tensors, streams and solver plan), as bench.py did at first"""
import datetime
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench


def fps(seq, n=150):
    for f in range(10):
        seq.frame(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(n):
        seq.frame(10 + f)
    torch.cuda.synchronize()
    return round(n / (time.perf_counter() - t0), 1)


lazy, first = "lazy" in sys.argv[1:], "first" in sys.argv[1:]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
kw = dict(rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
if not lazy:
    kw["device_id"] = dev
out = {}
if first:
    dist.init_process_group("nccl", **kw)
    if "touch" in sys.argv[1:]:
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
seq = bench.Sequence("C2", dev)
seq.fuse_first = False
seq.enable_pcg_shadow()
if first:
    out["group_first_then_sequence"] = fps(seq)
    seq2 = bench.Sequence("C2", dev)
    seq2.fuse_first = False
    seq2.enable_pcg_shadow()
    out["second_sequence_same_process"] = fps(seq2)
    del seq2
else:
    out["phase0_no_group"] = fps(seq)
    dist.init_process_group("nccl", **kw)
out["phase1_group_created" + ("_lazy" if lazy else "")] = fps(seq)
t = torch.ones(1, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
out["phase2_after_all_reduce"] = fps(seq)
dist.barrier()
out["phase2b_after_barrier"] = fps(seq)
dist.destroy_process_group()
out["phase3_group_destroyed"] = fps(seq)
out["env"] = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_", "TORCH_NCCL", "HIP_"))}
os.write(2, (json.dumps(out) + "\n").encode())
