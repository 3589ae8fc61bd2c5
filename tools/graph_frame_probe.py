#!/usr/bin/env python3
"""Does the whole C2 frame (graph build + solve + fused sweep on two streams) replay as ONE HIP graph, and what does the
frame cost then?  Eager frames (a ctypes call per launch: host-paced) against torch.cuda.CUDAGraph replays of the same
launches with the frame's inputs copied into static buffers first.   python tools/graph_frame_probe.py [C2] [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
seq = bench.Sequence(name, dev, n_frames=16)
seq.fuse_first = False
seq.enable_pcg_shadow()


def run(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(n):
        fn(f)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


for f in range(10):
    seq.frame(f)
print("eager: %.1f frames/s" % run(lambda f: seq.frame(10 + f), K))
err = float((seq.solver.translations() - seq.t_true[(10 + K - 1) % seq.n_frames]).abs().max())
print("  max |t - t*| %.2e" % err)

# static inputs
depth_in = torch.empty_like(seq.depth[0])
live_in = torch.empty_like(seq.live[0])
seq.depth = depth_in[None]
live_all = seq.live
seq.live = live_in[None]
n_frames = seq.n_frames
seq.n_frames = 1
depth_all = torch.from_numpy(__import__("numpy").stack(seq.depth_np)).to(dev)
depth_in.copy_(depth_all[0]), live_in.copy_(live_all[0])
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        seq.frame(0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    seq.frame(0)
torch.cuda.synchronize()


def replay(f):
    depth_in.copy_(depth_all[f % n_frames], non_blocking=True)
    live_in.copy_(live_all[f % n_frames], non_blocking=True)
    g.replay()


for f in range(5):
    replay(f)
print("graph replay: %.1f frames/s" % run(lambda f: replay(10 + f), K))
err = float((seq.solver.translations() - seq.t_true[(10 + K - 1) % n_frames]).abs().max())
print("  max |t - t*| %.2e ; stats %s" % (err, {k: seq.solver.stats()[k] for k in ("gn_iters", "pcg_iters", "gn_noop")}))
