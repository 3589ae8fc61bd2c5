"""hipEvent timing of dfa_marching_cubes on the integrated synthetic depth frame.
usage: python tools/mc_timing.py [C2] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dynfu_amd as A
from dynfu_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
A.load()
cfg = synth.CONFIGS[name]
fx, fy, cx, cy = synth.intrinsics(cfg)
voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
depth = torch.from_numpy(synth.depth_frame(cfg, 0)).cuda()
dists = torch.empty(depth.shape, dtype=torch.uint16, device="cuda")
A.compute_dists(depth, dists, fx, fy, cx, cy)
dim = cfg["dim"]
vol = torch.empty((dim, dim, dim), dtype=torch.int32, device="cuda")
A.tsdf_clear_integrate(vol, dists, voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy)
tri, nv = A.mc_default_tables()
tri, nv = torch.from_numpy(tri).cuda(), torch.from_numpy(nv).cuda()
_, total = A.marching_cubes(vol, voxel, tri, nv, 0)
total = int(total.item())
for label, cap in (("count only", 0), ("count + emit", total)):
    for _ in range(3):
        A.marching_cubes(vol, voxel, tri, nv, cap)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        A.marching_cubes(vol, voxel, tri, nv, cap)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = 4.0 * dim ** 3 / 1e9
    print(f"{name} marching cubes {dim}^3, {label}: {ms:.3f} ms  ({gb / ms * 1e3:.0f} GB/s of volume read), "
          f"{total} vertices = {total // 3} triangles")
