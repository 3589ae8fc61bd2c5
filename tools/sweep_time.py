"""fused clear + integrate sweep alone, hipEvent-timed: python tools/sweep_time.py [C2 C4 ...]
(development library + DFA_TSDF_COARSE=0 for the A/B of the four-runs-at-a-time classification)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dynfu_amd as A
from dynfu_amd import synth

A.load()
for name in (sys.argv[1:] or ["C2", "C4"]):
    cfg = synth.CONFIGS[name]
    intr = synth.intrinsics(cfg)
    voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
    dim = cfg["dim"]
    depth = torch.from_numpy(synth.depth_frame(cfg, 3)).cuda()
    dists = torch.empty((cfg["height"], cfg["width"]), dtype=torch.uint16, device="cuda")
    A.compute_dists(depth, dists, *intr)
    vol = torch.empty((dim, dim, dim), dtype=torch.int32, device="cuda")
    occ = A.tsdf_occupancy(vol)
    for label, kw in (("fused", {}), ("fused + occupancy map", dict(occupancy=occ))):
        for _ in range(3):
            A.tsdf_clear_integrate(vol, dists, voxel, trunc, synth.MAX_WEIGHT, vol2cam, *intr, **kw)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for e0, e1 in ev:
            e0.record()
            A.tsdf_clear_integrate(vol, dists, voxel, trunc, synth.MAX_WEIGHT, vol2cam, *intr, **kw)
            e1.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        nbytes = 4.0 * dim ** 3 + 2.0 * cfg["width"] * cfg["height"]
        print("%s %s: %.4f ms  %.2f TB/s  (%.3f of 8 TB/s)  checksum %d" % (name, label, ms, nbytes / ms / 1e9, nbytes / ms / 1e9 / 8, int(vol.view(-1)[::4097].sum())))
