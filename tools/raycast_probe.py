#!/usr/bin/env python3
"""Raycast alone at a BASELINE configuration: fuses one frame, then casts both variants REPS times (the program
rocprofv3 is pointed at for the raycast's kernel statistics and FETCH / WRITE / TCC hit counters), and prints the work
tally + timings as one JSON line.   python tools/raycast_probe.py C2 [reps] [--moved]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import dynfu_amd as A  # noqa: E402
from dynfu_amd import synth  # noqa: E402


class Seq:
    pass


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 20
    A.load()
    cfg = synth.CONFIGS[name]
    seq = Seq()
    seq.A, seq.synth, seq.cfg = A, synth, cfg
    seq.intr = synth.intrinsics(cfg)
    seq.voxel, seq.trunc, seq.vol2cam, seq.cam2vol, seq.rinv = synth.volume_params(cfg)
    dim, W, H = cfg["dim"], cfg["width"], cfg["height"]
    if "--moved" in sys.argv:
        from tests.gpu_util_cpu import aff12, rot
        R = rot([0.1, 1, 0.05], 0.07)
        seq.cam2vol = aff12(R, seq.cam2vol[9:] + np.array([0.03, -0.02, 0.01], np.float32))
        seq.rinv = np.linalg.inv(R).astype(np.float32).reshape(-1)
    depth = torch.from_numpy(synth.depth_frame(cfg, 0)).cuda()
    dists = torch.empty((H, W), dtype=torch.uint16, device="cuda")
    seq.vol = torch.empty((dim, dim, dim), dtype=torch.int32, device="cuda")
    A.compute_dists(depth, dists, *seq.intr)
    A.tsdf_clear_integrate(seq.vol, dists, seq.voxel, seq.trunc, synth.MAX_WEIGHT, seq.vol2cam, *seq.intr)
    out = bench.raycast_probe(seq, name, reps=reps)
    out["config"] = name
    print(json.dumps(out))


if __name__ == "__main__":
    main()
