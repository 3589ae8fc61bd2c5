#!/usr/bin/env python3
"""PCG iterations per Gauss-Newton iteration, frame by frame (full launch budget): how far the count of one
Gauss-Newton iteration moves between consecutive frames — what a launch budget taken from the previous frames has to
cover.   python tools/ns_iteration_history.py C2 [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pcg = dict(bench.NS_PCG, adaptive_launch=int(os.environ.get("ADAPTIVE", "0")))
seq = bench.Sequence6(name, torch.device("cuda", 0), 64, pcg, n_frames=frames)
hist = []
for f in range(frames):
    seq.frame(f)
    torch.cuda.synchronize()
    st = seq.solver.stats()
    hist.append(st["pcg_it_hist"])
    print(f, st["pcg_it_hist"], "launches", st["pcg_launches"], "short", st["pcg_short"])
import numpy as np
h = np.array([x[:len(hist[0])] for x in hist])
d = np.abs(np.diff(h, axis=0))
print("max |delta| per GN iteration between consecutive frames:", d.max(0).tolist(), "mean:", np.round(d.mean(0), 2).tolist())
