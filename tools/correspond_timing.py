"""hipEvent timing of dfa_correspond (findCorrespondingFrame) on the synthetic canonical / live clouds.
usage: python tools/correspond_timing.py [C2] [reps]"""
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
c = synth.canonical(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
nodes, nw = dev(c["node_pos"]), dev(c["node_w"])
verts, normals = dev(c["verts"]), dev(c["normals"])
idx, w = A.knn(nodes, nw, verts, cfg["k"])
# frame-to-frame motion: the canonical cloud has already been warped by the previous frame's field
# (dyn_fusion.cpp:196), so the live surface is one frame's increment away from it
t_true = synth.true_translations(c["node_pos"], 3, cfg["k"]) - synth.true_translations(c["node_pos"], 2, cfg["k"])
if len(sys.argv) > 3:  # "full": the whole displacement of frame 3 (several cells of the grid)
    t_true = synth.true_translations(c["node_pos"], 3, cfg["k"])
live = dev(synth.live_vertices(c["verts"], idx.cpu().numpy(), w.cpu().numpy(), t_true))
# live cloud in a different order than the canonical one (marching cubes gives no alignment)
perm = torch.randperm(live.shape[0], device="cuda")
live_shuffled = live[perm].contiguous()
for label, lv in (("index-aligned live order", live), ("shuffled live order", live_shuffled)):
    for _ in range(3):
        A.correspond(verts, normals, lv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ov, on, ix = A.correspond(verts, normals, lv)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name} correspond {verts.shape[0]} canonical x {lv.shape[0]} live, {label}: {ms:.3f} ms")
hit = (ix == perm.to(torch.int32)).float().mean().item()
print(f"   live vertex matched to its own canonical source: {100 * hit:.1f} %")
