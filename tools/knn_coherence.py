"""dev tool: k-NN time for queries in caller order (Fibonacci spiral: spatially incoherent) vs the
same queries sorted by a coarse spatial key."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
c = synth.canonical(cfg)
k = cfg["k"]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
nodes, node_w = dev(c["node_pos"]), dev(c["node_w"])
v = c["verts"]
key = np.floor((v - v.min(0)) / 0.06).astype(np.int64)
order = np.lexsort((key[:, 0], key[:, 1], key[:, 2]))
for name, q in (("caller order", v), ("cell-sorted", v[order])):
    q = dev(q)
    for _ in range(3):
        A.knn(nodes, node_w, q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        A.knn(nodes, node_w, q, k)
    e1.record()
    torch.cuda.synchronize()
    print("%-14s %.1f us per call (grid build included)" % (name, e0.elapsed_time(e1) / 20 * 1e3))
