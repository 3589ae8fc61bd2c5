"""k-NN time against the ORDER of the queries (same set): the synthetic Fibonacci order (64 consecutive queries form a
ring), sorted by a 32^3 cell index, and shuffled.  usage: python tools/knn_order_timing.py C2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
A.load()
cfg = synth.CONFIGS[name]; k = cfg["k"]; c = synth.canonical(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
nodes, nw = dev(c["node_pos"]), dev(c["node_w"])
v = c["verts"]
lo, hi = v.min(0), v.max(0)
cell = np.minimum(((v - lo) / (hi - lo).max() * 32).astype(np.int64), 31)
key = cell[:, 0] + 32 * (cell[:, 1] + 32 * cell[:, 2])
orders = {"fibonacci": np.arange(len(v)), "cell-sorted": np.argsort(key, kind="stable"),
          "shuffled": np.random.default_rng(0).permutation(len(v))}
ref = None
for nm, o in orders.items():
    q = dev(v[o])
    for _ in range(3):
        idx, w = A.knn(nodes, nw, q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        idx, w = A.knn(nodes, nw, q, k)
    e1.record(); torch.cuda.synchronize()
    back = np.empty_like(o); back[o] = np.arange(len(o))
    got = idx.cpu().numpy()[back]
    if ref is None: ref = got
    print("%-12s %8.1f us per call (grid build included)   same lists as the first order: %s" %
          (nm, e0.elapsed_time(e1) / 20 * 1e3, np.array_equal(ref, got)))
