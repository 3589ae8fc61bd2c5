"""dev check of the north-star solve: HIP vs the double-precision oracle on a synthetic frame.
usage: python tools/solve6_dev.py [T1] [frame]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dynfu_amd as A
import oracle as O
from dynfu_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "T1"
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 6
A.load()
cfg = synth.CONFIGS[name]
k = cfg["k"]
c = synth.canonical(cfg)
intr = synth.intrinsics(cfg)
depth = synth.depth_frame(cfg, frame)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P, Nm = A.compute_points_normals(dev(depth), *intr)
Pr, Nr = O.points_normals(depth, *intr)
same = lambda a, b: np.array_equal(a.view(np.uint32), b.view(np.uint32))
print("points/normals bit-exact:", same(P.cpu().numpy(), Pr), same(Nm.cpu().numpy(), Nr))
kw = dict(num_iter=2, gn_iter=3, linear_iter=60, lambda_=200.0)
s = A.Solver6(cfg["D"], len(c["verts"]), k)
nodes, ndq, nw, cv, cn = map(dev, (c["node_pos"], c["node_dq"], c["node_w"], c["verts"], c["normals"]))
s.set_problem(nodes, ndq, nw, cv, cn)
prm = A.Solve6Params(**kw)
s.solve(P, Nm, *intr, prm)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    s.solve(P, Nm, *intr, prm)
torch.cuda.synchronize()
print("HIP solve %.2f ms" % ((time.time() - t0) / 5 * 1e3), s.stats())
dq = s.node_dq().cpu().numpy()
if len(c["verts"]) <= 70000:
    t0 = time.time()
    dq_ref, st = O.solve6(c["node_pos"], c["node_dq"], c["node_w"], k, c["verts"], c["normals"], Pr, Nr, intr, threads=8, **kw)
    print("oracle %.2f s" % (time.time() - t0), st)
    idx, wn, reg = O.graph6(c["node_pos"], c["node_w"], k, c["verts"])
    pa, _ = O.warp6(dq, idx, wn, c["verts"])
    pb, _ = O.warp6(dq_ref, idx, wn, c["verts"])
    d = np.linalg.norm(pa - pb, axis=1)
    print("warped vertices HIP vs oracle: mean %.2e max %.2e m" % (d.mean(), d.max()))
    wv, _ = s.warp()
    print("device warp vs oracle warp of the same dq: max %.2e" % np.abs(wv.cpu().numpy() - pa).max())
    print("dq diff max", np.abs(dq - dq_ref).max())
