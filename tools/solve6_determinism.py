"""stress: the same north-star solve repeated; prints how many distinct results appear"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "T1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kw = dict(num_iter=int(sys.argv[3]) if len(sys.argv) > 3 else 2, gn_iter=int(sys.argv[4]) if len(sys.argv) > 4 else 3,
          linear_iter=int(sys.argv[5]) if len(sys.argv) > 5 else 80, lambda_=200.0)
A.load()
cfg = synth.CONFIGS[name]; k = cfg["k"]; c = synth.canonical(cfg); intr = synth.intrinsics(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P, Nm = A.compute_points_normals(dev(synth.depth_frame(cfg, 6)), *intr)
s = A.Solver6(cfg["D"], len(c["verts"]), k)
keep = list(map(dev, (c["node_pos"], c["node_dq"], c["node_w"], c["verts"], c["normals"])))
seen = {}
for i in range(reps):
    s.set_problem(*keep)
    s.solve(P, Nm, *intr, A.Solve6Params(**kw))
    st = s.stats()
    h = hashlib.md5(s.node_dq().cpu().numpy().tobytes()).hexdigest()[:8]
    seen.setdefault(h, []).append((i, st["final_cost"], st["pcg_iters"]))
for h, v in seen.items():
    print(h, len(v), v[0][1:], [x[0] for x in v][:10])
