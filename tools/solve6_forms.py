"""one north-star Gauss-Newton step with a long PCG under the current assembly settings (DFA_S6_ASSEMBLE, DFA_S6_RC):
prints the step's statistics and a few node transforms, to compare the forms across processes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
A.load()
cfg = synth.CONFIGS[name]; k = cfg["k"]; c = synth.canonical(cfg); intr = synth.intrinsics(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P, Nm = A.compute_points_normals(dev(synth.depth_frame(cfg, 6)), *intr)
s = A.Solver6(cfg["D"], len(c["verts"]), k)
s.set_problem(*map(dev, (c["node_pos"], c["node_dq"], c["node_w"], c["verts"], c["normals"])))
s.solve(P, Nm, *intr, A.Solve6Params(num_iter=1, gn_iter=1, linear_iter=400, lambda_=200.0, pcg_tol=1e-7))
st = s.stats()
dq = s.node_dq().cpu().numpy()
np.save("/tmp/dq_%s_%s.npy" % (os.environ.get("DFA_S6_ASSEMBLE", "d"), os.environ.get("DFA_S6_RC", "d")), dq)
print(os.environ.get("DFA_S6_ASSEMBLE"), os.environ.get("DFA_S6_RC"), {n: st[n] for n in ("initial_cost", "final_cost", "pcg_iters", "gn_iters")},
      "sum|dq|", float(np.abs(dq).sum()), dq[1234 % len(dq)][:4])
