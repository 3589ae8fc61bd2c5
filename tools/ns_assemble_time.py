"""dev (GPU): per-launch time of the north-star linearise / assemble kernels.  usage: python tools/ns_assemble_time.py C3"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dynfu_amd as A
from dynfu_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
A.load()
cfg = synth.CONFIGS[name]
c = synth.canonical(cfg)
intr = synth.intrinsics(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
maps = [A.compute_points_normals(dev(synth.depth_frame(cfg, f)), *intr) for f in range(3, 9)]
s = A.Solver6(cfg["D"], len(c["verts"]), cfg["k"])
keep = [dev(c[n]) for n in ("node_pos", "node_dq", "node_w", "verts", "normals")]
s.set_problem(*keep)
prm = A.Solve6Params(num_iter=2, gn_iter=5, linear_iter=4, pcg_tol=1e-6, **synth.SOLVER)
lin, asm = [], []
s.enable_timing(True)
for P, Nm in maps:
    s.solve(P, Nm, *intr, prm)
    tm = s.timing()
    lin.append(tm["linearise_ms"] / tm["gn_iterations"]), asm.append(tm["assemble_ms"] / tm["gn_iterations"])
print("%s: linearise %.4f ms, assemble %.4f ms per Gauss-Newton iteration (median of %d frames); %s" %
      (name, np.median(lin[1:]), np.median(asm[1:]), len(lin) - 1, os.environ.get("DFA_TAG", "")))
