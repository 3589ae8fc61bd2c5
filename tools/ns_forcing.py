"""dev (GPU): time and iteration counts of the north-star solve under PCG forcing schedules / iteration caps.
usage: python tools/ns_forcing.py C2 [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dynfu_amd as A
from dynfu_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
A.load()
cfg = synth.CONFIGS[name]
k = cfg["k"]
c = synth.canonical(cfg)
intr = synth.intrinsics(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
maps = [A.compute_points_normals(dev(synth.depth_frame(cfg, f)), *intr) for f in range(3, 3 + frames)]
s = A.Solver6(cfg["D"], len(c["verts"]), k)
keep = [dev(c[n]) for n in ("node_pos", "node_dq", "node_w", "verts", "normals")]
s.set_problem(*keep)
gn = cfg["gn_iters"]
outer = 2 if gn % 2 == 0 else 1
for tol, cap, first, decay in [(1e-6, 64, 0, 1), (1e-2, 64, 0, 1), (1e-2, 32, 0, 1), (1e-3, 64, 0.1, 0.5), (1e-3, 32, 0.1, 0.5),
                               (1e-3, 48, 0.1, 0.5), (1e-1, 16, 0, 1), (1e-6, 8, 0, 1)]:
    prm = A.Solve6Params(num_iter=outer, gn_iter=gn // outer, linear_iter=cap, pcg_tol=tol, pcg_tol_first=first, pcg_tol_decay=decay,
                         **synth.SOLVER)
    for P, Nm in maps[:2]:
        s.solve(P, Nm, *intr, prm)
    s.enable_timing(True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tms = []
    e0.record()
    for P, Nm in maps:
        s.solve(P, Nm, *intr, prm)
    e1.record()
    e1.synchronize()
    tm = s.timing()
    st = s.stats()
    s.enable_timing(False)
    e0.record()
    for P, Nm in maps:
        s.solve(P, Nm, *intr, prm)
    e1.record()
    e1.synchronize()
    launches = (cap + 2) * st["gn_iters"]
    print("%s tol %g cap %d first %g decay %g: %.3f ms/solve | last frame: linearise %.3f assemble %.3f pcg %.3f ms; pcg its %s "
          "(%d launches, %.2f us per launch); rel %s; cost %.4g -> %.4g" %
          (name, tol, cap, first, decay, e0.elapsed_time(e1) / len(maps), tm["linearise_ms"], tm["assemble_ms"], tm["pcg_ms"],
           st["pcg_it_hist"], launches, 1e3 * tm["pcg_ms"] / launches, ["%.2g" % r for r in st["pcg_rel_hist"]],
           st["initial_cost"], st["final_cost"]), flush=True)
