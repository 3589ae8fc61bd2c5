"""dev tool: PCG kernel time as a function of the iteration count (fixed cost + per-iteration cost)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = synth.CONFIGS[cfgname]
c = synth.canonical(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
nodes, node_w, node_dq, verts = (dev(c[n]) for n in ("node_pos", "node_w", "node_dq", "verts"))
k = cfg["k"]
idx, w = A.knn(nodes, node_w, verts, k)
t_true = synth.true_translations(c["node_pos"], 7, cfg["k"])
live = dev(synth.live_vertices(c["verts"], idx.cpu().numpy(), w.cpu().numpy(), t_true))
s = A.Solver(cfg["D"], len(c["verts"]), k)
s.set_problem(nodes, node_dq, node_w, verts, live)
for iters in (0, 1, 2, 8, 32, 128, 256):
    prm = A.SolveParams(num_iter=1, nonlinear_iter=1, linear_iter=iters, pcg_tol=0.0, gn_tol=0.0, **synth.SOLVER)
    best = 1e9
    for rep in range(5):
        s.enable_timing(True)  # starts a new measurement (timings accumulate over solves otherwise)
        s.solve(prm)
        tm = s.timing()
        best = min(best, tm["pcg_ms"])
    st = s.stats()
    print("linear_iter %4d  executed %4d  pcg %.1f us  assemble %.1f us  nnz %d max_row %d" %
          (iters, st["pcg_iters"], best * 1e3, tm["assemble_ms"] * 1e3, tm["matrix_nnz"], st["max_row_nnz"]))
