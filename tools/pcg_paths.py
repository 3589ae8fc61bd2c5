"""compare the PCG paths (default routing vs DFA_PCG_VARIANT) on a synthetic config: iterations and costs per number
of outer iterations.  usage: DFA_PCG_VARIANT=4 python tools/pcg_paths.py C3"""
import os as _os
_os.environ.setdefault("DFA_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dynfu_amd", "libdynfu_amd_dev.so"))  # the DFA_* switches exist in the development flavour only
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
A.load()
cfg = synth.CONFIGS[name]; k = cfg["k"]; c = synth.canonical(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
nodes, nw, ndq, verts = map(dev, (c["node_pos"], c["node_w"], c["node_dq"], c["verts"]))
idx, w = A.knn(nodes, nw, verts, k)
t_true = synth.true_translations(c["node_pos"], 3, k)
live = dev(synth.live_vertices(c["verts"], idx.cpu().numpy(), w.cpu().numpy(), t_true))
s = A.Solver(cfg["D"], len(c["verts"]), k)
for n in (1, 2, 3, 5, 10):
    s.set_problem(nodes, ndq, nw, verts, live)
    s.solve(A.SolveParams(num_iter=n, nonlinear_iter=1, linear_iter=256, pcg_tol=1e-6, gn_tol=0.0, **synth.SOLVER))
    st = s.stats()
    err = float((s.translations() - dev(t_true)).abs().max())
    print(os.environ.get("DFA_PCG_VARIANT", "default"), name, "outer", n, "pcg_iters", st["pcg_iters"], "gn", st["gn_iters"],
          "cost %.6e -> %.6e" % (st["initial_cost"], st["final_cost"]), "max|t - t*| %.2e" % err)
