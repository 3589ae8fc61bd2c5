"""stats of one solve under every PCG path: python tools/pcg_variants.py T1 3 2   (config, num_iter, nonlinear_iter)"""
import os as _os
_os.environ.setdefault("DFA_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dynfu_amd", "libdynfu_amd_dev.so"))  # the DFA_* switches exist in the development flavour only
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "T1"
ni, nl = int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 2
A.load()
cfg = synth.CONFIGS[name]; k = cfg["k"]; c = synth.canonical(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
nodes, nw, ndq, verts = map(dev, (c["node_pos"], c["node_w"], c["node_dq"], c["verts"]))
idx, w = A.knn(nodes, nw, verts, k)
t_true = synth.true_translations(c["node_pos"], 3, k)
live = dev(synth.live_vertices(c["verts"], idx.cpu().numpy(), w.cpu().numpy(), t_true))
for variant in (None, "1", "3", "0"):
    if variant is None:
        os.environ.pop("DFA_PCG_VARIANT", None)
    else:
        os.environ["DFA_PCG_VARIANT"] = variant
    s = A.Solver(cfg["D"], len(c["verts"]), k)
    s.set_problem(nodes, ndq, nw, verts, live)
    s.solve(A.SolveParams(num_iter=ni, nonlinear_iter=nl, linear_iter=200, pcg_tol=1e-6, gn_tol=0.0, **synth.SOLVER))
    st = s.stats()
    err = float((s.translations() - dev(t_true)).abs().max())
    print("variant", variant, st, "max|t - t*| %.2e" % err)
    s.close()
