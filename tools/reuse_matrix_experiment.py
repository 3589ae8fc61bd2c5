""""measure matrix reuse across north-star Gauss-Newton iterations" — iterations >= 1 of an outer
iteration keep H and M^-1 of iteration 0 and re-linearise residuals and gradient only (orc6_params.reuse_matrix, the
pattern of the reference-parity solve's regradient).  Decided at the level of the fp64 statement (oracle/solve6_oracle.c,
CPU): the variant is compared with full Gauss-Newton on bench.py's own frames and parameters, with and without the stopping
rule.  Kill criterion (VERDICT): final cost per valid row worse than 2 %, or PCG iterations up by more than 1.5 x.

usage: python tools/reuse_matrix_experiment.py C2 0 50 > profiles/r05_reuse_matrix_C2.md   (config, first frame, frames)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
from dynfu_amd import synth  # noqa: E402

cfgn, f0, nf = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = synth.CONFIGS[cfgn]
intr = synth.intrinsics(cfg)
c = synth.canonical(cfg)
gn = cfg["gn_iters"]
outer = 2 if gn % 2 == 0 else 1
base = dict(num_iter=outer, gn_iter=gn // outer, linear_iter=64, pcg_tol=1e-3, pcg_tol_first=0.1, pcg_tol_decay=0.5, pcg_tol_adapt=0.9,
            threads=min(8, os.cpu_count() or 1), **synth.SOLVER)
variants = [("full GN, every iteration", dict()), ("full GN + stopping rule", dict(gn_tol=1e-3)),
            ("reuse_matrix, every iteration", dict(reuse_matrix=1)), ("reuse_matrix + stopping rule", dict(reuse_matrix=1, gn_tol=1e-3))]
rows = {name: [] for name, _ in variants}
for f in range(f0, f0 + nf):
    P, Nm = O.points_normals(synth.depth_frame(cfg, f), *intr)
    for name, kw in variants:
        dq, st = O.solve6(c["node_pos"], c["node_dq"], c["node_w"], cfg["k"], c["verts"], c["normals"], P, Nm, intr, **base, **kw)
        e, nv = O.cost6(c["node_pos"], dq, c["node_w"], cfg["k"], c["verts"], c["normals"], P, Nm, intr, **base)
        rows[name].append((st["final_cost"] / max(1, st["valid_last"]), e / max(1, nv), st["gn_solves"], st["pcg_iters"], st["gn_rejected"]))
print("# reuse of the normal matrix across Gauss-Newton iterations — %s, frames %d..%d (fp64 statement, bench.py's parameters)\n" % (cfgn, f0, f0 + nf - 1))
print("| variant | final cost / valid row (mean) | energy of the returned transforms, re-associated, / valid row (mean; worst frame vs full GN) | solves / frame | PCG iterations / frame | rejected steps / frame |")
print("|---|---|---|---|---|---|")
ref = np.array(rows["full GN, every iteration"])
for name, _ in variants:
    a = np.array(rows[name])
    print("| %s | %.4g | %.4g (x %.2f) | %.2f | %.1f | %.2f |" % (name, a[:, 0].mean(), a[:, 1].mean(), (a[:, 1] / ref[:, 1]).max(), a[:, 2].mean(), a[:, 3].mean(), a[:, 4].mean()))
