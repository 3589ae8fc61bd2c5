"""dev (GPU; a -DDFA_PCG_PROFILE -DDFA_DEV_AB build: bash tools/ab_variant.sh prof all -DDFA_PCG_PROFILE -DDFA_DEV_AB):
per-workgroup life of one reference-mode assemble_kernel launch.
usage: DFA_LIB_PATH=dynfu_amd/build/libdynfu_amd_prof.so python tools/ref_assemble_phases.py C4"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
seq = bench.Sequence(name, torch.device("cuda", 0), n_frames=3)
import dynfu_amd as A
L = A.load()
seq.params.num_iter = 2
for f in range(3):
    seq.build_graph(f)
    seq.solver.solve(seq.params)
torch.cuda.synchronize()
D = seq.D
n = min(D, 32768)
buf = np.zeros((n, 8), np.uint64)
L.dfa_dev_asm_timing.argtypes = [C.c_void_p, C.c_int]
assert L.dfa_dev_asm_timing(buf.ctypes.data, n) == 0
t = buf.astype(np.float64)
start, wall, rows, nnz = t[:, 0] - t[:, 0].min(), t[:, 1], t[:, 2], t[:, 3]
cyc = t[:, 4:8]
ratio = np.median(cyc.sum(1) / np.maximum(wall, 1))
us = lambda x: x / (ratio * 100.0)
print("%s: %d workgroups; clock64 per 100 MHz tick %.2f" % (name, n, ratio))
print("rows per node: mean %.0f p95 %.0f max %.0f; entries per matrix row: mean %.1f max %.0f" % (rows.mean(), np.percentile(rows, 95), rows.max(), nnz.mean(), nnz.max()))
print("workgroup life [us]: mean %.1f median %.1f p95 %.1f max %.1f" % (wall.mean() / 100, np.median(wall) / 100, np.percentile(wall, 95) / 100, wall.max() / 100))
for i, nm in enumerate(("init", "list + hash", "reduce + barrier", "compact + tail")):
    print("  %-18s mean %7.2f us (%4.1f %%)  p95 %7.2f" % (nm, us(cyc[:, i].mean()), 100 * cyc[:, i].sum() / cyc.sum(), us(np.percentile(cyc[:, i], 95))))
end = start + wall
print("launch span %.1f us; last start at %.1f us" % (end.max() / 100, start.max() / 100))
ev = np.concatenate([np.stack([start, np.ones(n)], 1), np.stack([end, -np.ones(n)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
alive = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0], append=ev[-1, 0])
print("workgroups alive, time-weighted mean: %.0f (256 CUs)" % ((alive * dt).sum() / max(dt.sum(), 1)))
print("start deciles [us]:", " ".join("%.0f" % (x / 100) for x in np.percentile(start, np.arange(0, 101, 10))))
