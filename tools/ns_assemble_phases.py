"""dev (GPU; a -DDFA_S6_TIMING build): per-workgroup phase clocks of one s6_assemble2 launch.
usage: DFA_EXTRA_CXXFLAGS=-DDFA_S6_TIMING python -c 'from dynfu_amd import build as B; B.build()'; python tools/ns_assemble_phases.py C3"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dynfu_amd as A
from dynfu_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
L = A.load()
cfg = synth.CONFIGS[name]
c = synth.canonical(cfg)
intr = synth.intrinsics(cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
maps = [A.compute_points_normals(dev(synth.depth_frame(cfg, f)), *intr) for f in range(3, 6)]
D = cfg["D"]
s = A.Solver6(D, len(c["verts"]), cfg["k"])
keep = [dev(c[n]) for n in ("node_pos", "node_dq", "node_w", "verts", "normals")]
s.set_problem(*keep)
prm = A.Solve6Params(num_iter=1, gn_iter=2, linear_iter=4, pcg_tol=1e-6, **synth.SOLVER)
for P, Nm in maps:
    s.solve(P, Nm, *intr, prm)
torch.cuda.synchronize()
n = min(D, 16384)
buf = np.zeros((n, 16), np.uint64)
L.dfa_dev_s6_timing.argtypes = [C.c_void_p, C.c_int]
assert L.dfa_dev_s6_timing(buf.ctypes.data, n) == 0
t = buf.astype(np.float64)
names = ["prologue", "stage", "(no prep)", "(wait 1)", "records", "unit sums", "M load", "epilogue", "tail"]
tot, wall = t[:, 10], t[:, 11]
ratio = np.median(tot / np.maximum(wall, 1))
print("%s: %d workgroups; clock64 per wall_clock64 tick: %.2f (wall = 100 MHz -> clock64 at %.0f MHz)" % (name, n, ratio, ratio * 100))
us = lambda x: x / (ratio * 100.0)
print("rows per node: mean %.0f, max %.0f; blocks per row: mean %.1f, max %.0f" % (t[:, 12].mean(), t[:, 12].max(), t[:, 13].mean(), t[:, 13].max()))
print("workgroup lifetime: mean %.1f us, median %.1f, p95 %.1f, max %.1f" % (us(tot.mean()), us(np.median(tot)), us(np.percentile(tot, 95)), us(tot.max())))
for i, nm in enumerate(names):
    col = t[:, 1 + i]
    print("  %-10s mean %7.2f us  (%4.1f %%)   p95 %7.2f" % (nm, us(col.mean()), 100 * col.sum() / tot.sum(), us(np.percentile(col, 95))))
sub = buf[:, 15]
print("  inside unit sums: waves wait for each other %.2f us, pair sums + stores + requests %.2f, second barrier %.2f, sums + barrier %.2f" %
      (us(t[:, 4].mean()), us((sub >> np.uint64(32)).astype(np.float64).mean()), us((sub & np.uint64(0xffffffff)).astype(np.float64).mean()),
       us((t[:, 6] - t[:, 4] - (sub >> np.uint64(32)).astype(np.float64) - (sub & np.uint64(0xffffffff)).astype(np.float64)).mean())))
start = t[:, 14] - t[:, 14].min()
end = start + wall
print("launch span (first start -> last end): %.1f us; last start at %.1f us" % (end.max() / 100.0, start.max() / 100.0))
print("start times [us], deciles:", " ".join("%.1f" % (x / 100.0) for x in np.percentile(start, np.arange(0, 101, 10))))
# concurrency: how many workgroups are alive at a time
ev = np.concatenate([np.stack([start, np.ones(n)], 1), np.stack([end, -np.ones(n)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
alive = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0], append=ev[-1, 0])
print("workgroups alive, time-weighted mean: %.0f (of %d CUs x 4)" % ((alive * dt).sum() / max(dt.sum(), 1), 256))
