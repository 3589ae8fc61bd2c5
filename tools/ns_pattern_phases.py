"""dev (GPU; a -DDFA_S6_TIMING build): per-workgroup phase clocks of s6_pattern_kernel (once per frame).  usage: ... C3"""
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
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
D = cfg["D"]
s = A.Solver6(D, len(c["verts"]), cfg["k"])
keep = [dev(c[n]) for n in ("node_pos", "node_dq", "node_w", "verts", "normals")]
for _ in range(3):
    s.set_problem(*keep)
torch.cuda.synchronize()
n = min(D, 16384)
buf = np.zeros((n, 16), np.uint64)
L.dfa_dev_s6_timing.argtypes = [C.c_void_p, C.c_int]
assert L.dfa_dev_s6_timing(buf.ctypes.data, n) == 0
t = buf.astype(np.float64)
names = ["sort + deal", "hash the neighbours", "rank the columns", "split by slot", "work units", "(end)"]
tot = t[:, 10]
us = lambda x: x / 2250.0
print("%s: %d workgroups, lifetime mean %.1f us (p95 %.1f)" % (name, n, us(tot.mean()), us(np.percentile(tot, 95))))
for i, nm in enumerate(names):
    col = t[:, 1 + i]
    print("  %-22s mean %7.2f us  (%4.1f %%)" % (nm, us(col.mean()), 100 * col.sum() / tot.sum()))
