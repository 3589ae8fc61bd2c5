#!/usr/bin/env python3
"""Shader cycles per phase of the team PCG (a -DDFA_PCG_PROFILE -DDFA_DEV_AB build: bash tools/ab_variant.sh prof all
-DDFA_PCG_PROFILE -DDFA_DEV_AB), thread 0 of member 0 of team 0, summed over one solve:
    DFA_LIB_PATH=dynfu_amd/build/libdynfu_amd_prof.so DFA_PCG_PROFILE_PRINT=1 python tools/team_pcg_phases.py C3
prints (through dfa_solver_get_stats) prof[0..5] = wait, copy, replicas, row product, publish, prologue."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

for name in sys.argv[1:] or ["C3"]:
    seq = bench.Sequence(name, torch.device("cuda", 0), n_frames=4)
    seq.build_graph(3)
    for _ in range(3):
        seq.solver.solve(seq.params)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    seq.solver.solve(seq.params)
    e1.record()
    torch.cuda.synchronize()
    print(name, "solve %.3f ms" % e0.elapsed_time(e1), "(labels of the line below: wait, copy, replicas, row product, publish, prologue)", file=sys.stderr)
    st = seq.solver.stats()
    print(name, st, seq.solver.team_pcg_info(), file=sys.stderr)
