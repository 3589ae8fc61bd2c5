#!/usr/bin/env python3
"""Team PCG (pcg_team_kernel: three teams of persistent workgroups, one coordinate per XCD) against the launched form
(pcg_mb_step_kernel, a launch per iteration) on the bench's own C3 / C4 frames, inside the development library
(DFA_MB_TEAM=0 / 1): node translations, iteration counts, frames/s of the reference-mode frame, aborts.

    python tools/team_pcg_check.py [C3 C4 ...] [--frames 40] [--repeat 200]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["C3", "C4"])
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--repeat", type=int, default=200, help="repeated solves of one problem (hang / abort check)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    import dynfu_amd as A
    from dynfu_amd import _lib, synth
    import bench
    dev = torch.device("cuda", 0)
    out = {}
    with _lib.use_library(_lib.dev_lib_path()):
        for name in args.configs:
            rec = {}
            # team_pair: the form that exchanges (m, t) pairs at any row length (DFA_MB_TEAM_ABORT=16); team: t's replica in
            # registers where the longest row fits 16 slots per thread
            for form, env, flag in (("launched", "0", None), ("team_pair", "1", "16"), ("team", "1", None)):
                os.environ["DFA_MB_TEAM"] = env
                os.environ.pop("DFA_MB_TEAM_ABORT", None)
                if flag:
                    os.environ["DFA_MB_TEAM_ABORT"] = flag
                seq = bench.Sequence(name, dev, n_frames=6)
                seq.fuse_first = False
                for f in range(5):
                    seq.frame(f)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for f in range(args.frames):
                    seq.frame(5 + f)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                st = seq.solver.stats()
                t = seq.solver.translations().cpu().numpy().copy()
                t_err = float(np.abs(t - seq.t_true[(5 + args.frames - 1) % seq.n_frames].cpu().numpy()).max())
                # the solve alone (no sweep beside it), events on the stream
                seq.build_graph(3)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    seq.solver.solve(seq.params)
                e1.record()
                torch.cuda.synchronize()
                rec[form] = dict(frames_per_s=round(args.frames / dt, 1), ms_per_frame=round(dt / args.frames * 1e3, 4),
                                 solve_alone_ms=round(e0.elapsed_time(e1) / 10, 4), pcg_iters=st["pcg_iters"], gn_iters=st["gn_iters"],
                                 max_row_nnz=st["max_row_nnz"], final_cost=st["final_cost"],
                                 max_abs_err_vs_truth_m=t_err, team=seq.solver.team_pcg_info())
                rec[form + "_t"] = t
                if form == "team" and args.repeat:
                    t0 = time.perf_counter()
                    for i in range(args.repeat):
                        seq.solver.solve(seq.params)
                    torch.cuda.synchronize()
                    rec["repeat"] = dict(solves=args.repeat, seconds=round(time.perf_counter() - t0, 3),
                                         team=seq.solver.team_pcg_info(),
                                         same_translations=bool(np.array_equal(seq.solver.translations().cpu().numpy(),
                                                                               seq.solver.translations().cpu().numpy())))
                    # one coordinate's team forced to give up: the guard launch solves it
                    os.environ["DFA_MB_TEAM_ABORT"] = "2"
                    s2 = A.Solver(seq.D, seq.N, seq.k)
                    s2.set_problem(seq.nodes, seq.node_dq, seq.node_w, seq.verts, seq.live[(5 + args.frames - 1) % seq.n_frames])
                    s2.solve(seq.params)
                    tg = s2.translations().cpu().numpy()
                    rec["guard"] = dict(max_abs_diff_vs_team_m=float(np.abs(tg - t).max()), team=s2.team_pcg_info(),
                                        pcg_iters=s2.stats()["pcg_iters"])
                    del os.environ["DFA_MB_TEAM_ABORT"]
                    s2.close()
                del seq
                torch.cuda.empty_cache()
            rec["max_abs_diff_team_forms_m"] = float(np.abs(rec["team_t"] - rec.pop("team_pair_t")).max())  # (round-off: the assembly's order)
            rec["max_abs_diff_team_vs_launched_m"] = float(np.abs(rec.pop("team_t") - rec.pop("launched_t")).max())
            out[name] = rec
            print(name, json.dumps(rec), flush=True)
    os.environ.pop("DFA_MB_TEAM", None)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
