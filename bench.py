#!/usr/bin/env python3
"""bench.py — frames/sec of the warp-solve + TSDF-fuse hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one frame of BASELINE.json's metric: compute_dists + clear+integrate (fused sweep)
of the 512^3 volume, graph build (k-NN of 262 144 canonical vertices among 2 048 nodes + the
node->rows transpose), 5 Gauss-Newton iterations (Tukey re-weighting, assembly of the normal
equations, block-Jacobi PCG <= 256 iterations or relative residual 1e-6), write-back of the
node transforms and the post-solve warpToLive of the canonical frame.  All inputs are
synthetic (dynfu_amd/synth.py) and resident in HBM before the timed region starts.

The fuse (HBM-bound, whole chip) and the solve (latency-bound, a few CUs) of one frame are
independent and run on two HIP streams; a frame ends when both have finished.

Multi-GPU: replicas only — every rank runs its own sequence on its own GPU with no data-path
collective; torch.distributed (RCCL) is used for the start/stop barrier and the
max-over-ranks time.  value = frames of all ranks / max time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
LDS_GATHER_PEAK_GBS = 921.6  # 3 CUs x 128 B/clk x 2.4 GHz: the LDS read rate of the three CUs the reference-mode PCG runs on
TIMING_SAMPLE = 8  # every 8th timed frame carries the hipEvent brackets of the per-kernel report
# HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x 2 for wide read streams + WRITE_SIZE, corrected as
# MI355X_MICROARCH.md prescribes).  Collected offline — PMC needs its own runs — by tools/round5_profile.sh and written to
# profiles/traffic.json by tools/profile_summary.py (nothing is copied by hand): every figure names the tracked file it was
# read from and the commit that file was measured at, so a figure older than the kernel it describes is visible as such.
def _load_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return {}
    return {tuple(k.split("/", 1)): (v["bytes_per_launch"], v.get("source"), v.get("commit")) for k, v in t.items()}


PMC_TRAFFIC = _load_traffic()
PMC_TRAFFIC_BYTES = {k: v[0] for k, v in PMC_TRAFFIC.items()}


def traffic_source(config, key):
    t = PMC_TRAFFIC.get((config, key))
    return "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) measured at commit %s" % (t[1], t[2]) if t else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4", "T0", "T1"])
    ap.add_argument("--fuse-first", action="store_true",
                    help="launch the TSDF sweep at the start of the frame (A/B; default: behind the graph build)")
    ap.add_argument("--fuse-after-build", action="store_true",
                    help="launch the TSDF sweep behind the graph build (A/B; default up to 2048 nodes: in the shadow of the "
                         "first PCG, behind the solver's overlap event)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-northstar", action="store_true", help="skip the short north-star-mode measurement of the default run")
    ap.add_argument("--no-pipelined-probe", action="store_true", help="skip the short pipelined-throughput measurement")
    ap.add_argument("--sequences-per-gpu", type=int, nargs="*", default=[1, 2, 4],
                    help="the multi_sequence figure: S independent sequences on this GPU, each on streams of its own (default 1 2 4)")
    ap.add_argument("--no-multi-sequence", action="store_true", help="skip the multi_sequence figure")
    ap.add_argument("--sequences-threads", action="store_true", help="multi_sequence: a host thread per sequence instead of one "
                                                                     "thread enqueuing round-robin")
    ap.add_argument("--warm-start", action="store_true", help="--mode northstar: every frame starts from the transforms the frame before solved (default: from the canonical state)")
    ap.add_argument("--no-live-depth", action="store_true", help="skip the short measurement on the reference's data flow with noisy depth")
    ap.add_argument("--live", default="targets", choices=["targets", "depth"],
                    help="targets: index-aligned live vertices canon + sum w t* (SURVEY 8d, the headline workload); depth: the "
                         "reference's data flow — noisy depth, marching-cubes live cloud, nearest-vertex correspondence "
                         "(prints the line of that workload alone)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short C3 / C4 measurements (both modes) of the default run")
    ap.add_argument("--no-raycast", action="store_true", help="skip the raycast figures (SURVEY 8d: reported separately)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the DynFusion::operator() sequence (the reference's own timed region, C++ adaptor classes)")
    ap.add_argument("--cpu-frames", type=int, default=2, help="frames of the bounded CPU sample per thread count of its sweep")
    ap.add_argument("--serial", action="store_true", help="run fuse and solve on one stream (A/B of the overlap)")
    ap.add_argument("--pipeline", action="store_true",
                    help="ref mode: build frame f+1's graphs (k-NN, transposition) on a third stream while frame f is "
                         "being solved (two solver plans); every frame still does all of its work")
    ap.add_argument("--mode", default="ref", choices=["ref", "northstar"],
                    help="ref: the reference's translation-only energy (energy.t); northstar: 6-DoF DQ-blend / "
                         "projective point-to-plane / ARAP solve (DESIGN.md 4.5) against the live depth map")
    ap.add_argument("--linear-iter", type=int, default=0, help="PCG iteration cap (default: 256 ref, 64 northstar)")
    ap.add_argument("--forcing", default="adaptive", choices=["adaptive", "geometric"],
                    help="northstar: PCG tolerance per Gauss-Newton iteration — Eisenstat-Walker (default) or 0.1 x 0.5^i")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the N > 1 run: nccl (= RCCL, the GPUs) or gloo (with --dry-run: the CPU test of "
                         "the rank launcher)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group check without a GPU: every rank sleeps instead of running frames, the line "
                         "carries the same contract fields (tests/test_bench_contract.py)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--dry-run-hang-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--no-rccl-selfcheck", action="store_true",
                    help="N = 1: do not build the one-rank process group that exercises RCCL init / all-reduce / barrier")
    ap.add_argument("--rank-timeout", type=float, default=3600.0,
                    help="--gpus N outside torchrun: wall-clock limit of the rank processes in seconds")
    ap.add_argument("--gn-tol", type=float, default=None,
                    help="northstar: dfa_solve6_params.gn_tol (default 1e-3: stopping rule + step acceptance; 0: every "
                         "Gauss-Newton iteration runs)")
    ap.add_argument("--overlap", action="store_true",
                    help="northstar: the sweep on a second stream beside the solve (the default of this mode is stream order: "
                         "its kernels fill the chip)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the barrier-bracketed timed region (the same K frames) runs this many times back to back; value / "
                         "ms_per_step are those of the MEDIAN region (a 20-step region is 13 ms: one region alone moves by percents)")
    ap.add_argument("--force-launcher", action="store_true",
                    help="--gpus 1: go through the rank launcher anyway (a fresh child with WORLD_SIZE=1, its core slice, rank 0's "
                         "line forwarded, the wall-clock limit) - the path every N > 1 run takes")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (every secondary figure, per-kernel table, notes); stdout carries ONE line of "
                         "at most 4 KB with the contract's fields")
    ap.add_argument("--no-adaptive-launch", action="store_true",
                    help="northstar: enqueue the full PCG launch budget of every Gauss-Newton iteration (A/B of "
                         "dfa_solve6_params.adaptive_launch)")
    return ap.parse_args()


STREAM_PROBES = []  # what concurrent_stream() found, per call (goes into config.streams_probe)


def concurrent_stream(device, other=()):
    """A HIP stream whose work really runs BESIDE the current stream's (and beside `other` streams').  HIP maps streams onto a
    few hardware queues, assigned at a stream's FIRST USE; two streams on one queue serialise.  Which queue a stream lands
    on depends on every stream the process has used before — with an RCCL communicator created first, the sweep stream of
    the first sequence shared the solve stream's queue and the C2 frame went from 0.65 to 0.77 ms (tools/rccl_launch_tax.py,
    tools/stream_queue_probe.py: about one stream in eight aliases the current one).  So candidates are created and used once
    until one demonstrably overlaps: a long fill on the busy streams, a one-element kernel on the candidate that must finish
    before the fill does."""
    import torch
    main = torch.cuda.current_stream(device)
    busy = [main] + list(other)
    big = torch.empty(96 << 20, dtype=torch.float32, device=device)  # 384 MiB: a fill is ~0.1 ms
    one = torch.zeros(1, device=device)
    rejected = []
    cand = None
    for attempt in range(8):
        cand = torch.cuda.Stream(device=device)
        with torch.cuda.stream(cand):  # first use: the hardware queue is bound (and created: milliseconds) here
            one.add_(1.0)
        ok = True
        for b in busy:
            e0, e_busy, e_side = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            torch.cuda.synchronize(device)
            with torch.cuda.stream(b):
                e0.record(b)
                for _ in range(4):
                    big.fill_(1.0)
                e_busy.record(b)
            cand.wait_event(e0)
            with torch.cuda.stream(cand):
                one.add_(1.0)
                e_side.record(cand)
            torch.cuda.synchronize(device)
            ok = ok and e0.elapsed_time(e_side) < 0.5 * e0.elapsed_time(e_busy)
        if ok:
            break
        rejected.append(cand)  # (kept alive until the choice is made: a destroyed stream's queue slot is handed out again)
    STREAM_PROBES.append(dict(candidates_tried=len(rejected) + 1, overlaps=bool(ok)))
    del big
    return cand


class Sequence:
    """Device-resident inputs + plan of one synthetic sequence."""

    def __init__(self, cfg_name, device, n_frames=None):
        import torch

        import dynfu_amd as A
        from dynfu_amd import synth
        self.A, self.synth, self.torch = A, synth, torch
        self.cfg = cfg = synth.CONFIGS[cfg_name]
        self.intr = synth.intrinsics(cfg)
        self.voxel, self.trunc, self.vol2cam, self.cam2vol, self.rinv = synth.volume_params(cfg)
        dim, W, H = cfg["dim"], cfg["width"], cfg["height"]
        self.n_frames = n_frames or synth.N_FRAMES  # the sequence repeats after this many frames
        self.depth_np = [synth.depth_frame(cfg, f) for f in range(self.n_frames)]
        self.depth = torch.from_numpy(np.stack(self.depth_np)).to(device)
        self.dists = torch.empty((H, W), dtype=torch.uint16, device=device)
        self.vol = torch.zeros((dim, dim, dim), dtype=torch.int32, device=device)
        c = self.canon = synth.canonical(cfg)
        self.k, self.D, self.N = cfg["k"], cfg["D"], len(c["verts"])
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.nodes, self.node_w, self.node_dq = dev(c["node_pos"]), dev(c["node_w"]), dev(c["node_dq"])
        self.verts, self.normals = dev(c["verts"]), dev(c["normals"])
        # live vertices of every frame: canon + sum_j w_j t*_j(frame)   (device-side, once)
        idx, w = A.knn(self.nodes, self.node_w, self.verts, self.k)
        t_true = torch.from_numpy(np.stack([synth.true_translations(c["node_pos"], f, cfg["k"])
                                            for f in range(self.n_frames)])).to(device)
        idx_l = idx.long().clamp(min=0)
        self.live = torch.stack([(self.verts.double() + (w.double()[..., None] * t_true[f].double()[idx_l]).sum(1))
                                 .float() for f in range(self.n_frames)])
        self.t_true = t_true
        self.solver = A.Solver(self.D, self.N, self.k)
        if os.environ.get("DFA_BENCH_DETERMINISTIC"):  # the order-stable variant (dfa_solver_set_deterministic), for A/B runs
            self.solver.set_deterministic(True)
        self.params = A.SolveParams(num_iter=cfg["gn_iters"], nonlinear_iter=1, linear_iter=256, pcg_tol=1e-6,
                                    gn_tol=0.0, **synth.SOLVER)
        self.s_fuse = concurrent_stream(device)
        self.warped = None
        self.fuse_events = []

    def fuse(self, f, timed_events=None):
        A = self.A
        A.compute_dists(self.depth[f % self.n_frames], self.dists, *self.intr)
        if timed_events is not None:
            e0, e1 = (self.torch.cuda.Event(enable_timing=True) for _ in range(2))
            e0.record()
        # (occ: the volume's occupancy map where a marching-cubes pass follows the sweep — SequenceLive)
        A.tsdf_clear_integrate(self.vol, self.dists, self.voxel, self.trunc, self.synth.MAX_WEIGHT, self.vol2cam,
                               *self.intr, occupancy=getattr(self, "occ", None), occupancy_known=getattr(self, "occ_known", False))
        if timed_events is not None:
            e1.record()
            timed_events.append((e0, e1))

    def build_graph(self, f):
        self.solver.set_problem(self.nodes, self.node_dq, self.node_w, self.verts, self.live[f % self.n_frames])

    def solve(self, f, graph_built=False):
        A = self.A
        if not graph_built:
            self.build_graph(f)
        self.solver.solve(self.params)
        # post-solve warpToLive of the canonical frame, through the plan's k-NN graph (same output as the stand-alone
        # dfa_warp_to_live, which would search the 262 144 x 2 048 neighbours a second time)
        self.warped, _ = self.solver.warp_to_live(self.normals)

    def frame(self, f, serial=False, timed_events=None):
        torch = self.torch
        cur = torch.cuda.current_stream()
        if serial:
            self.fuse(f, timed_events)
            self.solve(f)
            return
        if getattr(self, "pipelined", False) and self.D <= 2048:
            # fuse(f) in the shadow of the first PCG of frame f, graph build of frame f + 1 in the shadow of the second
            self.solve_pipelined(f, timed_events, shadows=True)
        elif getattr(self, "pipelined", False) or getattr(self, "fuse_first", False):
            self.s_fuse.wait_stream(cur)
            with torch.cuda.stream(self.s_fuse):
                self.fuse(f, timed_events)
            if getattr(self, "pipelined", False):
                self.solve_pipelined(f)
            else:
                self.solve(f)
        elif getattr(self, "ev_overlap", None) is not None:
            # The volume sweep (all CUs for ~0.13 ms) runs in the shadow of the first PCG (three CUs for ~0.11 ms): the
            # solver calls back behind its first assembly launch; the callback records an event there, lets the fuse
            # stream wait for it and enqueues the sweep (dfa_solver_set_overlap_callback).  Launched
            # behind the graph build instead (--fuse-after-build), the sweep holds every wave slot of the chip while
            # the first linearisation wants them: that kernel then takes 124 us instead of 22
            # (profiles/r02a_kernel_stats_bench_c2.csv).  Same work per frame either way.
            self.build_graph(f)
            self._overlap_job = (f, timed_events)
            self.solve(f, graph_built=True)  # calls self._fuse_in_shadow() behind its first assembly launch
        else:
            # The graph build (grid, k-NN, transposition: short kernels that fill the chip) runs alone; the volume
            # sweep starts behind it and overlaps the Gauss-Newton iterations.  Same work per frame as launching
            # both at once (--fuse-first), where the sweep slows the graph build's kernels by ~2x.
            self.build_graph(f)
            self.s_fuse.wait_stream(cur)
            with torch.cuda.stream(self.s_fuse):
                self.fuse(f, timed_events)
            self.solve(f, graph_built=True)
        cur.wait_stream(self.s_fuse)

    # ---- software pipeline across frames (--pipeline): two plans; the graphs of frame f+1 (a function of the node
    # positions and the new frame's vertices only) are built on a third stream while frame f is solved
    def enable_pcg_shadow(self):
        """fuse in the shadow of the first PCG (plans of <= 2048 nodes: the register-resident PCG on three CUs)"""
        self.ev_overlap = self.torch.cuda.Event()
        self.solver.set_overlap_callback(self._fuse_in_shadow)

    def _fuse_in_shadow(self, gn_iteration):
        if gn_iteration > 0:
            return
        torch = self.torch
        f, timed_events = self._overlap_job
        self.ev_overlap.record()  # on the solve's stream, behind the first assembly
        self.s_fuse.wait_event(self.ev_overlap)
        with torch.cuda.stream(self.s_fuse):
            self.fuse(f, timed_events)

    def enable_pipeline(self):
        torch = self.torch
        self.pipelined = True
        if getattr(self, "ev_overlap", None) is not None:
            self.solver.set_overlap_callback(None)
            self.ev_overlap = None
        self.plans = [self.solver, self.A.Solver(self.D, self.N, self.k)]
        if self.D <= 2048:
            self.ev_pipe = [torch.cuda.Event(), torch.cuda.Event()]
            for plan in self.plans:
                plan.set_overlap_callback(self._pipeline_shadow)
        self.s_graph = concurrent_stream(torch.device("cuda", torch.cuda.current_device()), other=[self.s_fuse])
        self.graph_ready = [None, None]   # event: plan i holds the graphs of its next frame
        self.plan_free = [None, None]     # event: plan i's last solve (and warp) has finished
        self.next_graph = None            # frame whose graphs plan[f % 2] holds

    def _build_graph(self, f):
        torch = self.torch
        i = f % 2
        with torch.cuda.stream(self.s_graph):
            if self.plan_free[i] is not None:
                self.s_graph.wait_event(self.plan_free[i])
            self.plans[i].set_problem(self.nodes, self.node_dq, self.node_w, self.verts, self.live[f % self.n_frames])
            ev = torch.cuda.Event()
            ev.record(self.s_graph)
            self.graph_ready[i] = ev

    def _pipeline_shadow(self, gn_iteration):
        """overlap callback of the pipelined schedule: the sweep of this frame behind the first assembly launch, the
        graph build of the next frame behind the second (each then runs beside a PCG on three CUs)"""
        torch = self.torch
        f, timed_events = self._pipe_job
        if gn_iteration in (0, -1):
            self.ev_pipe[0].record()
            self.s_fuse.wait_event(self.ev_pipe[0])
            with torch.cuda.stream(self.s_fuse):
                self.fuse(f, timed_events)
        if gn_iteration in (1, -1):
            self.ev_pipe[1].record()
            self.s_graph.wait_event(self.ev_pipe[1])
            self._build_graph(f + 1)
            self._pipe_built = True

    def solve_pipelined(self, f, timed_events=None, shadows=False):
        torch = self.torch
        cur = torch.cuda.current_stream()
        i = f % 2
        if self.next_graph != f:          # first frame (or a jump in the sequence): no graphs built ahead
            self.s_graph.wait_stream(cur)
            self._build_graph(f)
        if not shadows:
            self._build_graph(f + 1)      # runs concurrently with the solve below
        self.next_graph = f + 1
        cur.wait_event(self.graph_ready[i])
        self.solver = self.plans[i]
        self._pipe_job, self._pipe_built = (f, timed_events), False
        self.solver.solve(self.params)
        if shadows and not self._pipe_built:  # a solve with a single Gauss-Newton iteration has no second shadow
            self.s_graph.wait_stream(cur)
            self._build_graph(f + 1)
        self.warped, _ = self.solver.warp_to_live(self.normals)
        ev = torch.cuda.Event()
        ev.record(cur)
        self.plan_free[i] = ev


# inexact Newton with the Eisenstat-Walker forcing term (--forcing geometric: 0.1 x 0.5^i instead)
# gn_tol: the Gauss-Newton stopping rule + step acceptance (dfa_solve6_params.gn_tol; the reference runs Opt with earlyOut =
# true and nonLinearIter as a cap, src/dynfu/dyn_fusion.cpp:183-189) — `fixed_iterations` beside every north-star figure is
# the same frame with gn_tol = 0 (every iteration runs, as rounds 1-4 measured it)
NS_PCG = dict(pcg_tol=1e-3, pcg_tol_first=0.1, pcg_tol_decay=0.5, pcg_tol_adapt=0.9, adaptive_launch=1, gn_tol=1e-3)


class Sequence6(Sequence):
    """North-star mode: the live input of the solve is the depth frame itself (vertex / normal maps by
    computePointNormals), the unknowns are 6-DoF node twists (dfa_solver6)."""

    def __init__(self, cfg_name, device, linear_iter, pcg=None, n_frames=None):
        super().__init__(cfg_name, device, n_frames)
        A, cfg = self.A, self.cfg
        del self.solver, self.live
        self.solver = A.Solver6(self.D, self.N, self.k)
        gn = cfg["gn_iters"]
        outer = 2 if gn % 2 == 0 else 1
        self.pcg = dict(NS_PCG if pcg is None else pcg)
        self.params = A.Solve6Params(num_iter=outer, gn_iter=gn // outer, linear_iter=linear_iter, **self.pcg, **self.synth.SOLVER)
        self.gn_total = outer * (gn // outer)

    warm_start = False  # --warm-start: a frame starts from the transforms the frame before solved (as a warp field does between
                        # frames, and the C++ NorthStarSolver) instead of from the canonical state; the graphs are still rebuilt

    def build_graph(self, f):
        dq0 = self.node_dq
        if self.warm_start and getattr(self, "_solved", None) is not None:
            dq0 = self._solved
        self.solver.set_problem(self.nodes, dq0, self.node_w, self.verts, self.normals)

    def frame(self, f, serial=None, timed_events=None):
        # Every kernel of this mode fills the chip (linearise, assembly, a PCG step = thousands of workgroups): a sweep on
        # a second stream does not hide behind them, it takes their CUs — measured with streams that really overlap
        # (concurrent_stream): C2 781 -> 845, C3 301 -> 319, C4 162 -> 165 frames/s in stream order.  The reference-mode
        # frame (its PCG holds 3 CUs) keeps the second stream; `--overlap` gives this mode one too (A/B).
        return super().frame(f, not self.overlap if serial is None else serial, timed_events)

    overlap = False

    def solve(self, f, graph_built=False):
        A = self.A
        if not graph_built:
            self.build_graph(f)
        P, Nm = A.compute_points_normals(self.depth[f % self.n_frames], *self.intr)
        self.solver.solve(P, Nm, *self.intr, self.params)
        self.warped, self.warped_n = self.solver.warp()
        if self.warm_start:
            self._solved = self.solver.node_dq()  # (a copy: the next set_problem borrows its argument while the plan rewrites its own)


class SequenceLive(Sequence):
    """Reference mode on the reference's actual DATA FLOW (src/dynfu/dyn_fusion.cpp:119-134, 212-242) instead of the
    index-aligned zero-residual targets of SURVEY §8(d): the depth frame carries sigma = 1 mm noise, the live cloud is the
    marching-cubes soup of the frame's fused volume (every s-th vertex, so that the solve keeps this configuration's N
    rows), each live vertex is paired with its NEAREST canonical vertex (dfa_correspond = findCorrespondingFrame), and
    that (corresponding canonical, live) pair list is what the solver sees: residuals along the surface, outliers at
    the silhouette, Tukey weights below one."""

    N_NOISY = 12

    def __init__(self, cfg_name, device):
        super().__init__(cfg_name, device)
        A, torch, cfg = self.A, self.torch, self.cfg
        noisy = [self.synth.depth_frame(cfg, f, noise_mm=1.0) for f in range(self.N_NOISY)]
        self.depth = torch.from_numpy(np.stack(noisy)).to(device)
        self.n_frames = self.N_NOISY
        tri, nv = A.mc_default_tables()
        self.tri, self.nv = torch.from_numpy(tri).to(device), torch.from_numpy(nv).to(device)
        # the sweep records which 32 x 2 x 8-voxel boxes can hold surface; marching cubes reads only those (as the adaptor's
        # TsdfVolume / MarchingCubes pair does)
        self.occ = None if os.environ.get("DFA_BENCH_NO_OCCUPANCY") else A.tsdf_occupancy(self.vol)
        # volume and map start as zeros and only the sweeps write them: the map describes the volume, and the sweep leaves
        # the boxes of zeros that stay zeros alone (dfa_tsdf_clear_integrate_known_occ)
        self.occ_known = self.occ is not None and not os.environ.get("DFA_BENCH_NO_KNOWN_OCCUPANCY")
        # one sizing pass (host synchronisation outside any timed region).  The soup comes out in voxel order (z-major), so
        # the object's vertices (z < 2 m) precede the background plane's (z = 2.5 m): the live cloud of the solve is a
        # strided sample of that prefix — the canonical cloud covers the object only, and a plane vertex has
        # no canonical neighbour within a metre
        self.fuse(0)
        _, total = A.marching_cubes(self.vol, self.voxel, self.tri, self.nv, 0, occupancy=self.occ)
        self.mc_total = int(total.item())
        self.mc_cap = int(self.mc_total * 1.15) + 1024
        pts, _ = A.marching_cubes(self.vol, self.voxel, self.tri, self.nv, self.mc_cap, occupancy=self.occ)
        on_object = (pts[: self.mc_total, 2] + float(self.vol2cam[11])) < 2.0
        self.mc_object = int(on_object.sum().item())
        assert bool(on_object[: self.mc_object].all()), "the object's vertices are not a prefix of the soup"
        usable = int(0.98 * self.mc_object)  # the count moves a little from frame to frame
        self.rows = min(self.N, usable)
        self.stride = max(1, usable // self.rows)
        self.vol2cam_t = torch.tensor(self.vol2cam[9:12], device=device)

    def frame(self, f, serial=True, timed_events=None):
        A = self.A
        self.fuse(f, timed_events)                                                             # dyn_fusion.cpp:58,113-114
        pts, _ = A.marching_cubes(self.vol, self.voxel, self.tri, self.nv, self.mc_cap, occupancy=self.occ)  # :119-121
        live = (pts[: self.stride * self.rows: self.stride, :3] + self.vol2cam_t).contiguous()  # volume -> camera frame
        corr_v, _, _ = A.correspond(self.verts, self.normals, live, want_index=False)           # :212-242
        self.solver.set_problem(self.nodes, self.node_dq, self.node_w, corr_v, live)            # opt_solver.cpp:15-54
        self.solver.solve(self.params)
        self.warped, _ = self.solver.warp_to_live(None)
        self.live_last = live


def multi_sequence_probe(cfg_name, device, counts=(1, 2, 4), rounds=60, warmup=8, threads=False):
    """BASELINE config 5's workload shape on ONE device: S independent sequences (own volume, solver plan and streams), a
    frame of each enqueued round-robin by one host thread, no synchronisation between frames.  The reference-mode PCG of a C2
    frame holds 3 of 256 CUs for half of the frame; what a second and a fourth sequence make of the idle chip tells the
    reader of the 1/2/4/8-GPU curve how much of a node one GPU already covers.  Reported: aggregate frames/s per S, and the
    per-frame latency inside a sequence (hipEvents on its solve stream) — a secondary figure, NOT the headline."""
    import torch
    out = {}
    seqs, streams = [], []
    main = torch.cuda.current_stream(device)
    for S in sorted(set(counts)):
        while len(seqs) < S:
            # a solve stream of its own that runs beside every stream already in use, then (inside Sequence) a sweep stream
            # that runs beside that one
            s_solve = concurrent_stream(device, other=streams)
            with torch.cuda.stream(s_solve):
                q = Sequence(cfg_name, device, n_frames=12)
                q.fuse_first = False
                q.enable_pcg_shadow()
            streams += [s_solve, q.s_fuse]
            q.s_solve = s_solve
            seqs.append(q)
        lat = [[] for _ in range(S)]

        def one(i, q, n, f0, timed):
            torch.cuda.set_device(device)
            with torch.cuda.stream(q.s_solve):
                for r in range(n):
                    if timed:
                        e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                        e0.record()
                    q.frame(f0 + r)
                    if timed:
                        e1.record()
                        lat[i].append((e0, e1))

        def run(n, f0, timed):
            # --sequences-threads: a host thread per sequence (measured: no better than one thread enqueuing round-robin —
            # 2 055 against 2 200 frames/s at S = 4 —: the threads share the interpreter lock between their calls)
            if threads and S > 1:
                import threading
                ts = [threading.Thread(target=one, args=(i, q, n, f0, timed)) for i, q in enumerate(seqs[:S])]
                [t.start() for t in ts]
                [t.join() for t in ts]
            else:
                for r in range(n):
                    for i, q in enumerate(seqs[:S]):
                        one(i, q, 1, f0 + r, timed)
        run(warmup, 0, False)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        run(rounds, warmup, False)
        t_enq = time.perf_counter() - t0  # the host is done enqueuing here; close to dt = the host is what paces the rounds
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        run(min(rounds, 30), warmup + rounds, True)
        torch.cuda.synchronize(device)
        ms = sorted(a.elapsed_time(b) for per in lat for a, b in per)
        t_err = max(float((q.solver.translations() - q.t_true[(warmup + rounds + min(rounds, 30) - 1) % q.n_frames]).abs().max()) for q in seqs[:S])
        out[str(S)] = dict(value=round(S * rounds / dt, 1), unit="frames/s (all sequences together)", sequences=S, rounds=rounds,
                           ms_per_round=round(dt / rounds * 1e3, 4), host_enqueue_ms_per_round=round(t_enq / rounds * 1e3, 4),
                           frame_latency_ms=dict(median=round(ms[len(ms) // 2], 4), p95=round(ms[min(len(ms) - 1, int(0.95 * len(ms)))], 4),
                                                 note="first to last launch of a frame on its sequence's solve stream, while the other sequences run"),
                           max_abs_translation_error_vs_ground_truth_m=round(t_err, 6))
    base = out[str(min(int(k) for k in out))]["value"]
    for k in out:
        out[k]["vs_one_sequence"] = round(out[k]["value"] / base, 3)
    out["note"] = ("secondary figure: S self-contained C2 sequences share one GPU (2 HIP streams each, chosen by a concurrency probe), "
                   + ("a host thread per sequence" if threads else "one host thread enqueues every launch (~45 per frame: its launch rate "
                                                                   "is part of what saturates)"))
    del seqs
    torch.cuda.empty_cache()
    return out


def live_depth_probe(cfg_name, device, steps=30, warmup=5, seq=None):
    """frames/s of the same configuration on the reference's data flow with noisy depth (SequenceLive), with what the
    solve went through: PCG iterations, the share of rows the Tukey weight rejects, energies."""
    import torch
    seq = seq or SequenceLive(cfg_name, device)
    for f in range(warmup):
        seq.frame(f)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for f in range(steps):
        seq.frame(warmup + f)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    st = seq.solver.stats()
    tw = seq.solver.tukey_weights()
    resid = (seq.warped - seq.live_last).norm(dim=1)
    out = dict(value=round(steps / dt, 2), unit="frames/s", steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 4),
               workload="%s, reference-parity energy on the reference's data flow: depth with sigma = 1 mm noise -> fused %d^3 "
                        "volume -> marching cubes (%d vertices, %d of them on the object, every %dth of those kept: %d rows) -> "
                        "nearest canonical vertex of every live vertex (dfa_correspond, %d canonical vertices) -> graph build "
                        "(%d nodes, k=%d) -> %d GN x PCG<=256 -> warp; one stream"
                        % (cfg_name, seq.cfg["dim"], seq.mc_total, seq.mc_object, seq.stride, seq.rows, seq.N, seq.D, seq.k,
                           seq.cfg["gn_iters"]),
               pcg_iterations_last_frame=st["pcg_iters"], gn_iterations_last_frame=st["gn_iters"],
               gn_iterations_noop_last_frame=st["gn_noop"], cost_first=st["initial_cost"], cost_last=st["final_cost"],
               tukey_rejected_fraction=round(float((tw == 0).float().mean()), 5),
               tukey_mean_weight=round(float(tw.mean()), 5),
               residual_after_warp_mm=dict(median=round(float(resid.median()) * 1e3, 3), p95=round(float(resid.quantile(0.95)) * 1e3, 3)),
               note="secondary figure: SURVEY 8(d)'s index-aligned zero-residual targets are the headline workload")
    return out


def cpu_quota():
    """CPUs' worth of time the cgroup grants this process (cpu.max of cgroup v2, cfs quota / period of v1), or None: on the
    GPU boxes of this pool nproc and the affinity mask say 256 while the quota is 16 — 256 OpenMP threads then time-slice
    each other on 16 CPUs' worth of time, which is what made round 5's `all_cores` figure 25 x slower than 16 threads"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = float(f.read())
        return quota / period if quota > 0 else None
    except (OSError, ValueError):
        return None


def cpu_baseline6(cfg_name, frames, params):
    """CPU statement of the north-star frame (oracle/solve6_oracle.c, double precision) on the host cores."""
    import oracle as O
    from dynfu_amd import synth
    cfg = synth.CONFIGS[cfg_name]
    threads = min(os.cpu_count() or 1, int(os.environ.get("DFA_CPU_THREADS", "16")))
    fx, fy, cx, cy = synth.intrinsics(cfg)
    voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
    dim, k = cfg["dim"], cfg["k"]
    c = synth.canonical(cfg)
    vol = np.zeros((dim, dim, dim), np.uint32)
    depths = [synth.depth_frame(cfg, f) for f in range(frames)]
    kw = {n: getattr(params, n) for n, _ in params._fields_}
    t0 = time.perf_counter()
    pcg = 0
    for f in range(frames):
        dists = O.compute_dists(depths[f], fx, fy, cx, cy)
        O.lib().orc_tsdf_clear(vol.ctypes.data, dim, dim, dim)
        O.tsdf_integrate(vol, dists, voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy, threads=threads)
        P, Nm = O.points_normals(depths[f], fx, fy, cx, cy)
        dq, st = O.solve6(c["node_pos"], c["node_dq"], c["node_w"], k, c["verts"], c["normals"], P, Nm, (fx, fy, cx, cy),
                          threads=threads, **kw)
        pcg += st["pcg_iters"]
    dt = time.perf_counter() - t0
    return dict(value=round(frames / dt, 4), unit="frames/s", cores=threads, kind="port",
                sample="%d full frames of config %s, north-star mode (compute_dists, clear, integrate %d^3, "
                       "computePointNormals, k-NN graphs, %d GN x block-Jacobi PCG (%d PCG iterations in total)) by the "
                       "C statement in oracle/solve6_oracle.c (fp64 solve), OpenMP over %d of the host's %d cores; %.1f s"
                       % (frames, cfg_name, dim, params.num_iter * params.gn_iter, pcg, threads, os.cpu_count() or 1, dt))


def pipelined_probe(seq, f0, device, steps=100, warmup=10):
    """Throughput of the same sequence with the graph build of frame f+1 (a function of the canonical vertices and node
    positions only: k-NN, transposition, record packing) overlapped with the solve of frame f on a third stream and a
    second solver plan (`--pipeline`).  Every frame does all of its work; only the order across frames changes.  A
    secondary figure: `value` above is the self-contained frame."""
    import torch
    seq.enable_pipeline()
    for f in range(warmup):
        seq.frame(f0 + f)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for f in range(steps):
        seq.frame(f0 + warmup + f)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    t_err = float((seq.solver.translations() - seq.t_true[(f0 + warmup + steps - 1) % seq.n_frames]).abs().max())
    return dict(value=round(steps / dt, 2), unit="frames/s", steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 4),
                streams="solve of frame f || fuse of frame f (behind its first assembly) || graph build of frame f+1 (behind its "
                        "second) on three HIP streams, two solver plans",
                max_abs_translation_error_vs_ground_truth_m=round(t_err, 6))


def northstar_rooflines(seq, config, st, tm, fuse_ms):
    """roofline entries of the north-star kernels from the hipEvent timings of the last timed frame (dfa_solver6_get_timing:
    events on the solve's stream around every launch group) — the dominant one first.  Algorithmic bytes:
    linearise reads a vertex (canon 12 + normal 12 + k indices and weights 8 k + the live pixel 32 bytes; the k node
    transforms come from L2) and writes ONE record per vertex (l 32 + h 4 K + 16 bytes, K = 4 or 8 the kernels' template);
    the assembly reads that record once per node it touches (k times: 48 + 4 K bytes + 4 of the row's list entry), a 4-byte
    pair record per (row, neighbour) of slot 0 and of the upper blocks (1 + (k - 1) / 2 per row), and writes the block matrix
    (36 floats per block, the upper half computed and mirrored); a PCG iteration reads the matrix (36 floats + a column id
    per block), three gathered 6-vectors per block and ~12 vectors of 6 D floats."""
    cfg = seq.cfg
    dim, Wd, Hd, k = cfg["dim"], cfg["width"], cfg["height"], seq.k
    V = dim ** 3
    # (the event brackets cover every Gauss-Newton slot the host enqueued; the slots behind the end of an outer iteration are
    # launches that return at entry: the averages are over the linearisations evaluated / the normal equations solved)
    nblk = tm["matrix_blocks"]
    stop = st["stop_hist"][:seq.gn_total]
    gn_lin = max(1, sum(1 for c in stop if c != 3))
    gn = max(1, sum(1 for c in stop if c == 0))
    launches = max(1, st["pcg_launches"])
    fuse_bytes = 4.0 * V + 2.0 * Wd * Hd
    kk = 4 if k <= 4 else 8
    lin_bytes = seq.N * (24 + 8 * k + 32) + seq.N * (48 + 4 * kk)
    asm_bytes = seq.N * k * (52 + 4 * kk) + seq.N * k * (1 + (k - 1) / 2.0) * 4 + nblk * 36 * 4
    pcg_bytes = nblk * (36 * 4 + 4 + 3 * 24) + 12 * 24.0 * seq.D
    asm_ms, lin_ms, pcg_ms = tm["assemble_ms"] / gn, tm["linearise_ms"] / gn_lin, tm["pcg_ms"] / launches

    # SURVEY 8(d)'s own prices (what the PROBLEM needs, independent of this implementation's data flow): per Gauss-Newton
    # iteration N x (12 canonV + 12 normal + 12 liveV + 4 tau + 4 k idx) + D x 28 in — charged to the linearisation, which
    # reads the inputs — and nnz_blocks x 36 x 4 out — charged to the assembly, which writes the matrix; per PCG iteration
    # nnz_blocks x b^2 x 4 + 6 n x 4 with b = 6, n = 6 D.  `frac` is computed from THESE; the bytes this design chose to
    # move (records re-read once per neighbour through the transposed lists, pair records) stay beside them as
    # implementation_bytes_per_launch, and traffic_over_survey = counter traffic / survey bytes.
    sv_lin = seq.N * (12 + 12 + 12 + 4 + 4.0 * k) + 28.0 * seq.D
    sv_asm = nblk * 36 * 4.0
    sv_pcg = nblk * 36 * 4.0 + 6 * 6 * seq.D * 4.0

    def entry(kernel, key, ms, survey, impl, per_frame, total_ms, **extra):
        gbs = survey / (ms * 1e-3) / 1e9 if ms > 0 else float("nan")
        src = PMC_TRAFFIC_BYTES.get((config, key))
        return dict(kernel=kernel, bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(gbs / HBM_PEAK_GBS, 4), traffic=src, traffic_source=traffic_source(config, key),
                    avg_launch_ms=round(ms, 5), launches_per_frame=per_frame, survey_bytes_per_launch=survey,
                    implementation_bytes_per_launch=impl,
                    implementation_frac=round(impl / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None,
                    traffic_over_survey=round(src / survey, 2) if src else None,
                    ms_per_frame=round(total_ms, 4), **extra)

    pair_ms = asm_ms + lin_ms
    ents = [
        entry("s6_assemble2_kernel<%d,%d> (block normal matrix of one Gauss-Newton iteration)" % (kk, 448 if kk == 4 else 352),
              "s6_assemble", asm_ms, sv_asm, asm_bytes, gn, tm["assemble_ms"],
              note="a lane per work unit walks its share of one block's (row, neighbour) list and owns the 36 distinct entries "
                   "of its 8 x 8 moment; rows staged by LDS-DMA.  Bound by instruction issue and the per-workgroup chain of "
                   "barriers and round trips (SQ counters, phase clocks: DESIGN.md 4.5), not by HBM"),
        entry("s6_linearise_kernel<%d> (+ s6_nodes, s6_reg: residuals and row factors of one Gauss-Newton iteration)" % kk,
              "s6_linearise", lin_ms, sv_lin, lin_bytes, gn_lin, tm["linearise_ms"]),
        entry("s6_pcg_step_kernel (one Chronopoulos-Gear PCG iteration per launch)", "s6_pcg_step", pcg_ms, sv_pcg, pcg_bytes, launches,
              tm["pcg_ms"], matrix_blocks=nblk, launches_without_an_iteration=max(0, st["pcg_launches"] - st["pcg_iters"]),
              note="launch/latency-bound below ~2k nodes (two dependent memory round trips + the inter-kernel gap); a launch "
                   "whose PCG has converged costs ~3.5 us, which is why the launch count follows the iteration count"),
        entry("integrate_runs_kernel<FUSED_CLEAR,32,8> (clear+integrate %d^3)" % dim, "fused_integrate", fuse_ms, fuse_bytes, fuse_bytes, 1, fuse_ms),
    ]
    pair_traffic = (PMC_TRAFFIC_BYTES.get((config, "s6_assemble")) or 0) + (PMC_TRAFFIC_BYTES.get((config, "s6_linearise")) or 0)
    ents.append(dict(kernel="residual / Jacobian assembly of one Gauss-Newton iteration = s6_linearise + s6_assemble2 (SURVEY 8d prices "
                            "the pair as one piece)", bound="hbm",
                     achieved=round((sv_lin + sv_asm) / (pair_ms * 1e-3) / 1e9, 1) if pair_ms > 0 else None, peak=HBM_PEAK_GBS,
                     unit="GB/s", frac=round((sv_lin + sv_asm) / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pair_ms > 0 else None,
                     traffic=pair_traffic or None, avg_launch_ms=round(pair_ms, 5), launches_per_frame=gn,
                     survey_bytes_per_launch=sv_lin + sv_asm, implementation_bytes_per_launch=lin_bytes + asm_bytes,
                     traffic_over_survey=round(pair_traffic / (sv_lin + sv_asm), 2) if pair_traffic else None,
                     ms_per_frame=0.0))  # (ms_per_frame 0: a derived entry, sorted last)
    ents.sort(key=lambda e: -e["ms_per_frame"])
    return ents


def northstar_fields(seq, st):
    # history slots: outer x gn_iter + gn (+ the closing check).  stop_hist 0: linearised and solved; 1: converged there; 2: the
    # step before it was rejected there; 3: skipped.  cost_per_gn lists the ACCEPTED linearisations (0, 1) in order.
    stop = st["stop_hist"]
    solved = [i for i, c in enumerate(stop) if c == 0]
    accepted = [i for i, c in enumerate(stop) if c in (0, 1)]
    return dict(gn_iterations=st["gn_iters"], gn_solves=st["gn_solves"], gn_steps_rejected=st["gn_rejected"],
                gn_outer_iterations_converged=st["gn_converged"], gn_tol=round(float(seq.params.gn_tol), 6),
                gn_slots="".join("sCRx"[c] for c in stop) + "  (s solved, C converged, R step rejected and undone, x skipped; "
                         "%d outer x %d%s)" % (seq.params.num_iter, seq.params.gn_iter, " + closing check" if seq.params.gn_tol > 0 else ""),
                pcg_iterations=st["pcg_iters"], pcg_iterations_per_gn=[st["pcg_it_hist"][i] for i in solved],
                pcg_iteration_cap=seq.params.linear_iter,
                pcg_relative_residual_per_gn=[round(st["pcg_rel_hist"][i], 5) for i in solved],
                pcg_tolerance_per_gn=[round(st["pcg_tol_hist"][i], 5) for i in solved],
                pcg_tolerance_schedule=("Eisenstat-Walker: %g first, then clamp(%g x (r.z)_0,i / (r.z)_0,i-1, %g, %g)" %
                                        (seq.pcg["pcg_tol_first"], seq.pcg["pcg_tol_adapt"], seq.pcg["pcg_tol"], seq.pcg["pcg_tol_first"])
                                        if seq.pcg.get("pcg_tol_adapt", 0) > 0 else
                                        "max(%g, %g x %g^i) at Gauss-Newton iteration i of an outer iteration" %
                                        (seq.pcg["pcg_tol"], seq.pcg["pcg_tol_first"], seq.pcg["pcg_tol_decay"])),
                pcg_launches=st["pcg_launches"], pcgs_cut_short_by_the_launch_budget=st["pcg_short"],
                valid_rows=st["valid_last"], final_cost=float("%.5g" % st["final_cost"]),
                cost_per_gn=[float("%.5g" % st["cost_hist"][i]) for i in accepted],
                cost_of_rejected_steps=[float("%.5g" % st["cost_hist"][i]) for i, c in enumerate(stop) if c == 2],
                cost_per_gn_by_outer_iteration=[[float("%.5g" % st["cost_hist"][i]) for i in accepted
                                                 if i // seq.params.gn_iter == o and i < seq.gn_total]
                                                for o in range(seq.params.num_iter)],
                valid_rows_per_gn=[st["valid_hist"][i] for i in accepted],
                cost_per_valid_row_per_gn=[float("%.4g" % (st["cost_hist"][i] / max(1, st["valid_hist"][i]))) for i in accepted])


def northstar_fixed_iterations(seq, f0, device, steps, warmup=3):
    """the same frames (f0 + warmup .. f0 + warmup + steps: every frame starts from the canonical transforms, so a frame's
    work does not depend on what ran before it) with gn_tol = 0 — every Gauss-Newton iteration runs, what rounds 1-4
    measured — beside the default (stopping rule on).  The plan's launch budget restarts by itself when the stopping rule
    changes."""
    import torch
    A = seq.A
    keep = seq.params
    kw = {n: getattr(keep, n) for n, _ in keep._fields_}
    kw["gn_tol"] = 0.0
    seq.params = A.Solve6Params(**kw)
    for f in range(warmup):
        seq.frame(f0 + f)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for f in range(steps):
        seq.frame(f0 + warmup + f)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    st = seq.solver.stats()
    f = northstar_fields(seq, st)
    seq.params = keep
    return dict(value=round(steps / dt, 2), unit="frames/s", steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 4),
                **{k: f[k] for k in ("gn_solves", "pcg_iterations", "pcg_launches", "final_cost", "cost_per_gn", "valid_rows_per_gn",
                                     "cost_per_valid_row_per_gn")})


def northstar_timed_frames(seq, f0, device, frames=5):
    """`frames` more frames with hipEvent brackets on the solve's stream around every launch group (outside any timed
    region); per-kernel times are the MEDIAN over those frames (one frame alone moves by 2x with where the concurrent
    volume sweep happens to land)."""
    import torch
    fuse_events, tms = [], []
    seq.solver.enable_timing(True)
    for i in range(frames):
        seq.frame(f0 + i, None, fuse_events)
        torch.cuda.synchronize(device)
        tms.append(seq.solver.timing())
    st = seq.solver.stats()
    seq.solver.enable_timing(False)
    tm = dict(tms[-1])
    for key in ("linearise_ms", "assemble_ms", "pcg_ms"):
        tm[key] = float(np.median([t[key] for t in tms]))
    fuse_ms = float(np.median([a.elapsed_time(b) for a, b in fuse_events]))
    return st, tm, fuse_ms


def northstar_probe(cfg_name, device, steps=30, warmup=8, cpu_frames=0):
    """The same frame in north-star mode (6-DoF DQ-blend / projective point-to-plane / ARAP solve), timed
    on this GPU after the main measurement: a secondary figure of the default bench line, with its own rooflines."""
    import torch
    seq = Sequence6(cfg_name, device, 64)
    seq.fuse_first = False
    for f in range(warmup):
        seq.frame(f)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for f in range(steps):
        seq.frame(warmup + f)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    st, tm, fuse_ms = northstar_timed_frames(seq, warmup + steps, device)
    rl = northstar_rooflines(seq, cfg_name, st, tm, fuse_ms)
    fixed = northstar_fixed_iterations(seq, warmup - 3, device, steps)
    out = dict(value=round(steps / dt, 2), unit="frames/s", steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 4),
               workload="%s north-star mode: at most %d GN iterations (stopping rule gn_tol = %g) x block-Jacobi PCG (inexact Newton), "
                        "6-DoF twists per node, DQ blend, projective point-to-plane data term against the live depth map, ARAP "
                        "regulariser; same fuse" % (cfg_name, seq.gn_total, seq.params.gn_tol),
               solve=northstar_fields(seq, st), fixed_iterations=fixed, roofline=rl[0], roofline_other=rl[1:],
               note="parity unpinned (the reference has no such solve): checked against the fp64 statement oracle/solve6_oracle.c")
    params = seq.params
    del seq
    torch.cuda.empty_cache()
    if cpu_frames > 0:
        out["cpu_baseline"] = cpu_baseline6(cfg_name, cpu_frames, params)
    return out


def main_northstar(args, torch, replicas, rank, world, device):
    """bench line of the north-star mode (same contract; per-kernel figures from hipEvent timings of the phases of the
    last timed frame)."""
    n_gpus = ranks_seen(device)
    lin = args.linear_iter or 64
    pcg = dict(NS_PCG)
    if args.no_adaptive_launch:
        pcg["adaptive_launch"] = 0
    if args.forcing == "geometric":
        pcg["pcg_tol_adapt"] = 0.0
    if args.gn_tol is not None:
        pcg["gn_tol"] = args.gn_tol
    seq = Sequence6(args.config, device, lin, pcg)
    seq.fuse_first = args.fuse_first
    seq.overlap = args.overlap or args.fuse_first
    seq.warm_start = bool(args.warm_start)
    cfg = seq.cfg
    K, Wm = args.steps, args.warmup
    ser = True if args.serial else None  # (None: the mode's default — stream order unless --overlap / --fuse-first)
    for f in range(Wm):
        seq.frame(f, ser)
    fuse_events = []

    def timed():
        # (the per-phase hipEvents only around the LAST timed frame: an event costs ~5 us of stream time, and a C3 frame has
        # sixty of them — bracketed on every frame the line read 325 frames/s where the same frames run at 410)
        for f in range(K):
            if f == K - 1:
                seq.solver.enable_timing(True)
            seq.frame(Wm + f, ser, fuse_events if f % TIMING_SAMPLE == 0 else None)

    dt_max, regions = median_region(replicas, timed, device, args.repeats)
    st = seq.solver.stats()
    tm = seq.solver.timing()  # hipEvents on the solve stream around every launch group of the LAST timed frame
    seq.solver.enable_timing(False)
    if rank != 0:
        replicas.shutdown()
        return
    dim, Wd, Hd = cfg["dim"], cfg["width"], cfg["height"]
    fuse_ms = float(np.mean([a.elapsed_time(b) for a, b in fuse_events])) if fuse_events else float("nan")
    rl = northstar_rooflines(seq, args.config, st, tm, fuse_ms)
    out = dict(metric="frames/sec (warp-solve + TSDF fuse), 512^3 vol / 2k nodes / VGA depth",
               value=round(n_gpus * K / dt_max, 2), unit="frames/s", n_gpus=n_gpus, steps=K, warmup=Wm,
               ms_per_step=round(dt_max / K * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f32", data="synthetic",
               config=dict(workload="%s north-star: %d^3 TSDF, %dx%d depth, %d nodes, k=%d, %d vertices, <=%d GN (gn_tol %g) x "
                                    "block-Jacobi PCG<=%d, 6-DoF / point-to-plane / ARAP"
                                    % (args.config, dim, Wd, Hd, seq.D, seq.k, seq.N, seq.gn_total, seq.params.gn_tol, lin),
                           workload_detail="inexact Newton: %s; DQ blend of k-NN nodes, projective association, lambda=200"
                                           % northstar_fields(seq, st)["pcg_tolerance_schedule"],
                           parallelism="replicas x%d (one sequence per GPU, no collective)" % n_gpus, ranks_seen=n_gpus,
                           rccl_selfcheck=rccl_selfcheck(), repeats=len(regions), launcher=launcher_mark(),
                           streams="one HIP stream (every kernel of this mode fills the chip: a second stream for the sweep costs 6-8 %)"
                                   if not seq.overlap or args.serial else ("fuse || graph build + solve on two HIP streams" if args.fuse_first else
                                                                           "graph build, then fuse || solve on two HIP streams"),
                           last_frame=northstar_fields(seq, st)),
               region_ms=[round(r * 1e3, 4) for r in regions], roofline=rl[0], roofline_other=rl[1:])
    if not args.no_cpu_baseline and world == 1:
        params = seq.params
        del seq
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline6(args.config, max(1, args.cpu_frames // 2), params)
    finish(out, args)
    replicas.shutdown()


def end_to_end(cfg_name, frames=14, skip=4):
    """The reference's OWN timed region (src/apps/demo.cpp:90-95: the time inside `(*dynfu)(depth)`, sequence
    src/dynfu/dyn_fusion.cpp:48-145) through the C++ adaptor classes: DynFusion::operator() — bilateral filter, dists,
    clear + integrate, marching cubes, warp, correspondence, graph build, solve, node insertion — over the synthetic
    depth sequence at this configuration's volume and image size, in the reference's mode and with the north-star solve.
    Run by dynfu_amd/host/build/sequence_bench (a child process: its own HIP context; this process is idle meanwhile);
    the vertices are whatever marching cubes extracts (~1 M at 512^3), the nodes what the seeding rule and the insertion
    make of them — NOT the fixed 262 144 / 2 048 of the figure above."""
    import subprocess
    import tempfile

    from dynfu_amd import build as B, synth
    exe = B.SEQ_BENCH
    if not os.path.exists(exe):
        return dict(error="dynfu_amd/host/build/sequence_bench is not built (python -c 'import __graft_entry__ as g; g.build()')")
    cfg = synth.CONFIGS[cfg_name]
    W, H, dim = cfg["width"], cfg["height"], cfg["dim"]
    out = dict(unit="ms per frame inside DynFusion::operator() (device synchronised inside the timed call)",
               sequence="%d synthetic %dx%d depth frames (dynfu_amd/synth.py), %d^3 volume; the first %d frames (seeding, "
                        "plan and scratch allocation, node insertion settling) are not counted" % (frames, W, H, dim, skip))
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "frames.u16")
        np.stack([synth.depth_frame(cfg, f) for f in range(frames)]).astype("<u2").tofile(raw)
        for mode in ("ref", "northstar"):
            r = subprocess.run([exe, raw, str(W), str(H), str(frames), str(dim), mode], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                out[mode] = dict(error=r.stderr[-500:])
                continue
            rec = json.loads(r.stdout.strip().splitlines()[-1])
            ms = sorted(rec.pop("frame_ms")[skip:])
            rec.update(median_ms=round(ms[len(ms) // 2], 3), p95_ms=round(ms[min(len(ms) - 1, int(0.95 * len(ms)))], 3),
                       frames_per_s=round(1e3 / ms[len(ms) // 2], 1), frames_measured=len(ms))
            out[mode] = rec
    return out


def host_cores():
    """cores this process may run on (what `nproc` prints), and the machine's count"""
    try:
        return len(os.sched_getaffinity(0)), os.cpu_count() or 1
    except (AttributeError, OSError):
        return os.cpu_count() or 1, os.cpu_count() or 1


CPU_SWEEP_THREADS = (8, 16, 32, 64)


def cpu_baseline(cfg_name, frames, variants=True):
    """The CPU restatement (oracle/, kind "port": NOT Ceres, NOT the reference's CUDA path — neither exists for this path in
    a buildable form) timed on the host cores for `frames` frames of the same workload PER THREAD COUNT.  The thread count
    is swept, not guessed (8 / 16 / 32 / 64, whichever this process may use): `value` is the best of them, `cores` its thread
    count, `by_threads` the whole sweep.  SURVEY 8(d) asks for two more figures — (a) a single thread, (b) OpenMP over all
    host cores (`nproc` of this box, stated) —: `single_thread`, `all_cores`.  Also returns the oracle's node translations
    of the sample's last frame (a checker for the line's own solve of that frame)."""
    import oracle as O
    from dynfu_amd import synth
    cfg = synth.CONFIGS[cfg_name]
    nproc, machine = host_cores()
    quota = cpu_quota()
    usable = max(1, min(nproc, int(quota + 0.5))) if quota else nproc  # CPUs this process can really keep busy
    forced = os.environ.get("DFA_CPU_THREADS")
    counts = [int(forced)] if forced else sorted({min(t, nproc) for t in CPU_SWEEP_THREADS} | {usable})
    fx, fy, cx, cy = synth.intrinsics(cfg)
    voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
    dim, k = cfg["dim"], cfg["k"]
    c = synth.canonical(cfg)
    vol = np.zeros((dim, dim, dim), np.uint32)
    depths = [synth.depth_frame(cfg, f) for f in range(frames)]
    idx = O.knn(c["node_pos"], c["verts"], k, threads=min(usable, 16))
    w = np.zeros(idx.shape, np.float32)
    d2 = ((c["verts"][:, None, :].astype(np.float64) - c["node_pos"][idx].astype(np.float64)) ** 2).sum(-1)
    w = np.exp(-d2 / (2 * float(c["node_w"][0]) ** 2)).astype(np.float32)
    lives = [synth.live_vertices(c["verts"], idx, w, synth.true_translations(c["node_pos"], f, cfg["k"])) for f in range(frames)]

    def run(nthreads, nframes):
        t0 = time.perf_counter()
        pcg, t_last = 0, None
        for f in range(nframes):
            dists = O.compute_dists(depths[f], fx, fy, cx, cy)
            O.lib().orc_tsdf_clear(vol.ctypes.data, dim, dim, dim)
            O.tsdf_integrate(vol, dists, voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy, threads=nthreads)
            t_last, dq, st = O.solve_ref(c["node_pos"], c["node_dq"], c["node_w"], k, c["verts"], lives[f],
                                         num_iter=cfg["gn_iters"], nonlinear_iter=1, linear_iter=256, pcg_tol=1e-6,
                                         use_double=False, threads=nthreads, **synth.SOLVER)
            O.warp_to_live(c["node_pos"], dq, c["node_w"], k, c["verts"], c["normals"], threads=nthreads)
            pcg += st["pcg_iters"]
        return time.perf_counter() - t0, pcg, t_last

    t_total = time.perf_counter()
    by, t_last, pcg = {}, None, 0
    for n in counts:
        O.tsdf_integrate(vol[:8], O.compute_dists(depths[0], fx, fy, cx, cy), voxel, trunc, 64, vol2cam, fx, fy, cx, cy,
                         threads=n)  # warm the thread pool at this width
        dt, pcg, t_last = run(n, frames)
        by[str(n)] = dict(value=round(frames / dt, 4), seconds=round(dt, 2))
    best = max(by, key=lambda n: by[n]["value"])
    out = dict(value=by[best]["value"], unit="frames/s", cores=int(best), kind="port", by_threads=by,
               sample="%d full frames of config %s per thread count (compute_dists, clear, integrate %d^3, k-NN graph, %d GN x PCG "
                      "(%d PCG iterations in total), write-back, warpToLive) by the C restatement in oracle/, fp32, OpenMP over "
                      "%s threads of the host's %d cores (affinity mask %d, cgroup quota %s CPUs): `value` is the best of the sweep"
                      % (frames, cfg_name, dim, cfg["gn_iters"], pcg, " / ".join(by), machine, nproc, "%.0f" % quota if quota else "none"),
               sample_short="%d frames of %s per thread count (%s), C restatement in oracle/ (not Ceres), fp32, OpenMP; best reported"
                            % (frames, cfg_name, "/".join(by)),
               host_cores=dict(nproc=nproc, machine=machine, cgroup_quota_cpus=quota, usable=usable))
    if variants:
        dt1, _, _ = run(1, 1)
        out["single_thread"] = dict(value=round(1 / dt1, 4), unit="frames/s", cores=1, sample="1 frame, %.1f s" % dt1)
        if str(usable) in by:
            out["all_cores"] = dict(by[str(usable)], unit="frames/s", cores=usable, sample="the %d-thread figure of the sweep IS all cores here" % usable)
        else:
            dta, _, _ = run(usable, 1)
            out["all_cores"] = dict(value=round(1 / dta, 4), unit="frames/s", cores=usable, sample="1 frame, %.1f s" % dta)
        # `all_cores` = every CPU this process can keep busy: the cgroup quota where there is one (threads beyond it only
        # time-slice each other), else the affinity mask
        out["cores_note"] = (("cgroup quota: %.0f CPUs' worth of time on a %d-thread host (threads beyond it time-slice each other); "
                              % (quota, machine) if quota and quota < nproc else "") +
                             "one short OpenMP region per PCG iteration, then a serial scatter")
    out["seconds_total"] = round(time.perf_counter() - t_total, 1)
    return out, t_last, frames - 1


def fuse_variants_probe(seq, reps=30):
    """The fused clear + integrate sweep of the configuration, alone on the device, three ways: every voxel stored (the
    headline's), with the occupancy map kept beside it (dfa_tsdf_clear_integrate_occ: what a marching-cubes pass behind it
    wants), and over a map KNOWN to describe the volume (dfa_tsdf_clear_integrate_known_occ: boxes of zeros that stay zeros are
    not stored again).  Same volume bits each way (tests/test_gpu_mc.py); events on the launch stream."""
    import torch
    A, synth = seq.A, seq.synth
    vol = torch.zeros_like(seq.vol)
    occ = A.tsdf_occupancy(vol)
    dists = []
    for f in range(4):
        d = torch.empty_like(seq.dists)
        A.compute_dists(seq.depth[f % seq.n_frames], d, *seq.intr)
        dists.append(d)
    out = {}
    for name, kw in (("every_voxel_stored", {}), ("occupancy_kept", dict(occupancy=occ)),
                     ("occupancy_known", dict(occupancy=occ, occupancy_known=True))):
        if name == "occupancy_known":
            A.tsdf_clear(vol, occupancy=occ)
        run = lambda f: A.tsdf_clear_integrate(vol, dists[f % 4], seq.voxel, seq.trunc, synth.MAX_WEIGHT, seq.vol2cam, *seq.intr, **kw)
        for f in range(4):
            run(f)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for f in range(reps):
            run(f)
        e1.record()
        torch.cuda.synchronize()
        out[name + "_ms"] = round(e0.elapsed_time(e1) / reps, 4)
    out["boxes_without_weights"] = round(float((occ == 0).float().mean()), 4)
    out["note"] = ("per call: the tile pass of the dists image + the sweep; `value` above is measured with every voxel stored; the "
                   "live-depth flow and the C++ TsdfVolume use the map")
    return out


def raycast_probe(seq, config, reps=20):
    """"+ raycast, reported separately".  Both variants of the raycast (src/kfusion/tsdf_volume.cpp:95-129,
    tsdf_volume.cu:128-386) through the volume the timed frames have just fused, from the integration pose, timed with
    events on the launch stream; priced the way SURVEY 8(d) says — rays x steps x 4 B + hits x 64 x 4 B + 32 W H out — with
    the step / hit / distinct-voxel counts of these very rays (dfa_tsdf_raycast_tally)."""
    import torch
    A, synth, cfg = seq.A, seq.synth, seq.cfg
    W, H, dim = cfg["width"], cfg["height"], cfg["dim"]
    pts = torch.empty((H, W, 4), dtype=torch.float32, device=seq.vol.device)
    nrm = torch.empty_like(pts)
    dep = torch.empty((H, W), dtype=torch.uint16, device=seq.vol.device)
    args = (seq.vol, seq.voxel, seq.trunc, seq.cam2vol, seq.rinv, *seq.intr, synth.RAYCAST_STEP_FACTOR, synth.GRADIENT_DELTA_FACTOR)
    tally = A.tsdf_raycast_tally(*args, W, H, unique=True)
    fetch_bytes = 4.0 * (tally["march_fetches"] + tally["trilinear_fetches"])
    out = dict(work=tally, survey_formula="rays x steps x 4 B + hits x 64 x 4 B + 32 W H out (depth variant: 18 W H out)")
    for name, fn, tail, out_bytes in (("points", A.tsdf_raycast_points, (pts, nrm), 32.0 * W * H),
                                      ("depth", A.tsdf_raycast_depth, (dep, nrm), 18.0 * W * H)):
        for _ in range(3):
            fn(*args, *tail)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record()
            fn(*args, *tail)
            e1.record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        nbytes = fetch_bytes + out_bytes
        gbs = nbytes / (ms * 1e-3) / 1e9
        low = 4.0 * tally["unique_voxels"] + out_bytes
        key = "raycast_" + name
        out[name] = dict(kernel="raycast_%s_kernel (a ray per lane, 8 x 8-pixel tile per wave)" % name, bound="hbm",
                         achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                         traffic=PMC_TRAFFIC_BYTES.get((config, key)), traffic_source=traffic_source(config, key),
                         avg_launch_ms=round(ms, 4), launches=reps, survey_bytes_per_launch=nbytes,
                         unique_voxel_bytes_per_launch=low, frac_of_peak_on_unique_bytes=round(low / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         gathers_per_us=round((tally["march_fetches"] + tally["trilinear_fetches"]) / (ms * 1e3), 1),
                         note="a gather kernel: every fetch is a 4-byte read of its own; what bounds it is the rate of cache-line "
                              "requests and the dependent chain of march steps, not HBM (DESIGN.md 4.1)")
    return out


def config_probe(cfg_name, mode, device, steps=10, warmup=3, n_frames=6):
    """One of the OTHER BASELINE configurations (C3: 512^3 / 4 k nodes / k = 8 / 10 GN, "+ ARAP" = north-star mode; C4:
    1024^3 / 8 k nodes / 720p) under the same clock as the headline: `steps` self-contained frames after `warmup`,
    device synchronised on both sides; reference-parity energy (`ref`) or the 6-DoF solve (`northstar`).  Short on
    purpose (the default run must finish in minutes): treat single-digit-percent differences as noise."""
    import torch
    t_build = time.perf_counter()
    if mode == "northstar":
        seq = Sequence6(cfg_name, device, 64, n_frames=n_frames)
        seq.fuse_first = False
    else:
        seq = Sequence(cfg_name, device, n_frames=n_frames)
        seq.fuse_first = False
        if seq.D <= 2048:
            seq.enable_pcg_shadow()
    t_build = time.perf_counter() - t_build
    for f in range(warmup):
        seq.frame(f)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for f in range(steps):
        seq.frame(warmup + f)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    cfg = seq.cfg
    out = dict(value=round(steps / dt, 2), unit="frames/s", steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 4),
               frames="frames %d..%d of a cycle over the first %d frames of the synthetic sequence: the surface is within %.0f degrees "
                      "of its %d-frame period of the canonical one (every frame is solved from the canonical state, so later frames "
                      "are larger deformations: `--mode northstar --config %s` walks the whole period and reads lower)"
                      % (warmup, warmup + steps - 1, n_frames, 360.0 * (n_frames - 1) / seq.synth.N_FRAMES, seq.synth.N_FRAMES, cfg_name),
               workload="%s %s: %d^3 TSDF, %dx%d depth, %d nodes, k=%d, %d vertices, %d GN iterations"
                        % (cfg_name, "north-star mode (6-DoF / projective point-to-plane / ARAP)" if mode == "northstar" else
                           "reference-parity energy (energy.t)", cfg["dim"], cfg["width"], cfg["height"], seq.D, seq.k, seq.N,
                           cfg["gn_iters"]), setup_s=round(t_build, 1))
    if mode == "northstar":
        st, tm, fuse_ms = northstar_timed_frames(seq, warmup + steps, device, frames=3)
        rl = northstar_rooflines(seq, cfg_name, st, tm, fuse_ms)
        f = northstar_fields(seq, st)
        out.update(solve={k: f[k] for k in ("gn_iterations", "gn_solves", "gn_steps_rejected", "gn_outer_iterations_converged", "gn_tol",
                                            "gn_slots", "pcg_iterations", "pcg_launches", "pcgs_cut_short_by_the_launch_budget",
                                            "valid_rows", "final_cost", "cost_per_gn", "cost_of_rejected_steps", "valid_rows_per_gn",
                                            "cost_per_valid_row_per_gn")},
                   early_out=dict(value=out["value"], ms_per_step=out["ms_per_step"]),
                   fixed_10=northstar_fixed_iterations(seq, warmup - 3, device, steps), roofline=rl[0], roofline_other=rl[1:])
    else:
        st = seq.solver.stats()
        t_err = float((seq.solver.translations() - seq.t_true[(warmup + steps - 1) % seq.n_frames]).abs().max())
        out.update(pcg_iterations_last_frame=st["pcg_iters"], gn_iterations_last_frame=st["gn_iters"],
                   gn_iterations_noop_last_frame=st["gn_noop"], max_abs_translation_error_vs_ground_truth_m=round(t_err, 6))
    del seq
    torch.cuda.empty_cache()
    return out


def other_configs(device, cpu=True):
    out = {}
    # C1 IS the CPU-path configuration of BASELINE.json ("256^3, ~500 nodes, Ceres CPU solver"): its line carries the CPU
    # restatement timed beside it (the umbrella sequence itself is not in the image: the same synthetic scene at C1's sizes)
    try:
        out["C1_ref"] = config_probe("C1", "ref", device, steps=20, warmup=5)
        if cpu:
            out["C1_ref"]["cpu_baseline"] = cpu_baseline("C1", 4)[0]
    except Exception as e:  # noqa: BLE001
        out["C1_ref"] = dict(error="%s: %s" % (type(e).__name__, e))
    for name in ("C3", "C4"):
        for mode in ("ref", "northstar"):
            try:
                out["%s_%s" % (name, mode)] = config_probe(name, mode, device)
            except Exception as e:  # noqa: BLE001
                out["%s_%s" % (name, mode)] = dict(error="%s: %s" % (type(e).__name__, e))
    return out


_LINE_OUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too — RCCL prints its version block through C stdio
    when its communicator goes away, i.e. AFTER the line (seen on the first run of the one-rank self-check) — so this
    process keeps the real stdout for the line alone and points file descriptor 1 at stderr for everything else."""
    global _LINE_OUT
    if _LINE_OUT is None:
        sys.stdout.flush()
        _LINE_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    if _LINE_OUT is None:
        print(line, flush=True)
    else:
        _LINE_OUT.write(line + "\n")
        _LINE_OUT.flush()


LINE_LIMIT = 4096   # bytes of the stdout line (r05's 28 KB line outgrew the driver's reader: BENCH_r05.parsed == null)
STRING_LIMIT = 160  # characters of any string inside it


def _num(x, digits=4):
    """a bare number for the line (None where the figure is missing or failed)"""
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits + 2, x)) if abs(x) < 1 else round(x, 2)


def _val(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _short(s, n=STRING_LIMIT):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _slim_roofline(e):
    """the contract's roofline object of one kernel: the HBM view first (frac = algorithmic bytes / launch time / 8 TB/s),
    the roof that actually binds the kernel - where that is not HBM - as `other_view`"""
    if not isinstance(e, dict):
        return None
    alg = e.get("algorithmic_bytes_per_launch", e.get("survey_bytes_per_launch"))
    out = dict(kernel=_short(str(e.get("kernel", "")).split(" (")[0], 64), bound=e.get("bound"), achieved=e.get("achieved"),
               peak=e.get("peak"), unit=e.get("unit"), frac=e.get("frac"), traffic=e.get("traffic"),
               algorithmic_bytes_per_launch=alg,
               traffic_over_algorithmic=(round(e["traffic"] / alg, 3) if e.get("traffic") and alg else None),
               avg_launch_ms=e.get("avg_launch_ms"), launches_per_frame=e.get("launches_per_frame"))
    if isinstance(e.get("lds_gather_view"), dict):
        v = e["lds_gather_view"]
        out["other_view"] = dict(bound=v.get("bound"), achieved=v.get("achieved"), peak=v.get("peak"), frac=v.get("frac"))
    return out


def contract_line(out, detail_file):
    """The ONE stdout line: the contract's fields, `roofline` of the dominant kernel (HBM view), `cpu_baseline` as numbers,
    every secondary figure as a bare number — at most LINE_LIMIT bytes, no string above STRING_LIMIT characters.  Everything
    else (per-kernel tables, notes, histories) is in the detail file."""
    cfg = out.get("config", {})
    sc = cfg.get("rccl_selfcheck") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = dict(workload=_short(cfg.get("workload", "")), parallelism=_short(cfg.get("parallelism", ""), 80),
                          ranks_seen=cfg.get("ranks_seen"), repeats=cfg.get("repeats"),
                          rccl_selfcheck=({k: sc.get(k) for k in ("ok", "init_ms", "backend") if k in sc} if "ok" in sc
                                          else dict(skipped=True)),
                          launcher=cfg.get("launcher"))
    for k in ("max_abs_translation_error_vs_ground_truth_m", "max_abs_translation_diff_vs_oracle_m"):
        if k in cfg:
            line["config"][k] = cfg[k]
    line["roofline"] = _slim_roofline(out.get("roofline"))
    others = [e for e in out.get("roofline_other", []) if isinstance(e, dict)]
    if others:
        line["roofline_other"] = [{k: v for k, v in _slim_roofline(e).items()
                                   if k in ("kernel", "frac", "achieved", "avg_launch_ms", "launches_per_frame",
                                            "traffic_over_algorithmic")} for e in others[:3]]
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = dict(value=cb.get("value"), unit=cb.get("unit"), cores=cb.get("cores"), kind=cb.get("kind"),
                                    sample=_short(cb.get("sample_short", cb.get("sample", ""))),
                                    single_thread=_val(cb, "single_thread", "value"), all_cores=_val(cb, "all_cores", "value"),
                                    host_cores=_val(cb, "host_cores", "nproc"), cpu_quota=_val(cb, "host_cores", "cgroup_quota_cpus"),
                                    by_threads={k: v.get("value") for k, v in (cb.get("by_threads") or {}).items()} or None,
                                    cores_note=_short(cb.get("cores_note", "")) or None)
    sec = {}
    lat = out.get("frame_latency_ms")
    if isinstance(lat, dict):
        sec["frame_latency_ms"] = dict(median=lat.get("median"), p95=lat.get("p95"))
    oc = out.get("other_configs")
    if isinstance(oc, dict):
        sec["other_configs"] = {k: _num(_val(v, "value")) for k, v in oc.items()}
        c1 = _val(oc, "C1_ref", "cpu_baseline")
        if isinstance(c1, dict):
            sec["other_configs"]["C1_ref_cpu"] = dict(value=c1.get("value"), cores=c1.get("cores"),
                                                      single_thread=_val(c1, "single_thread", "value"))
    for key, path in (("northstar_mode", ("northstar_mode", "value")), ("northstar_fixed_iterations", ("northstar_mode", "fixed_iterations", "value")),
                      ("northstar_cpu", ("northstar_mode", "cpu_baseline", "value")),
                      ("live_depth_mode", ("live_depth_mode", "value")), ("pipelined", ("pipelined", "value")),
                      ("end_to_end_ref_ms", ("end_to_end", "ref", "median_ms")),
                      ("end_to_end_northstar_ms", ("end_to_end", "northstar", "median_ms")),
                      ("raycast_points_ms", ("raycast", "points", "avg_launch_ms")),
                      ("raycast_depth_ms", ("raycast", "depth", "avg_launch_ms")),
                      ("fuse_every_voxel_ms", ("fuse_variants", "every_voxel_stored_ms")),
                      ("fuse_known_occupancy_ms", ("fuse_variants", "occupancy_known_ms"))):
        v = _val(out, *path)
        if v is not None:
            sec[key] = _num(v)
    ms = out.get("multi_sequence")
    if isinstance(ms, dict):
        sec["multi_sequence"] = {k: _num(_val(v, "value")) for k, v in ms.items() if isinstance(v, dict)}
    if sec:
        line["secondary"] = sec
    line["detail_file"] = detail_file
    # the limit is a promise, not a hope: secondary figures go first, then the second-rank rooflines
    for drop in (None, ("secondary", "multi_sequence"), ("secondary",), ("roofline_other",)):
        if drop is not None:
            d = line
            for k in drop[:-1]:
                d = d.get(k, {})
            d.pop(drop[-1], None)
        text = json.dumps(line, separators=(",", ":"))
        if len(text.encode()) < LINE_LIMIT:
            break
    return text


def finish(out, args):
    """writes the full record to the detail file and prints the contract line (rank 0)"""
    path = getattr(args, "detail_file", None)
    shown = None
    if path:
        try:
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
            shown = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
        except OSError as e:
            print("bench.py: could not write %s: %s" % (path, e), file=sys.stderr)
    emit(contract_line(out, shown))


def median_region(replicas, fn, device, repeats):
    """the barrier + synchronize-bracketed region `fn` `repeats` times back to back (the same frames every time);
    returns (median seconds, [seconds of every region]) — MAX over ranks each (replicas.timed_region), so every rank gets the
    same list"""
    regions = [replicas.timed_region(fn, device) for _ in range(max(1, repeats))]
    return sorted(regions)[len(regions) // 2], regions


def under_profiler():
    """rocprofv3 preloads a library that initialises the GPU before this program's first line runs: starting another
    program from here would then be an exec from a process that has touched the GPU — what this pool forbids.  The children
    of this script (the rank launcher, the DynFusion::operator() sequence) are skipped / refused under it."""
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROFILER_", "ROCPROF_")) for k in os.environ)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes of this very script (one per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run would), wait for all of them, forward
    rank 0's JSON line and exit non-zero if any rank failed.  This parent never initialises the GPU (no torch import, no
    HIP call), and no process that has touched the GPU is ever re-executed: the children are new interpreters."""
    import subprocess
    import tempfile
    from dynfu_amd import replicas
    n = args.gpus
    port = replicas._free_port()
    procs, errs = [], []
    errdir = tempfile.mkdtemp(prefix="dfa_bench_ranks_")
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DFA_BENCH_LAUNCHED_BY="bench.py")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
        # ranks > 0: stderr into a file of their own (rank 0 keeps the terminal's), shown if the run fails
        err = open(os.path.join(errdir, "rank%d.err" % rank), "w+") if rank > 0 else None
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=err, text=True))
    # a rank that dies leaves the others in a barrier: poll, and end the rest as soon as one has failed — or when the
    # wall-clock limit is over (a rank that hangs without failing would otherwise hold this parent for ever)
    failed, out0 = None, None
    pending = set(range(n))
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + args.rank_timeout
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        if failed is None and pending and time.monotonic() > deadline:
            failed = (min(pending), 124)
        time.sleep(0.05)
    if failed is not None:
        for r in pending:
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        print("bench.py: rank %d of %d %s; no line is printed" %
              (failed[0], n, "exited with code %d" % failed[1] if failed[1] != 124 or failed[0] not in pending
               else "was still running after %.0f s (--rank-timeout)" % args.rank_timeout), file=sys.stderr)
        for r, err in enumerate(errs):
            if err is None:
                continue
            err.seek(0)
            tail = err.read()[-1500:].strip()
            if tail:
                print("---- stderr of rank %d (%s):\n%s" % (r, err.name, tail), file=sys.stderr)
        raise SystemExit(failed[1] if 0 < failed[1] < 256 else 1)
    reader.join(timeout=30)
    out0 = buf[0] if buf else ""
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if not lines:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        raise SystemExit(1)
    print(lines[-1], flush=True)


def launcher_mark():
    """how this rank process was started: by bench.py's own launcher (launch_ranks), by torch.distributed.run, or directly"""
    if os.environ.get("DFA_BENCH_LAUNCHED_BY") == "bench.py":
        return "bench.py"
    return "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else None


def ranks_seen(device=None):
    """number of live ranks of the process group, counted by an all-reduce (not read from the environment)"""
    from dynfu_amd import replicas
    return replicas.count_ranks(device)


def rccl_selfcheck():
    """what the one-rank process group of an N = 1 run found (dynfu_amd/replicas.py: init(single_rank_group=True))"""
    from dynfu_amd import replicas
    if replicas.selfcheck is not None and "collective_host_ms" not in replicas.selfcheck:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if replicas.selfcheck.get("backend") == "nccl" and torch.cuda.is_available() else None
        replicas.selfcheck["collective_host_ms"] = replicas.time_collectives(dev)
    return replicas.selfcheck if replicas.selfcheck is not None else dict(skipped="a process group of WORLD_SIZE ranks is in use"
                                                                                 if int(os.environ.get("WORLD_SIZE", "1")) > 1
                                                                                 else "--no-rccl-selfcheck")


def main_dry_run(args):
    """The launcher / process-group path with no GPU work: every rank sleeps 20 ms per step inside the same
    barrier-bracketed timed region and rank 0 prints a line with the contract's fields (tests only)."""
    from dynfu_amd import replicas
    rank, local, world = replicas.env_world()
    replicas.pin_to_core_slice()
    replicas.init(backend=args.backend, single_rank_group=not args.no_rccl_selfcheck)
    seen = ranks_seen()
    if rank == args.dry_run_fail_rank:
        os._exit(7)
    if rank == args.dry_run_hang_rank:
        time.sleep(1e6)
    K, Wm = args.steps, args.warmup
    dt_max, regions = median_region(replicas, lambda: time.sleep(0.02 * K), None, args.repeats)
    if rank == 0:
        finish(dict(metric="frames/sec (warp-solve + TSDF fuse), 512^3 vol / 2k nodes / VGA depth",
                              value=round(seen * K / dt_max, 2), unit="frames/s", n_gpus=seen, steps=K, warmup=Wm,
                              ms_per_step=round(dt_max / K * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
                              dtype="f32", data="none (dry run: sleeps)",
                              config=dict(workload="dry run of the rank launcher (no GPU work)", ranks_seen=seen,
                                          rccl_selfcheck=rccl_selfcheck(), repeats=len(regions), launcher=launcher_mark(),
                                          parallelism="replicas x%d (no collective; counted by all-reduce, %s)" % (seen, args.backend)),
                              region_ms=[round(r * 1e3, 4) for r in regions]), args)
    replicas.shutdown()


def main():
    args = parse()
    if (args.gpus > 1 or args.force_launcher) and "WORLD_SIZE" not in os.environ:
        if under_profiler():
            raise SystemExit("bench.py --gpus N starts rank processes: not under rocprofv3 (profile one rank: --gpus 1)")
        return launch_ranks(args, sys.argv[1:])
    claim_stdout()  # (a rank process: from here on stdout carries the JSON line and nothing else)
    if args.dry_run:
        return main_dry_run(args)
    import torch

    from dynfu_amd import replicas
    rank, local, world = replicas.env_world()
    launched = os.environ.get("DFA_BENCH_LAUNCHED_BY") == "bench.py"
    if world > 1 or launched:
        replicas.pin_to_core_slice()  # a contiguous slice of the host's cores per rank (in-process)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (without torchrun, `python bench.py --gpus N` starts the N ranks "
                         "itself)" % (args.gpus, world))
    # The DynFusion::operator() sequence runs in a CHILD process with its own HIP context.  It is started here, before this
    # process has made its first GPU call (torch.cuda.is_available() below is one), and has finished before the timed
    # region starts: nothing of it overlaps the measurement.
    e2e = None
    if world == 1 and args.mode == "ref" and args.live == "targets" and not args.no_end_to_end and under_profiler():
        e2e = dict(skipped="running under rocprofv3: no child process is started from a profiled process")
    elif world == 1 and args.mode == "ref" and args.live == "targets" and not args.no_end_to_end:
        if not os.path.exists("/dev/kfd"):  # (a file probe: nothing here may touch the GPU before the child has run)
            raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
        try:
            e2e = end_to_end(args.config)
        except Exception as e:  # noqa: BLE001
            e2e = dict(error="%s: %s" % (type(e).__name__, e))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # nccl = RCCL; only barriers, one MAX and one SUM all-reduce use it.  N = 1: a one-rank group runs the same calls
    # (config.rccl_selfcheck), guarded so that a failure there costs the line nothing
    replicas.init(backend=args.backend, device=device, single_rank_group=not args.no_rccl_selfcheck)
    n_gpus = ranks_seen(device)  # counted, not assumed
    if n_gpus != world:
        raise SystemExit("bench.py: %d ranks answered the all-reduce, WORLD_SIZE=%d" % (n_gpus, world))

    if args.mode == "northstar":
        return main_northstar(args, torch, replicas, rank, world, device)
    if args.live == "depth":
        seq = SequenceLive(args.config, device)
        dt_max = replicas.timed_region(lambda: [seq.frame(f) for f in range(args.warmup)], device)  # warm-up, untimed below
        K = args.steps
        dt_max, regions = median_region(replicas, lambda: [seq.frame(args.warmup + f) for f in range(K)], device, args.repeats)
        if rank == 0:
            rec = live_depth_probe(args.config, device, steps=min(K, 30), warmup=0, seq=seq)
            rec.update(metric="frames/sec (warp-solve + TSDF fuse), 512^3 vol / 2k nodes / VGA depth", value=round(n_gpus * K / dt_max, 2),
                       n_gpus=n_gpus, steps=K, warmup=args.warmup, ms_per_step=round(dt_max / K * 1e3, 4), higher_is_better=True,
                       scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                       config=dict(workload=rec.pop("workload"), ranks_seen=n_gpus, repeats=len(regions), rccl_selfcheck=rccl_selfcheck(),
                                   parallelism="replicas x%d (one sequence per GPU, no collective)" % n_gpus),
                       region_ms=[round(r * 1e3, 4) for r in regions])
            finish(rec, args)
        replicas.shutdown()
        return
    seq = Sequence(args.config, device)
    seq.fuse_first = args.fuse_first
    shadow = seq.D <= 2048 and not (args.fuse_first or args.fuse_after_build or args.serial or args.pipeline)
    if shadow:
        seq.enable_pcg_shadow()
    if args.pipeline and not args.serial:
        seq.enable_pipeline()
    cfg = seq.cfg
    K, Wm = args.steps, args.warmup

    for f in range(Wm):
        seq.frame(f, args.serial)
    # Per-kernel durations come from hipEvent brackets INSIDE the timed region, on the streams the kernels run on.
    # An event costs ~5 us of stream time (15 of them per frame would be 6 % of a C2 frame), so every
    # TIMING_SAMPLE-th frame is bracketed and the others run as a caller without instrumentation would.
    fuse_events = []
    plans = getattr(seq, "plans", [seq.solver])
    for plan in plans:
        plan.enable_timing(1)  # new measurement ...
        plan.enable_timing(0)  # ... paused

    def timed():
        for f in range(K):
            sampled = f % TIMING_SAMPLE == 0
            for plan in plans:
                plan.enable_timing(2 if sampled else 0)
            seq.frame(Wm + f, args.serial, fuse_events if sampled else None)

    # barrier + synchronize on both sides, MAX over ranks
    dt_max, regions = median_region(replicas, timed, device, args.repeats)
    for plan in plans:
        plan.enable_timing(0)
    tm = max((plan.timing() for plan in plans), key=lambda t: t["solves"])  # sums over the sampled frames
    st = seq.solver.stats()  # last frame

    # ---- parity spot check of the last frame (outside the timed region)
    t_err = float((seq.solver.translations() - seq.t_true[(Wm + K - 1) % seq.n_frames]).abs().max())

    if rank != 0:
        replicas.shutdown()
        return

    # ---- per-frame latency (SURVEY.md §8d asks for median and p95): a separate pass OUTSIDE the timed region, every
    # frame bracketed by two events and synchronised, so that a frame cannot hide behind the launches of the next
    lat = []
    for f in range(min(K, 50)):
        e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
        e0.record()
        seq.frame(Wm + f, args.serial)
        e1.record()
        e1.synchronize()
        lat.append(e0.elapsed_time(e1))
    lat.sort()

    # ---- roofline of the kernels measured live with HIP events on their launch streams
    dim, Wd, Hd = cfg["dim"], cfg["width"], cfg["height"]
    V = dim ** 3
    fuse_ms = float(np.mean([a.elapsed_time(b) for a, b in fuse_events])) if fuse_events else float("nan")
    fuse_bytes = 4.0 * V + 2.0 * Wd * Hd  # fused clear+integrate writes every voxel once
    fuse_gbs = fuse_bytes / (fuse_ms * 1e-3) / 1e9
    # per-kernel numbers: hipEvent brackets on the solve stream, summed over ALL K timed frames (plans of the
    # pipelined variant: the measured one)
    nnz = tm["matrix_nnz"]
    frames_timed = max(1, tm["solves"])
    its_total = tm["pcg_iters"]
    its = its_total / frames_timed  # PCG iterations per frame (mean)
    # PCG: per iteration the matrix (4 B value + 4 B column per non-zero) + 6 vectors of 3D floats
    pcg_bytes = its * (8.0 * nnz + 24.0 * 3 * seq.D)  # per frame
    pcg_total_ms = tm["pcg_ms"] / frames_timed        # per frame
    pcg_gbs = pcg_bytes / (pcg_total_ms * 1e-3) / 1e9 if pcg_total_ms > 0 else float("nan")
    launches_pf = tm["pcg_launches"] / frames_timed
    fuse_entry = dict(kernel="integrate_runs_kernel<FUSED_CLEAR,32,8> (clear+integrate %d^3, runs of 8 voxels classified "
                             "against min/max tiles of the depth image)" % dim, bound="hbm",
                      achieved=round(fuse_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(fuse_gbs / HBM_PEAK_GBS, 4),
                      traffic=PMC_TRAFFIC_BYTES.get((args.config, "fused_integrate")),
                      traffic_source=traffic_source(args.config, "fused_integrate"),
                      avg_launch_ms=round(fuse_ms, 4), launches_per_frame=1, algorithmic_bytes_per_launch=fuse_bytes)
    split = seq.D <= 2048  # register-resident kernel, one workgroup per coordinate
    pcg_common = dict(traffic=PMC_TRAFFIC_BYTES.get((args.config, "pcg")), traffic_source=traffic_source(args.config, "pcg"),
                      algorithmic_bytes_per_launch=round(pcg_bytes / max(1e-9, launches_pf), 1),
                      avg_launch_ms=round(pcg_total_ms / max(1e-9, launches_pf), 4),
                      launches_per_frame=round(launches_pf, 2), pcg_iterations_per_frame=round(its, 1), matrix_nnz=nnz,
                      algorithmic_bytes_per_frame=round(pcg_bytes, 1), frames_measured=frames_timed,
                      # SURVEY 8(d) read literally prices a 3 x 3 block per non-zero (nnz_blocks x 9 x 4 + 6 n x 4, n = 3 D); the
                      # matrix of the translation-only energy is A (x) I_3, so one scalar per block is stored and the smaller
                      # figure above is the one `achieved` uses
                      survey_8d_literal_bytes_per_frame=round(its * (36.0 * nnz + 72.0 * seq.D), 1))
    if split:
        # What bounds it is the gather of p from LDS — 4 B per non-zero and coordinate, on the three CUs the three
        # workgroups occupy — not HBM: the matrix is read once per launch and stays in registers (traffic above: 1/16 of
        # the algorithmic bytes).  The HBM figures stay in the entry because the contract's metric is HBM-based.
        lds_bytes = 12.0 * nnz * its
        lds_gbs = lds_bytes / (pcg_total_ms * 1e-3) / 1e9 if pcg_total_ms > 0 else float("nan")
        # `frac` is the contract's HBM view (algorithmic bytes / launch time / 8 TB/s); the LDS-gather roof that binds the
        # kernel is the secondary view
        pcg_entry = dict(kernel="pcg_paired_kernel<..,NC=1> (Jacobi PCG, matrix in registers, 3 workgroups = 3 coordinates)",
                         bound="hbm", achieved=round(pcg_gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=round(pcg_gbs / HBM_PEAK_GBS, 6),
                         lds_gather_view=dict(bound="lds-gather", achieved=round(lds_gbs, 1), peak=LDS_GATHER_PEAK_GBS, unit="GB/s",
                                              frac=round(lds_gbs / LDS_GATHER_PEAK_GBS, 4), lds_gather_bytes_per_frame=lds_bytes,
                                              peak_is="3 CUs x 128 B/clk (ds_read_b32, conflict-free) x 2.4 GHz; random 4-byte "
                                                      "gathers of 32 lanes into 32 banks measure ~6.5-7.4 clk per wave instruction "
                                                      "against 2 (tools/microbench_lds_gather.hip)"),
                         **pcg_common)
        pcg_entry["note"] = ("register/LDS-resident and synchronisation-bound by design: one workgroup per coordinate, %d "
                             "barrier-separated iterations per frame, 253 of 256 CUs idle while it runs (the fuse fills them); "
                             "frac is the contract's HBM view, lds_gather_view the roof that binds the kernel" % round(its))
    else:
        team = seq.solver.team_pcg_info()
        pcg_entry = dict(kernel=("pcg_team_kernel (Jacobi PCG by three teams of persistent workgroups, a coordinate per XCD, the matrix "
                                 "in registers; + its guard launch)" if team["launches"] and not team["disabled"] else
                                 "pcg_mb_step_kernel (many-workgroup Jacobi PCG, a launch per iteration)"),
                         bound="hbm", achieved=round(pcg_gbs, 2), peak=HBM_PEAK_GBS,
                         unit="GB/s", frac=round(pcg_gbs / HBM_PEAK_GBS, 6), team_pcg=team, **pcg_common)
        pcg_entry["note"] = ("synchronisation-bound by design: an iteration is one barrier among 32 workgroups through one XCD's L2 "
                             "(~2.9 us at C3); the matrix is read once per launch and stays in registers")
    dominant, other = (pcg_entry, fuse_entry) if pcg_total_ms > fuse_ms else (fuse_entry, pcg_entry)

    out = dict(metric="frames/sec (warp-solve + TSDF fuse), 512^3 vol / 2k nodes / VGA depth",
               value=round(n_gpus * K / dt_max, 2), unit="frames/s", n_gpus=n_gpus, steps=K, warmup=Wm,
               ms_per_step=round(dt_max / K * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f32", data="synthetic",
               config=dict(workload="%s: %d^3 TSDF, %dx%d depth, %d nodes, k=%d, %d vertices, %d GN x PCG<=256 (tol 1e-6), "
                                    "reference-parity energy (energy.t), lambda=200"
                                    % (args.config, dim, Wd, Hd, seq.D, seq.k, seq.N, cfg["gn_iters"]),
                           workload_detail="4 B voxels; Gauss-Newton iterations behind a gradient at the round-off floor are no-ops and "
                                           "return at entry; PCG tolerance 1e-6 per linearisation, never below 1e-12 of the solve's "
                                           "first gradient",
                           parallelism="replicas x%d (one sequence per GPU, no collective)" % n_gpus, ranks_seen=n_gpus,
                           rccl_selfcheck=rccl_selfcheck(), streams_probe=STREAM_PROBES[:1], repeats=len(regions),
                           launcher=launcher_mark(),
                           streams="serial" if args.serial else ("fuse || graph build of frame f+1 || solve of frame f on three "
                                                                 "HIP streams, two solver plans" if args.pipeline else
                                                                 ("fuse || graph build + solve on two HIP streams" if args.fuse_first
                                                                  else ("graph build, then solve on one HIP stream with the fuse on a "
                                                                        "second one in the shadow of the first PCG" if shadow else
                                                                        "graph build, then fuse || solve on two HIP streams"))),
                           pcg_iterations_last_frame=st["pcg_iters"], gn_iterations_last_frame=st["gn_iters"],
                           gn_iterations_noop_last_frame=st["gn_noop"],
                           max_abs_translation_error_vs_ground_truth_m=round(t_err, 6)),
               frame_latency_ms=dict(median=round(lat[len(lat) // 2], 4), p95=round(lat[min(len(lat) - 1, int(0.95 * len(lat)))], 4),
                                     frames=len(lat), note="each frame synchronised, measured after the timed region"),
               region_ms=[round(r * 1e3, 4) for r in regions],
               roofline=dominant, roofline_other=[other],
               solve_kernels_ms_per_frame=dict(pcg=round(pcg_total_ms, 4), assemble=round(tm["assemble_ms"] / frames_timed, 4)))
    if world == 1:
        # secondary figures: a failure in one of them must not cost the line its contract fields
        if not args.no_raycast:
            try:
                out["raycast"] = raycast_probe(seq, args.config)
            except Exception as e:  # noqa: BLE001
                out["raycast"] = dict(error="%s: %s" % (type(e).__name__, e))
        if not args.no_raycast:
            try:
                out["fuse_variants"] = fuse_variants_probe(seq)
            except Exception as e:  # noqa: BLE001
                out["fuse_variants"] = dict(error="%s: %s" % (type(e).__name__, e))
        if not (args.pipeline or args.serial or args.no_pipelined_probe):
            try:
                out["pipelined"] = pipelined_probe(seq, Wm + K, device)
            except Exception as e:  # noqa: BLE001
                out["pipelined"] = dict(error="%s: %s" % (type(e).__name__, e))
        # the frame the CPU baseline below ends on, solved here too: its translations are compared with the oracle's
        t_cmp, f_cmp = None, max(1, args.cpu_frames) - 1
        if not args.no_cpu_baseline and not (args.pipeline or args.serial):
            try:
                seq.frame(f_cmp)
                t_cmp = seq.solver.translations().cpu().numpy().copy()
            except Exception:  # noqa: BLE001
                t_cmp = None
        del seq
        torch.cuda.empty_cache()
        if not args.no_multi_sequence and args.sequences_per_gpu:
            try:
                out["multi_sequence"] = multi_sequence_probe(args.config, device, counts=args.sequences_per_gpu,
                                                             threads=args.sequences_threads)
            except Exception as e:  # noqa: BLE001
                out["multi_sequence"] = dict(error="%s: %s" % (type(e).__name__, e))
        if not args.no_northstar:
            try:
                out["northstar_mode"] = northstar_probe(args.config, device,
                                                        cpu_frames=0 if args.no_cpu_baseline else max(1, args.cpu_frames // 2))
            except Exception as e:  # noqa: BLE001
                out["northstar_mode"] = dict(error="%s: %s" % (type(e).__name__, e))
        if not args.no_live_depth:
            try:
                out["live_depth_mode"] = live_depth_probe(args.config, device)
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                out["live_depth_mode"] = dict(error="%s: %s" % (type(e).__name__, e))
        if e2e is not None:
            out["end_to_end"] = e2e
        if not args.no_other_configs and args.config == "C2":
            out["other_configs"] = other_configs(device, cpu=not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            out["cpu_baseline"], t_cpu, f_cpu = cpu_baseline(args.config, max(1, args.cpu_frames))
            # the oracle's own answer for its last frame against this line's solve of that frame (t_cmp: above)
            if t_cmp is not None and f_cpu == f_cmp:
                out["config"]["max_abs_translation_diff_vs_oracle_m"] = float("%.3g" % np.abs(t_cmp - t_cpu).max())
                out["config"]["oracle_check"] = ("frame %d solved by the HIP path and by oracle/solve_oracle (fp32 CPU restatement, the "
                                                 "cpu_baseline leg): largest difference of a node translation component" % f_cmp)
    finish(out, args)
    replicas.shutdown()


if __name__ == "__main__":
    main()
