"""-m gpu parity tests of the north-star (6-DoF) solve: HIP (through the C ABI) vs the
double-precision oracle oracle/solve6_oracle.c.

The mode is not in the reference (parity unpinned); the oracle is pinned by
tests/test_oracle_solve6.py (finite differences, dense solve).  Tolerances (fp32 kernels vs fp64
oracle, same iteration counts): energies within 1e-3 relative, association counts equal up to
pixel-rounding ties (<= 0.1 % of the rows), warped vertices within 5e-5 m on average and 1e-3 m
at worst, computePointNormals bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from dynfu_amd import synth  # noqa: E402
from gpu_util import bits, dev, host  # noqa: E402


@pytest.fixture(scope="module")
def A():
    import dynfu_amd
    dynfu_amd.load()
    return dynfu_amd


@pytest.mark.parametrize("shape", [(120, 160), (37, 53), (2, 2), (1, 5)])
def test_points_normals_bit_exact(A, shape):
    rng = np.random.default_rng(3)
    H, W = shape
    depth = (1500 + 300 * np.sin(np.arange(W) / 9.0)[None, :] + rng.integers(0, 40, shape)).astype(np.uint16)
    depth[rng.random(shape) < 0.05] = 0
    fx, fy, cx, cy = 131.25, 128.0, W / 2 - 0.5, H / 2 - 0.5
    P, Nm = A.compute_points_normals(dev(depth), fx, fy, cx, cy)
    Pr, Nr = O.points_normals(depth, fx, fy, cx, cy)
    assert np.array_equal(bits(host(P)), bits(Pr)) and np.array_equal(bits(host(Nm)), bits(Nr))


def _scene(name, frame):
    cfg = synth.CONFIGS[name]
    c = synth.canonical(cfg)
    intr = synth.intrinsics(cfg)
    depth = synth.depth_frame(cfg, frame)
    return cfg, c, intr, depth


def _solve_both(A, cfg, c, intr, depth, node_dq, threads=8, **kw):
    k = cfg["k"]
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    s = A.Solver6(cfg["D"], len(c["verts"]), k)
    keep = [dev(c["node_pos"]), dev(node_dq), dev(c["node_w"]), dev(c["verts"]), dev(c["normals"])]
    s.set_problem(*keep)
    s.solve(P, Nm, *intr, A.Solve6Params(**kw))
    st = s.stats()
    dq = host(s.node_dq())
    wv, wn_ = s.warp()
    dq_ref, st_ref = O.solve6(c["node_pos"], node_dq, c["node_w"], k, c["verts"], c["normals"], host(P), host(Nm), intr,
                              threads=threads, **kw)
    return s, dq, st, host(wv), host(wn_), dq_ref, st_ref


@pytest.mark.parametrize("name,frame,k_override", [("T0", 4, None), ("T1", 6, None), ("T1", 9, 4), ("T0", 5, 3), ("T1", 4, 6), ("T0", 2, 1)])
def test_solve_matches_the_oracle(A, name, frame, k_override):
    cfg, c, intr, depth = _scene(name, frame)
    if k_override:
        cfg = dict(cfg, k=k_override)
    # short run: fp32 and fp64 trajectories have not drifted apart yet (the problem has nearly free sliding
    # modes, so rounding differences grow over many truncated-PCG Gauss-Newton steps)
    kw = dict(num_iter=1, gn_iter=2, linear_iter=40, lambda_=200.0)
    s, dq, st, wv, wn_, dq_ref, st_ref = _solve_both(A, cfg, c, intr, depth, c["node_dq"], **kw)
    assert st["overflow"] == 0 and st["gn_iters"] == st_ref["gn_iters"] == 2
    assert st["initial_cost"] == pytest.approx(st_ref["initial_cost"], rel=1e-4)
    # k = 1 (one node per vertex): the 40 iterations leave the PCG at a relative residual of 1e-2, where the energy after the
    # step moves by 3e-3 with the summation order of the assembly (and letting it converge lets the free sliding modes drift
    # by millimetres between fp32 and fp64): a wider band there
    assert st["final_cost"] == pytest.approx(st_ref["final_cost"], rel=5e-3 if cfg["k"] == 1 else 1e-3)
    assert abs(st["valid_first"] - st_ref["valid_first"]) <= 1e-3 * st_ref["valid_first"] + 2
    assert abs(st["pcg_iters"] - st_ref["pcg_iters"]) <= 0.05 * st_ref["pcg_iters"] + 2
    idx, wn, _ = O.graph6(c["node_pos"], c["node_w"], cfg["k"], c["verts"])
    p_ref, n_ref = O.warp6(dq_ref, idx, wn, c["verts"], c["normals"])
    d = np.linalg.norm(wv - p_ref, axis=1)
    # (k = 1: diagonal blocks with condition numbers of 1e5 in fp32 — the same wider band)
    assert d.mean() < (2e-4 if cfg["k"] == 1 else 5e-5) and d.max() < (2e-3 if cfg["k"] == 1 else 1e-3)
    assert np.abs(wn_ - n_ref).max() < (5e-3 if cfg["k"] == 1 else 2e-3)
    # the device warp is the oracle's DQ blend of the device's own transforms
    p_same, _ = O.warp6(dq, idx, wn, c["verts"])
    assert np.abs(wv - p_same).max() < 2e-6
    assert st["final_cost"] < 0.5 * st["initial_cost"]


def _threads():
    import os
    return max(1, min(16, os.cpu_count() or 1))


@pytest.mark.parametrize("name,frame", [("C2", 7), ("C3", 11)])
def test_solve_matches_the_oracle_at_baseline_sizes(A, name, frame):
    """BASELINE configs C2 (2 048 nodes, k = 4, 262 144 vertices) and C3 — whose "+ ARAP" IS this mode (4 096
    nodes, k = 8, 524 288 vertices): one frame, 2 Gauss-Newton iterations x 40 PCG iterations, HIP (fp32) against the fp64
    statement oracle/solve6_oracle.c with the tolerances of test_solve_matches_the_oracle.  The data term follows the
    gates of src/kfusion/cuda/proj_icp.cu:72-98 and the row shape of :343-350."""
    cfg, c, intr, depth = _scene(name, frame)
    k = cfg["k"]
    kw = dict(num_iter=1, gn_iter=2, linear_iter=40, lambda_=200.0)
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    s = A.Solver6(cfg["D"], len(c["verts"]), k)
    keep = [dev(c["node_pos"]), dev(c["node_dq"]), dev(c["node_w"]), dev(c["verts"]), dev(c["normals"])]
    s.set_problem(*keep)
    s.solve(P, Nm, *intr, A.Solve6Params(**kw))
    st = s.stats()
    wv, wn_ = s.warp()
    dq_ref, st_ref = O.solve6(c["node_pos"], c["node_dq"], c["node_w"], k, c["verts"], c["normals"], host(P), host(Nm), intr,
                              threads=_threads(), **kw)
    assert st["overflow"] == 0 and st["gn_iters"] == st_ref["gn_iters"] == 2
    assert st["initial_cost"] == pytest.approx(st_ref["initial_cost"], rel=1e-4)
    # (the energy after two truncated steps: 1.1e-3 apart at C2 with the fourth form's summation order, 0.8e-3 with the third's —
    # the moments themselves agree with an fp64 sum of the same records to 1.5e-6, tools/ns_plan_check.py)
    assert st["final_cost"] == pytest.approx(st_ref["final_cost"], rel=2e-3)
    assert abs(st["valid_first"] - st_ref["valid_first"]) <= 1e-3 * st_ref["valid_first"] + 2
    assert st["pcg_it_hist"] == st_ref["pcg_it_hist"] == [40, 40]
    # the residual the truncated PCGs stopped at, as both sides report it
    assert np.allclose(st["pcg_rel_hist"], st_ref["pcg_rel_hist"], rtol=0.05, atol=1e-5)
    assert np.allclose(st["cost_hist"], st_ref["cost_hist"], rtol=2e-3)
    idx, wn, _ = O.graph6(c["node_pos"], c["node_w"], k, c["verts"], threads=_threads())
    p_ref, n_ref = O.warp6(dq_ref, idx, wn, c["verts"], c["normals"])
    d = np.linalg.norm(host(wv) - p_ref, axis=1)
    assert d.mean() < 5e-5 and d.max() < 1e-3
    # normals: rotations only; the worst node of 2 048 / 4 096 (a nearly free tangential rotation) decides the maximum
    dn = np.abs(host(wn_) - n_ref)
    assert dn.mean() < 1e-4 and dn.max() < 1e-2
    assert st["final_cost"] < 0.05 * st["initial_cost"]
    s.close()


@pytest.mark.parametrize("name,frame", [("T1", 6), ("C2", 7)])
def test_forcing_schedule_matches_the_oracle(A, name, frame):
    """Inexact Newton: Gauss-Newton iteration i of an outer iteration stops its PCG at max(pcg_tol, first * decay^i).  The
    PCGs stop by tolerance, below the iteration cap; iteration counts agree with the fp64 statement up to the crossing of a
    threshold (+- 2 and 15 %), energies within 2 % (a PCG stopped one iteration earlier or later at a LOOSE tolerance is a
    visibly different step), the reported residuals are below the tolerance of their iteration."""
    cfg, c, intr, depth = _scene(name, frame)
    kw = dict(num_iter=2, gn_iter=3, linear_iter=64, lambda_=200.0, pcg_tol=1e-3, pcg_tol_first=0.1, pcg_tol_decay=0.5)
    s, dq, st, wv, wn_, dq_ref, st_ref = _solve_both(A, cfg, c, intr, depth, c["node_dq"], **kw)
    assert st["gn_iters"] == st_ref["gn_iters"] == 6
    tol = [0.1, 0.05, 0.025] * 2
    for i in range(6):
        assert st["pcg_it_hist"][i] < 64 and st_ref["pcg_it_hist"][i] < 64
        assert abs(st["pcg_it_hist"][i] - st_ref["pcg_it_hist"][i]) <= 2 + 0.15 * st_ref["pcg_it_hist"][i]
        assert 0 < st["pcg_rel_hist"][i] <= tol[i] * (1 + 1e-5)
    assert sum(st["pcg_it_hist"]) == st["pcg_iters"]
    assert st["cost_hist"][0] == pytest.approx(st_ref["cost_hist"][0], rel=1e-4)
    assert st["cost_hist"][-1] == st["final_cost"]
    assert st["final_cost"] == pytest.approx(st_ref["final_cost"], rel=0.02)
    assert st["final_cost"] < 0.05 * st["initial_cost"]
    idx, wn, _ = O.graph6(c["node_pos"], c["node_w"], cfg["k"], c["verts"], threads=_threads())
    p_ref, _ = O.warp6(dq_ref, idx, wn, c["verts"])
    # (loosely solved steps that differ by one PCG iteration: half a millimetre on average, the depth frame's own resolution)
    assert np.linalg.norm(wv - p_ref, axis=1).mean() < 5e-4
    # a PCG that runs into its cap reports the residual it was stopped at (here: above the tolerance asked for)
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    s.solve(P, Nm, *intr, A.Solve6Params(num_iter=1, gn_iter=1, linear_iter=5, lambda_=200.0, pcg_tol=1e-6))
    st5 = s.stats()
    _, st5_ref = O.solve6(c["node_pos"], c["node_dq"], c["node_w"], cfg["k"], c["verts"], c["normals"], host(P), host(Nm), intr,
                          threads=_threads(), num_iter=1, gn_iter=1, linear_iter=5, lambda_=200.0, pcg_tol=1e-6)
    assert st5["pcg_it_hist"] == [5] and st5["pcg_rel_hist"][0] == pytest.approx(st5_ref["pcg_rel_hist"][0], rel=0.02)
    assert 1e-6 < st5["pcg_rel_hist"][0] < 1.0


@pytest.mark.parametrize("name,frame", [("T1", 6), ("C2", 7)])
def test_adaptive_forcing_matches_the_oracle(A, name, frame):
    """dfa_solve6_params.pcg_tol_adapt: the Eisenstat-Walker forcing term.  The tolerance every PCG was asked for comes out
    of the device's own gradients — it agrees with the fp64 statement's (15 %: a ratio of two inner products of truncated
    solves), is tight behind a gradient that fell fast and sits at its upper bound once the fit stagnates; iteration counts
    and energies as in test_forcing_schedule_matches_the_oracle."""
    cfg, c, intr, depth = _scene(name, frame)
    kw = dict(num_iter=2, gn_iter=3, linear_iter=64, lambda_=200.0, pcg_tol=1e-3, pcg_tol_first=0.1, pcg_tol_adapt=0.9)
    s, dq, st, wv, wn_, dq_ref, st_ref = _solve_both(A, cfg, c, intr, depth, c["node_dq"], threads=_threads(), **kw)
    assert st["gn_iters"] == st_ref["gn_iters"] == 6
    assert st["pcg_tol_hist"][0] == pytest.approx(0.1) and st["pcg_tol_hist"][3] == pytest.approx(0.1)  # first of an outer iteration
    assert st["pcg_tol_hist"][1] < 0.03  # the first step took the energy down by orders of magnitude: a tight solve follows
    for i in range(6):
        assert 1e-3 <= st["pcg_tol_hist"][i] <= 0.1 * (1 + 1e-6)
        assert st["pcg_tol_hist"][i] == pytest.approx(st_ref["pcg_tol_hist"][i], rel=0.15)
        assert st["pcg_it_hist"][i] < 64 and abs(st["pcg_it_hist"][i] - st_ref["pcg_it_hist"][i]) <= 2 + 0.15 * st_ref["pcg_it_hist"][i]
        assert 0 < st["pcg_rel_hist"][i] <= st["pcg_tol_hist"][i] * (1 + 1e-5)
    assert st["final_cost"] == pytest.approx(st_ref["final_cost"], rel=0.02)
    assert st["final_cost"] < 0.05 * st["initial_cost"]
    # the same energy as the geometric schedule reaches (which is cheaper over 3 iterations per outer iteration and dearer
    # over 5: it keeps tightening after the fit has reached the noise — bench.py --forcing geometric)
    s.solve(*A.compute_points_normals(dev(depth), *intr), *intr,
            A.Solve6Params(**dict(kw, pcg_tol_adapt=0.0, pcg_tol_decay=0.5)))
    geo = s.stats()
    assert st["final_cost"] < 1.1 * geo["final_cost"]
    # same inputs, same bits
    s.solve(*A.compute_points_normals(dev(depth), *intr), *intr, A.Solve6Params(**kw))
    assert np.array_equal(host(s.node_dq()), dq)
    s.close()


def test_more_than_8192_nodes_matches_the_oracle(A):
    """9 216 nodes (k = 4, 1 179 648 vertices): the PCG's scalars are sums of one partial per workgroup of 8 nodes, of
    which a lane keeps 16 in registers — beyond 8 192 nodes the rest is summed in a loop (a round-2 build dropped them:
    the adaptor's 512^3 sequence, 8 469 nodes, ran on slightly wrong step lengths)."""
    cfg = dict(synth.CONFIGS["C2"], D=9216)
    c = synth.canonical(cfg)
    intr = synth.intrinsics(cfg)
    depth = synth.depth_frame(cfg, 5)
    kw = dict(num_iter=1, gn_iter=2, linear_iter=25, lambda_=200.0)
    s, dq, st, wv, wn_, dq_ref, st_ref = _solve_both(A, cfg, c, intr, depth, c["node_dq"], threads=_threads(), **kw)
    assert st["overflow"] == 0 and st["pcg_it_hist"] == st_ref["pcg_it_hist"] == [25, 25]
    assert np.allclose(st["pcg_rel_hist"], st_ref["pcg_rel_hist"], rtol=0.05)
    assert st["final_cost"] == pytest.approx(st_ref["final_cost"], rel=1e-3)
    sample = np.arange(0, len(c["verts"]), 37)
    idx, wn, _ = O.graph6(c["node_pos"], c["node_w"], cfg["k"], c["verts"][sample], threads=_threads())
    p_ref, _ = O.warp6(dq_ref, idx, wn, c["verts"][sample])
    d = np.linalg.norm(wv[sample] - p_ref, axis=1)
    assert d.mean() < 5e-5 and d.max() < 2e-3  # (mean 3e-5; the worst vertex of 7 k sampled, 1.02 mm, sits on a nearly free node)
    s.close()


def test_adaptive_launch_budget(A):
    """dfa_solve6_params.adaptive_launch: the budget of solve n is a function of the plan's solves up to n - 2 only (folded
    in order behind their completion events), so (a) the first two solves enqueue the full budget, (b) later ones what the
    Gauss-Newton iteration needed before (a running maximum + a quarter, at least two) — with the same bits as the full budget while the prediction holds, (c) the budgets, and with
    them the results, do not depend on whether the host waits for the device between solves, (d) other stopping rules or
    another problem size start from the full budget again, (e) a PCG that needs more than its budget is cut, says so, and
    the budget recovers."""
    cfg, c, intr, _ = _scene("T1", 6)
    frames = [6, 6, 6, 6, 6, 7, 7, 7]
    maps = {f: A.compute_points_normals(dev(synth.depth_frame(cfg, f)), *intr) for f in set(frames)}
    keep = [dev(c["node_pos"]), dev(c["node_dq"]), dev(c["node_w"]), dev(c["verts"]), dev(c["normals"])]
    kw = dict(num_iter=2, gn_iter=3, linear_iter=64, lambda_=200.0, pcg_tol=1e-3, pcg_tol_first=0.1, pcg_tol_decay=0.5)

    def run(adaptive, wait):
        s = A.Solver6(cfg["D"], len(c["verts"]), cfg["k"])
        s.set_problem(*keep)
        out = []
        for f in frames:
            s.solve(*maps[f], *intr, A.Solve6Params(adaptive_launch=adaptive, **kw))
            if wait:
                st = s.stats()  # synchronises
                out.append((host(s.node_dq()).copy(), st))
        st = s.stats()
        dq = host(s.node_dq()).copy()
        s.close()
        return out, dq, st

    full, dq_full, _ = run(0, True)
    assert all(st["pcg_launches"] == 6 * 65 and st["pcg_short"] == 0 for _, st in full)
    seq, dq_seq, st_seq = run(1, True)
    assert seq[0][1]["pcg_launches"] == seq[1][1]["pcg_launches"] == 6 * 65          # (a)
    assert all(st["pcg_launches"] < 6 * 65 for _, st in seq[2:])                    # (b)
    for i in range(2, len(frames)):
        # the budget of solve i from the counts of solves 0 .. i - 2
        want = 0
        for gi in range(6):
            pred = 0
            for j in range(0, i - 1):
                pred = max(full[j][1]["pcg_it_hist"][gi], pred - 1)
            want += min(64, pred + max(2, pred // 4)) + 1
        if all(st["pcg_short"] == 0 for _, st in seq[:i]):
            assert seq[i][1]["pcg_launches"] == want, (i, seq[i][1]["pcg_launches"], want)
    for (dq_a, st_a), (dq_f, st_f) in zip(seq, full):
        if st_a["pcg_short"] == 0:  # the prediction held: the launches left out were no-ops
            assert np.array_equal(dq_a, dq_f) and st_a["pcg_it_hist"] == st_f["pcg_it_hist"]
    assert sum(st["pcg_short"] for _, st in seq) <= 2  # consecutive frames: counts move by one or two
    _, dq_async, st_async = run(1, False)                                            # (c)
    assert np.array_equal(dq_async, dq_seq) and st_async["pcg_launches"] == st_seq["pcg_launches"]

    # (d) other stopping rules on the same plan: no stale prediction, the full budget again
    s = A.Solver6(cfg["D"], len(c["verts"]), cfg["k"])
    s.set_problem(*keep)
    for _ in range(4):
        s.solve(*maps[6], *intr, A.Solve6Params(adaptive_launch=1, **kw))
    assert s.stats()["pcg_launches"] < 6 * 65
    hard = dict(kw, pcg_tol=1e-5, pcg_tol_first=0.0)
    s.solve(*maps[6], *intr, A.Solve6Params(adaptive_launch=1, **hard))
    st = s.stats()
    assert st["pcg_launches"] == 6 * 65 and st["pcg_short"] == 0
    # (e) the same stopping rules on live data that needs more iterations than the plan has seen: transforms far from the
    # solution.  Cut PCGs are counted; once the plan has seen them (two solves later) the budget covers them
    for _ in range(4):
        s.solve(*maps[6], *intr, A.Solve6Params(adaptive_launch=1, **kw))
    easy = s.stats()
    rng = np.random.default_rng(5)
    far = c["node_dq"].copy()
    far[:, 5:8] += rng.normal(0, 0.004, (len(far), 3)).astype(np.float32)
    far_dev = dev(far)  # borrowed by the plan: keep it alive
    s.set_node_transforms(far_dev)
    shorts = []
    for _ in range(5):
        s.solve(*maps[7], *intr, A.Solve6Params(adaptive_launch=1, **kw))
        shorts.append(s.stats()["pcg_short"])
    assert easy["pcg_short"] == 0 and shorts[-1] == 0 and shorts[0] >= shorts[-1]
    s.close()


def test_c4_solve_properties(A):
    """BASELINE config C4 (8 192 nodes, k = 8, 1 048 576 vertices, 1280 x 720 depth): the fp64 statement needs minutes at
    this size, so the HIP path is checked through properties — the block rows fit the plan, the energy of the
    linearisations decreases, most canonical vertices in view are associated, every transform is finite and unit, a
    second solve of the same inputs gives the same bits — and against the oracle on what IS cheap at this size: the
    energy of the first linearisation (no solve) and the blended warp of the solved transforms."""
    cfg, c, intr, depth = _scene("C4", 9)
    k = cfg["k"]
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    s = A.Solver6(cfg["D"], len(c["verts"]), k)
    keep = [dev(c["node_pos"]), dev(c["node_dq"]), dev(c["node_w"]), dev(c["verts"]), dev(c["normals"])]
    s.set_problem(*keep)
    prm = A.Solve6Params(num_iter=2, gn_iter=2, linear_iter=64, lambda_=200.0, pcg_tol=1e-3, pcg_tol_first=0.1, pcg_tol_decay=0.5)
    s.solve(P, Nm, *intr, prm)
    st = s.stats()
    dq = host(s.node_dq())
    assert st["overflow"] == 0 and st["max_row_blocks"] <= 48 and st["gn_iters"] == 4
    assert st["cost_hist"][1] < 0.05 * st["cost_hist"][0] and st["final_cost"] < 0.05 * st["initial_cost"]
    assert st["valid_first"] > 0.5 * len(c["verts"]) and st["valid_last"] > 0.5 * len(c["verts"])
    assert all(0 < n < 64 for n in st["pcg_it_hist"])
    assert np.isfinite(dq).all() and np.abs(np.linalg.norm(dq[:, :4], axis=1) - 1).max() < 1e-5
    e0, nv0 = O.cost6(c["node_pos"], c["node_dq"], c["node_w"], k, c["verts"], c["normals"], host(P), host(Nm), intr,
                      lambda_=200.0, threads=_threads())
    assert st["initial_cost"] == pytest.approx(e0, rel=1e-4) and abs(st["valid_first"] - nv0) <= 1e-3 * nv0 + 2
    wv, _ = s.warp()
    sample = np.arange(0, len(c["verts"]), 61)
    idx, wn, _ = O.graph6(c["node_pos"], c["node_w"], k, c["verts"][sample], threads=_threads())
    p_same, _ = O.warp6(dq, idx, wn, c["verts"][sample])
    assert np.abs(host(wv)[sample] - p_same).max() < 2e-6
    # the warped cloud has followed the depth frame: the energy at the solved transforms, re-associated, stays low
    e1, nv1 = O.cost6(c["node_pos"], dq, c["node_w"], k, c["verts"], c["normals"], host(P), host(Nm), intr, lambda_=200.0,
                      threads=_threads())
    assert e1 < 0.1 * e0 and nv1 > 0.5 * len(c["verts"])
    s.solve(P, Nm, *intr, prm)
    assert np.array_equal(host(s.node_dq()), dq)
    s.close()


def test_long_solve_stays_close_to_the_oracle_and_is_reproducible(A):
    cfg, c, intr, depth = _scene("T1", 6)
    kw = dict(num_iter=2, gn_iter=3, linear_iter=80, lambda_=200.0)
    s, dq, st, wv, wn_, dq_ref, st_ref = _solve_both(A, cfg, c, intr, depth, c["node_dq"], **kw)
    assert st["gn_iters"] == 6 and st["final_cost"] < 0.05 * st["initial_cost"]
    assert st["final_cost"] == pytest.approx(st_ref["final_cost"], rel=0.02)
    idx, wn, _ = O.graph6(c["node_pos"], c["node_w"], cfg["k"], c["verts"])
    p_ref, _ = O.warp6(dq_ref, idx, wn, c["verts"])
    assert np.linalg.norm(wv - p_ref, axis=1).mean() < 2e-4
    # same inputs -> same bits (sorted transposed lists, fixed summation orders, no float atomics)
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    for _ in range(3):
        s.solve(P, Nm, *intr, A.Solve6Params(**kw))
        assert np.array_equal(host(s.node_dq()), dq)


def test_solve_from_perturbed_transforms_and_without_normals(A):
    cfg, c, intr, depth = _scene("T0", 3)
    rng = np.random.default_rng(5)
    dq0 = c["node_dq"].copy()
    for i in range(len(dq0)):
        dq0[i] = O.apply_twist6(c["node_pos"][i], dq0[i], np.r_[rng.normal(0, 0.01, 3), rng.normal(0, 0.002, 3)])
    kw = dict(num_iter=1, gn_iter=2, linear_iter=100, lambda_=500.0)
    s, dq, st, wv, wn_, dq_ref, st_ref = _solve_both(A, cfg, c, intr, depth, dq0, **kw)
    assert st["initial_cost"] == pytest.approx(st_ref["initial_cost"], rel=1e-4)
    # (64 nodes, 2 x 100 PCG iterations from a rough start: rounding differences are amplified along the nearly free modes)
    assert np.abs(dq - dq_ref).max() < 5e-3 and np.abs(dq - dq_ref).mean() < 2e-4
    # no canonical normals: the normal gate is skipped
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    s2 = A.Solver6(cfg["D"], len(c["verts"]), cfg["k"])
    keep = [dev(c["node_pos"]), dev(dq0), dev(c["node_w"]), dev(c["verts"])]
    s2.set_problem(*keep)
    s2.solve(P, Nm, *intr, A.Solve6Params(**kw))
    ref, st_ref2 = O.solve6(c["node_pos"], dq0, c["node_w"], cfg["k"], c["verts"], None, host(P), host(Nm), intr, **kw)
    st2 = s2.stats()
    assert st2["valid_first"] >= st["valid_first"]
    assert st2["initial_cost"] == pytest.approx(st_ref2["initial_cost"], rel=1e-4)


def test_no_live_data_regulariser_only_and_errors(A):
    import torch
    cfg, c, intr, depth = _scene("T0", 0)
    k = cfg["k"]
    empty = torch.full((cfg["height"], cfg["width"], 4), float("nan"), device="cuda")
    rng = np.random.default_rng(8)
    rough = c["node_dq"].copy()
    for i in range(len(rough)):
        rough[i] = O.apply_twist6(c["node_pos"][i], rough[i], np.r_[rng.normal(0, 0.05, 3), rng.normal(0, 0.01, 3)])
    s = A.Solver6(cfg["D"], len(c["verts"]), k)
    keep = [dev(c["node_pos"]), dev(rough), dev(c["node_w"]), dev(c["verts"]), dev(c["normals"])]
    s.set_problem(*keep)
    kw = dict(num_iter=2, gn_iter=3, linear_iter=200, lambda_=200.0, psi_reg=1.0, damping=1e-6)
    s.solve(empty, empty, *intr, A.Solve6Params(**kw))
    st = s.stats()
    ref, st_ref = O.solve6(c["node_pos"], rough, c["node_w"], k, c["verts"], c["normals"], host(empty), host(empty), intr, **kw)
    assert st["valid_first"] == 0 and st["initial_cost"] == pytest.approx(st_ref["initial_cost"], rel=1e-4)
    assert st["final_cost"] < 0.05 * st["initial_cost"]
    # zero iterations: transforms returned unchanged
    s.solve(empty, empty, *intr, A.Solve6Params(num_iter=0))
    assert np.array_equal(host(s.node_dq()), rough)
    with pytest.raises(A.DynfuAmdError):
        A.Solver6(16, 100, 9)  # k out of range
    with pytest.raises(A.DynfuAmdError):
        s.solve(empty, empty, *intr, A.Solve6Params(psi_data=0.0))
    small = A.Solver6(8, 10, 4)
    with pytest.raises(A.DynfuAmdError):
        small.set_problem(*keep)  # larger than the plan


def test_few_nodes_with_very_long_row_lists(A):
    """12 nodes under 32 768 vertices (k = 4): ~11 000 rows per node — longer than the 4 096 the pattern kernel sorts in LDS
    (the list stays in the order the transposition left it), 44 000 pairs per node (the slot bytes go through global scratch
    instead of LDS), ~25 staged passes per workgroup and record batches that are refilled inside a pass.  Energies and
    iteration counts against the oracle."""
    cfg, c, intr, depth = _scene("T1", 6)
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    D, k = 12, 4
    sel = np.linspace(0, len(c["node_pos"]) - 1, D).astype(int)
    nodes, dq, w = c["node_pos"][sel].copy(), c["node_dq"][sel].copy(), np.full(D, 0.6, np.float32)
    verts, normals = c["verts"].copy(), c["normals"].copy()
    kw = dict(num_iter=1, gn_iter=2, linear_iter=60, lambda_=100.0, pcg_tol=1e-3)  # (a 72-unknown system: below 1e-3 fp32 stalls)
    s = A.Solver6(D, len(verts), k)
    keep = [dev(nodes), dev(dq), dev(w), dev(verts), dev(normals)]
    s.set_problem(*keep)
    s.solve(P, Nm, *intr, A.Solve6Params(**kw))
    st = s.stats()
    ref, st_ref = O.solve6(nodes, dq, w, k, verts, normals, host(P), host(Nm), intr, threads=_threads(), **kw)
    assert len(verts) * k // D > 4096 * 2 and st["overflow"] == 0
    assert st["valid_first"] == st_ref["valid_first"] and st["gn_iters"] == st_ref["gn_iters"] == 2
    assert st["initial_cost"] == pytest.approx(st_ref["initial_cost"], rel=1e-4)
    assert st["final_cost"] == pytest.approx(st_ref["final_cost"], rel=2e-3)
    assert abs(st["pcg_iters"] - st_ref["pcg_iters"]) <= 0.2 * st_ref["pcg_iters"] + 2
    assert np.abs(host(s.node_dq()) - ref).max() < 2e-3
    s.close()


def test_degenerate_problems(A):
    import torch
    cfg, c, intr, depth = _scene("T0", 2)
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    kw = dict(num_iter=1, gn_iter=2, linear_iter=30, lambda_=100.0)
    # fewer nodes than k: neighbour lists are padded with -1
    D, k = 3, 4
    nodes, dq, w = c["node_pos"][:D].copy(), c["node_dq"][:D].copy(), np.full(D, 0.4, np.float32)
    verts, normals = c["verts"][::16].copy(), c["normals"][::16].copy()
    s = A.Solver6(D, len(verts), k)
    keep = [dev(nodes), dev(dq), dev(w), dev(verts), dev(normals)]
    s.set_problem(*keep)
    s.solve(P, Nm, *intr, A.Solve6Params(**kw))
    st = s.stats()
    ref, st_ref = O.solve6(nodes, dq, w, k, verts, normals, host(P), host(Nm), intr, **kw)
    assert st["overflow"] == 0 and st["valid_first"] == st_ref["valid_first"]
    assert st["initial_cost"] == pytest.approx(st_ref["initial_cost"], rel=1e-4)
    assert np.isfinite(host(s.node_dq())).all() and np.abs(host(s.node_dq()) - ref).max() < 5e-3
    # a single node, no vertices at all: nothing to do, transforms unchanged
    s1 = A.Solver6(1, 0, 1)
    one = [dev(nodes[:1]), dev(dq[:1]), dev(w[:1]), torch.zeros((0, 3), device="cuda")]
    s1.set_problem(*one)
    s1.solve(P, Nm, *intr, A.Solve6Params(**kw))
    assert np.allclose(host(s1.node_dq()), dq[:1], atol=1e-6)
    # vertices behind the camera / far outside the image are simply not associated
    far = verts.copy()
    far[:, 2] = -1.0
    s.set_problem(keep[0], keep[1], keep[2], dev(far), keep[4])
    s.solve(P, Nm, *intr, A.Solve6Params(**kw))
    assert s.stats()["valid_first"] == 0 and np.isfinite(host(s.node_dq())).all()


@pytest.mark.parametrize("name", ["T0", "T1"])
def test_vertices_without_a_nearest_node_are_passed_through(A, name):
    """a canonical vertex with NaN coordinates has no k-NN (ids -1): the vertex sort must still place it (the solver's
    arrays are a permutation of ALL vertices), the solve ignores it, and the warp hands it back at the caller's own
    index — with every other vertex exactly where a run without the broken vertices puts it"""
    import torch
    cfg, c, intr, depth = _scene(name, 3)
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    kw = dict(num_iter=1, gn_iter=2, linear_iter=30, lambda_=200.0)
    verts, normals = c["verts"].copy(), c["normals"].copy()
    N = len(verts)
    bad = np.array([0, 1, 77, N // 2, N - 1])
    verts[bad] = np.nan
    s = A.Solver6(cfg["D"], N, cfg["k"])
    # stale contents of a previous problem in the plan's buffers: a full problem first
    first = [dev(c["node_pos"]), dev(c["node_dq"]), dev(c["node_w"]), dev(c["verts"]), dev(c["normals"])]
    s.set_problem(*first)
    s.solve(P, Nm, *intr, A.Solve6Params(**kw))
    keep = [first[0], first[1], first[2], dev(verts), dev(normals)]
    s.set_problem(*keep)
    s.solve(P, Nm, *intr, A.Solve6Params(**kw))
    st = s.stats()
    wv, wn_ = s.warp()
    torch.cuda.synchronize()
    wv, wn_ = host(wv), host(wn_)
    assert st["overflow"] == 0 and np.isfinite(host(s.node_dq())).all()
    good = np.ones(N, bool)
    good[bad] = False
    assert np.isnan(wv[bad]).all() and np.isfinite(wv[good]).all()      # handed back where the caller put them
    assert np.array_equal(bits(wn_[bad]), bits(normals[bad]))            # their normals untouched
    # the same problem without the broken vertices: same transforms (the broken rows carry no weight), same warp
    s2 = A.Solver6(cfg["D"], int(good.sum()), cfg["k"])
    keep2 = [first[0], first[1], first[2], dev(verts[good]), dev(normals[good])]
    s2.set_problem(*keep2)
    s2.solve(P, Nm, *intr, A.Solve6Params(**kw))
    assert s2.stats()["valid_first"] == st["valid_first"]
    assert np.abs(host(s2.node_dq()) - host(s.node_dq())).max() < 2e-5
    assert np.abs(host(s2.warp()[0]) - wv[good]).max() < 2e-5


def test_one_plan_across_problems_of_different_size(A):
    """the plan's captured PCG graphs are keyed by what they captured (the node count): a plan that alternates between
    problems of different node / vertex counts — the adaptor's plan after a node insertion — gives, for each, exactly what a
    fresh plan gives"""
    cfg, c, intr, depth = _scene("T1", 5)
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    kw = dict(num_iter=1, gn_iter=3, linear_iter=40, lambda_=200.0, pcg_tol=1e-3, pcg_tol_first=0.1, pcg_tol_decay=0.5)
    problems = []
    for D, step in ((cfg["D"], 1), (cfg["D"] - 37, 2), (cfg["D"], 1), (cfg["D"] // 2, 3)):
        problems.append([dev(c["node_pos"][:D]), dev(c["node_dq"][:D]), dev(c["node_w"][:D]), dev(c["verts"][::step]),
                         dev(c["normals"][::step])])
    shared = A.Solver6(cfg["D"], len(c["verts"]), cfg["k"])
    for prob in problems:
        for adaptive in (0, 1):
            shared.set_problem(*prob)
            shared.solve(P, Nm, *intr, A.Solve6Params(adaptive_launch=adaptive, **kw))
            got, st = host(shared.node_dq()).copy(), shared.stats()
            fresh = A.Solver6(int(prob[0].shape[0]), int(prob[3].shape[0]), cfg["k"])
            fresh.set_problem(*prob)
            fresh.solve(P, Nm, *intr, A.Solve6Params(adaptive_launch=0, **kw))
            want, st_f = host(fresh.node_dq()).copy(), fresh.stats()
            fresh.close()
            assert st["overflow"] == 0 and st["pcg_it_hist"] == st_f["pcg_it_hist"]
            if st["pcg_short"] == 0:
                assert np.array_equal(got, want)
    shared.close()


# ---------------------------------------------------------------- Gauss-Newton stopping rule + step acceptance
_BENCH_PCG = dict(linear_iter=64, lambda_=200.0, pcg_tol=1e-3, pcg_tol_first=0.1, pcg_tol_adapt=0.9)


def _accepted(st):
    return [c for c, code in zip(st["cost_hist"], st["stop_hist"]) if code in (0, 1)]


@pytest.mark.parametrize("name,frame,outer,gn", [("T1", 6, 2, 3), ("C2", 7, 1, 5), ("C3", 11, 2, 5)])
def test_gn_stopping_rule_matches_the_oracle(A, name, frame, outer, gn):
    """dfa_solve6_params.gn_tol (the reference runs Opt with earlyOut = true and nonLinearIter as a cap:
    src/dynfu/dyn_fusion.cpp:183-189): with bench.py's own parameters at C2 (1 x 5) and C3 (2 x 5, Eisenstat-Walker forcing)
    the HIP path takes the decisions of the fp64 statement slot by slot — solved / converged / rejected / skipped —, ends
    on its energy, never lists an accepted linearisation above the one before it, and does no worse than the run that uses
    every iteration."""
    cfg, c, intr, depth = _scene(name, frame)
    kw = dict(_BENCH_PCG, num_iter=outer, gn_iter=gn, gn_tol=1e-3)
    s, dq, st, wv, wn_, dq_ref, st_ref = _solve_both(A, cfg, c, intr, depth, c["node_dq"], threads=_threads(), **kw)
    assert st["overflow"] == 0
    assert st["stop_hist"] == st_ref["stop_hist"], (st["stop_hist"], st_ref["stop_hist"], st["cost_hist"], st_ref["cost_hist"])
    assert (st["gn_solves"], st["gn_rejected"], st["gn_converged"], st["gn_iters"]) == \
           (st_ref["gn_solves"], st_ref["gn_rejected"], st_ref["gn_converged"], st_ref["gn_iters"])
    assert st["gn_solves"] < outer * gn  # the rule has something to say on these frames
    assert st["final_cost"] == pytest.approx(st_ref["final_cost"], rel=0.02)
    assert st["valid_hist"][0] == st["valid_first"] and abs(st["valid_hist"][0] - st_ref["valid_hist"][0]) <= 1e-3 * st_ref["valid_hist"][0] + 2
    for o in range(outer):
        costs = [cst for i, (cst, code) in enumerate(zip(st["cost_hist"], st["stop_hist"])) if code in (0, 1) and i // gn == o and i < outer * gn]
        assert all(b <= a * (1 + 1e-3) for a, b in zip(costs, costs[1:])), costs
    assert st["final_cost"] == _accepted(st)[-1]
    # skipped slots ran nothing
    for i, code in enumerate(st["stop_hist"]):
        if code != 0:
            assert st["pcg_it_hist"][i] == 0
    idx, wn, _ = O.graph6(c["node_pos"], c["node_w"], cfg["k"], c["verts"], threads=_threads())
    p_ref, _ = O.warp6(dq_ref, idx, wn, c["verts"])
    assert np.linalg.norm(wv - p_ref, axis=1).mean() < 5e-4
    # against the run that uses every iteration: fewer solves, an energy (re-associated, fresh weights: the oracle's cost of
    # the transforms each run returns) that is no higher
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    s.solve(P, Nm, *intr, A.Solve6Params(**dict(kw, gn_tol=0.0)))
    st_fix, dq_fix = s.stats(), host(s.node_dq())
    assert st_fix["gn_solves"] == st_fix["gn_iters"] == outer * gn and st_fix["stop_hist"] == [0] * (outer * gn)
    e_early, _ = O.cost6(c["node_pos"], dq, c["node_w"], cfg["k"], c["verts"], c["normals"], host(P), host(Nm), intr, lambda_=200.0, threads=_threads())
    e_fix, _ = O.cost6(c["node_pos"], dq_fix, c["node_w"], cfg["k"], c["verts"], c["normals"], host(P), host(Nm), intr, lambda_=200.0, threads=_threads())
    assert e_early <= 1.02 * e_fix, (e_early, e_fix)
    s.close()


def test_gn_rejected_step_is_undone_bit_for_bit(A):
    """A rejected step leaves the transforms of the last accepted linearisation: the same bits as a solve that was only
    given that many iterations.  A converged outer iteration keeps its last step: the bits of the solve that was given
    exactly the iterations it used."""
    cfg, c, intr, depth = _scene("C2", 7)
    P, Nm = A.compute_points_normals(dev(depth), *intr)
    s = A.Solver6(cfg["D"], len(c["verts"]), cfg["k"])
    keep = [dev(c["node_pos"]), dev(c["node_dq"]), dev(c["node_w"]), dev(c["verts"]), dev(c["normals"])]
    s.set_problem(*keep)
    kw = dict(_BENCH_PCG, num_iter=1, gn_iter=5)
    s.solve(P, Nm, *intr, A.Solve6Params(**kw, gn_tol=1e-3))
    st, dq = s.stats(), host(s.node_dq()).copy()
    assert 2 in st["stop_hist"], st["stop_hist"]  # this frame rejects a step (the oracle agrees: test above)
    j = st["stop_hist"].index(2)  # slot of the rejected linearisation: j steps were applied, the last one undone
    assert st["gn_solves"] == j and st["stop_hist"][j + 1:5] == [3] * (4 - j) and len(st["stop_hist"]) == 5
    s.solve(P, Nm, *intr, A.Solve6Params(**dict(kw, gn_iter=j - 1)))  # j - 1 steps, every one kept
    assert np.array_equal(host(s.node_dq()), dq)
    st_short = s.stats()
    assert st_short["cost_hist"] == st["cost_hist"][:j - 1]
    # gn_tol = 0.5: the second linearisation is "converged" unless the energy halved — here the first step takes it down by
    # more than that, the second does not
    s.solve(P, Nm, *intr, A.Solve6Params(**kw, gn_tol=0.5))
    st5 = s.stats()
    assert st5["stop_hist"][:3] == [0, 0, 1] and st5["gn_converged"] == 1 and st5["gn_solves"] == 2
    s.solve(P, Nm, *intr, A.Solve6Params(**dict(kw, gn_iter=2)))
    dq2 = host(s.node_dq()).copy()
    s.solve(P, Nm, *intr, A.Solve6Params(**kw, gn_tol=0.5))
    assert np.array_equal(host(s.node_dq()), dq2)  # converged: both steps kept
    s.close()


def test_gn_closing_check_and_sequences(A):
    """(a) An outer iteration that runs to its cap has its last step checked by a closing linearisation (one more history
    slot); (b) a sequence of frames through one plan with the launch budget on gives the same bits whether or not the host
    waits between solves, and the budget of the slots behind a stop decays instead of staying at the cap."""
    cfg, c, intr, _ = _scene("T1", 6)
    keep = [dev(c["node_pos"]), dev(c["node_dq"]), dev(c["node_w"]), dev(c["verts"]), dev(c["normals"])]
    frames = [6, 6, 6, 7, 7, 7, 8, 8]
    maps = {f: A.compute_points_normals(dev(synth.depth_frame(cfg, f)), *intr) for f in set(frames)}
    # (a) one iteration per outer iteration: nothing to compare inside it, the closing slot decides about the last step
    s = A.Solver6(cfg["D"], len(c["verts"]), cfg["k"])
    s.set_problem(*keep)
    s.solve(*maps[6], *intr, A.Solve6Params(**_BENCH_PCG, num_iter=1, gn_iter=1, gn_tol=1e-3))
    st = s.stats()
    assert len(st["stop_hist"]) == 2 and st["stop_hist"][0] == 0 and st["stop_hist"][1] in (1, 2) and st["gn_iters"] == 2
    assert st["stop_hist"][1] == 1 and st["final_cost"] == st["cost_hist"][1] < 0.1 * st["cost_hist"][0]  # a first step from rest: kept
    dq1 = host(s.node_dq()).copy()
    s.solve(*maps[6], *intr, A.Solve6Params(**_BENCH_PCG, num_iter=1, gn_iter=1))
    assert np.array_equal(host(s.node_dq()), dq1) and len(s.stats()["stop_hist"]) == 1
    s.close()

    # (b)
    kw = dict(_BENCH_PCG, num_iter=2, gn_iter=4, gn_tol=1e-3, adaptive_launch=1)

    def run(wait):
        s = A.Solver6(cfg["D"], len(c["verts"]), cfg["k"])
        s.set_problem(*keep)
        out = []
        for f in frames:
            s.solve(*maps[f], *intr, A.Solve6Params(**kw))
            if wait:
                out.append(s.stats())
        st, dq = s.stats(), host(s.node_dq()).copy()
        s.close()
        return out, st, dq

    seq, st_w, dq_w = run(True)
    _, st_a, dq_a = run(False)
    assert np.array_equal(dq_w, dq_a) and st_w["pcg_launches"] == st_a["pcg_launches"] and st_w["stop_hist"] == st_a["stop_hist"]
    assert all(x["gn_solves"] < 8 for x in seq)
    # solves 0, 1: the full cap on every slot; from solve 2 on what the slots needed — the slots behind a stop next to nothing
    assert seq[0]["pcg_launches"] == seq[1]["pcg_launches"] == 8 * 65 and all(x["pcg_launches"] < 8 * 65 // 3 for x in seq[2:])


def test_gn_stopping_rule_beyond_the_history_length(A):
    """more Gauss-Newton slots (5 x 8 + the closing check) than the statistics keep (DFA_SOLVE6_HIST = 32): the rule
    works on the state block's scalars, not on the histories — same counts and energy as the oracle, histories cut at 32."""
    cfg, c, intr, depth = _scene("T0", 4)
    kw = dict(_BENCH_PCG, num_iter=5, gn_iter=8, gn_tol=1e-3)
    s, dq, st, wv, wn_, dq_ref, st_ref = _solve_both(A, cfg, c, intr, depth, c["node_dq"], **kw)
    assert len(st["stop_hist"]) == len(st_ref["stop_hist"]) == 32 and st["stop_hist"] == st_ref["stop_hist"]
    assert (st["gn_solves"], st["gn_rejected"], st["gn_converged"], st["gn_iters"]) == \
           (st_ref["gn_solves"], st_ref["gn_rejected"], st_ref["gn_converged"], st_ref["gn_iters"])
    assert st["gn_solves"] < 40 and st["final_cost"] == pytest.approx(st_ref["final_cost"], rel=0.02)
    assert np.isfinite(dq).all()
    s.close()
