"""CPU tests of the north-star (6-DoF) solve's oracle, oracle/solve6_oracle.c.

The mode is not in the reference's code (PARITY UNPINNED, see the file header); what pins the
oracle is checked here: its analytic Jacobians against finite differences of its own warp, its
Gauss-Newton step against a dense least-squares solve of the same linearisation built in numpy, and
the behaviour of the whole solve on synthetic motion."""
import numpy as np
import pytest

import oracle as O
from dynfu_amd import synth


@pytest.fixture(scope="module")
def scene():
    cfg = synth.CONFIGS["T0"]
    c = synth.canonical(cfg)
    intr = synth.intrinsics(cfg)
    P, Nm = O.points_normals(synth.depth_frame(cfg, 5), *intr)
    return cfg, c, intr, P, Nm


def _perturbed(c, seed, rot=0.05, trans=0.01):
    rng = np.random.default_rng(seed)
    dq = c["node_dq"].copy()
    for i in range(len(dq)):
        dq[i] = O.apply_twist6(c["node_pos"][i], dq[i], np.r_[rng.normal(0, rot, 3), rng.normal(0, trans, 3)])
    return dq


def test_points_normals_follow_the_reference_kernel():
    # imgproc.cu:187-215 on a tilted plane z = 1 + 0.001 x (mm-quantised): normal ~ (-a, 0, 1)/|.| sign-flipped
    W, H, fx, fy, cx, cy = 40, 30, 100.0, 100.0, 19.5, 14.5
    depth = np.full((H, W), 1500, np.uint16)
    depth[10, 7] = 0
    P, Nm = O.points_normals(depth, fx, fy, cx, cy)
    assert np.isnan(P[:, -1]).all() and np.isnan(P[-1]).all()          # last row / column: :198
    assert np.isnan(P[10, 7]).all() and np.isnan(P[10, 6]).all() and np.isnan(P[9, 7]).all()  # any zero of the 3: :206
    ok = np.isfinite(P[..., 0])
    assert np.allclose(P[ok][:, 2], 1.5) and np.all(P[ok][:, 3] == 0)
    assert np.allclose(P[5, 5, :2], [1.5 * (5 - cx) / fx, 1.5 * (5 - cy) / fy], atol=1e-6)
    assert np.allclose(Nm[ok][:, :3], [0, 0, -1], atol=1e-6)           # fronto-parallel wall, normal towards the camera


def test_blend_is_rigid_when_all_nodes_agree_and_weights_are_normalised(scene):
    cfg, c, intr, P, Nm = scene
    k = cfg["k"]
    idx, wn, reg = O.graph6(c["node_pos"], c["node_w"], k, c["verts"])
    assert np.allclose(wn.sum(1), 1, atol=1e-6) and np.all(np.diff(idx, axis=1) != 0)
    assert np.all(reg != np.arange(len(reg))[:, None]) and np.all(reg >= 0)
    # one rigid motion for every node
    tw = np.array([0.02, -0.03, 0.05, 0.01, 0.02, -0.015])
    dq = np.stack([O.apply_twist6(np.zeros(3), q, tw) for q in c["node_dq"]])  # twist about the origin
    p, n = O.warp6(dq, idx, wn, c["verts"], c["normals"])
    th = np.linalg.norm(tw[:3])
    K = np.array([[0, -tw[2], tw[1]], [tw[2], 0, -tw[0]], [-tw[1], tw[0], 0]]) / th
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    assert np.abs(p - (c["verts"] @ R.T + tw[3:])).max() < 2e-6
    assert np.abs(n - c["normals"] @ R.T).max() < 2e-6


def test_data_jacobian_matches_finite_differences(scene):
    cfg, c, intr, P, Nm = scene
    k = cfg["k"]
    idx, wn, _ = O.graph6(c["node_pos"], c["node_w"], k, c["verts"])
    dq = _perturbed(c, 1)
    h, worst = 2e-3, 0.0
    for v in (3, 500, 4000, 8000):
        J, p = O.data_jacobian6(c["node_pos"], dq, idx[v], wn[v], c["verts"][v])
        for j in range(k):
            for col in range(6):
                tw = np.zeros(6)
                tw[col] = h
                n = idx[v, j]
                plus, minus = dq.copy(), dq.copy()
                plus[n] = O.apply_twist6(c["node_pos"][n], dq[n], tw)
                minus[n] = O.apply_twist6(c["node_pos"][n], dq[n], -tw)
                pp = O.data_jacobian6(c["node_pos"], plus, idx[v], wn[v], c["verts"][v])[1]
                pm = O.data_jacobian6(c["node_pos"], minus, idx[v], wn[v], c["verts"][v])[1]
                worst = max(worst, np.abs((pp - pm) / (2 * h) - J[j, col]).max())
    assert worst < 2e-4  # fp32 storage of the perturbed transforms limits the difference quotient


def _dense_step(c, dq, k, P, Nm, intr, prm):
    """one Gauss-Newton step from the formulas of DESIGN.md §4.5, dense, in numpy"""
    D = len(c["node_pos"])
    idx, wn, reg = O.graph6(c["node_pos"], c["node_w"], k, c["verts"])
    p, nw = O.warp6(dq, idx, wn, c["verts"], c["normals"])
    rows, rhs, wts = [], [], []
    fx, fy, cx, cy = intr
    Himg, Wimg = P.shape[:2]
    for v in range(len(p)):
        pv = p[v].astype(np.float64)
        if not pv[2] > 0:
            continue
        u, w = int(np.rint(fx * pv[0] / pv[2] + cx)), int(np.rint(fy * pv[1] / pv[2] + cy))
        if not (0 <= u < Wimg and 0 <= w < Himg) or np.isnan(P[w, u, 0]) or np.isnan(Nm[w, u, 0]):
            continue
        J, pd = O.data_jacobian6(c["node_pos"], dq, idx[v], wn[v], c["verts"][v])
        L, n = P[w, u, :3].astype(np.float64), Nm[w, u, :3].astype(np.float64)
        if np.linalg.norm(pd - L) > prm["dist_thresh"] or nw[v] @ n < prm["cos_thresh"]:
            continue
        r = n @ (pd - L)
        e = abs(r) / prm["tukey_offset"]
        rho = (1 - e * e / prm["psi_data"] ** 2) ** 2 if e < prm["psi_data"] else 0.0
        row = np.zeros(6 * D)
        for j in range(k):
            row[6 * idx[v, j]:6 * idx[v, j] + 6] += J[j] @ n
        rows.append(row), rhs.append(r), wts.append(rho)
    # regulariser
    def apply(q, x):
        return O.warp6(q[None], np.zeros((1, 1), np.int32), np.ones((1, 1), np.float32), x[None])[0][0].astype(np.float64)
    wreg2 = prm["lambda_"] / (D * k)
    ghat = np.stack([apply(dq[i], c["node_pos"][i]) for i in range(D)])
    for n in range(D):
        for m in reg[n]:
            y = apply(dq[n], c["node_pos"][m])
            e = y - ghat[m]
            en = np.linalg.norm(e)
            h = 1.0 if en <= prm["psi_reg"] else prm["psi_reg"] / en
            l = y - ghat[n]
            S = np.array([[0, l[2], -l[1]], [-l[2], 0, l[0]], [l[1], -l[0], 0]])
            for cc in range(3):
                row = np.zeros(6 * D)
                row[6 * n:6 * n + 3] = S[cc]
                row[6 * n + 3 + cc] = 1
                row[6 * m + 3 + cc] = -1
                rows.append(row), rhs.append(e[cc]), wts.append(wreg2 * h)
    Jm, r, w = np.array(rows), np.array(rhs), np.array(wts)
    Hm = Jm.T @ (w[:, None] * Jm) + prm["damping"] * np.eye(6 * D)
    g = -Jm.T @ (w * r)
    return np.linalg.solve(Hm, g).reshape(D, 6), float((w * r * r).sum())


def test_gauss_newton_step_equals_a_dense_solve(scene):
    cfg, c, intr, P, Nm = scene
    k = cfg["k"]
    prm = dict(O.SOLVE6_DEFAULTS, lambda_=50.0, num_iter=1, gn_iter=1, linear_iter=3000, pcg_tol=1e-12)
    dq0 = _perturbed(c, 2, rot=0.01, trans=0.003)
    delta, cost = _dense_step(c, dq0, k, P, Nm, intr, prm)
    dq1, st = O.solve6(c["node_pos"], dq0, c["node_w"], k, c["verts"], c["normals"], P, Nm, intr, **prm)
    assert st["initial_cost"] == pytest.approx(cost, rel=1e-6)  # the numpy side reads fp32 node positions back
    want = np.stack([O.apply_twist6(c["node_pos"][i], dq0[i], delta[i]) for i in range(len(dq0))])
    assert np.abs(dq1 - want).max() < 2e-6


def test_solve_pulls_the_canonical_surface_onto_the_live_depth(scene):
    cfg, c, intr, P, Nm = scene
    k = cfg["k"]
    kw = dict(num_iter=3, gn_iter=3, linear_iter=200, lambda_=200.0)
    dq, st = O.solve6(c["node_pos"], c["node_dq"], c["node_w"], k, c["verts"], c["normals"], P, Nm, intr, **kw)
    assert st["gn_iters"] == 9 and st["valid_first"] > 0.4 * len(c["verts"])
    assert st["final_cost"] < 0.25 * st["initial_cost"]
    c_after, nv = O.cost6(c["node_pos"], dq, c["node_w"], k, c["verts"], c["normals"], P, Nm, intr, **kw)
    rms0 = np.sqrt(st["initial_cost"] / st["valid_first"])
    rms1 = np.sqrt(c_after / nv)
    assert rms1 < 0.5 * rms0  # 2.6 mm -> 1.1 mm; what is left is the millimetre quantisation of the 160x120 depth map
    # (point-to-plane leaves sliding along the surface free, so the transforms themselves are not unique:
    # a second solve from the solution keeps the energy, not the node poses)
    _, st2 = O.solve6(c["node_pos"], dq, c["node_w"], k, c["verts"], c["normals"], P, Nm, intr, **kw)
    assert st2["final_cost"] < 1.1 * c_after


def test_regulariser_alone_keeps_a_rigid_field_and_smooths_a_rough_one(scene):
    cfg, c, intr, P, Nm = scene
    k = cfg["k"]
    empty = np.full_like(P, np.nan)  # no live data at all: only the regulariser acts
    kw = dict(num_iter=2, gn_iter=3, linear_iter=300, lambda_=200.0, psi_reg=1.0, damping=1e-9)
    tw = np.array([0.03, 0.01, -0.02, 0.02, -0.01, 0.04])
    rigid = np.stack([O.apply_twist6(np.zeros(3), q, tw) for q in c["node_dq"]])
    out, st = O.solve6(c["node_pos"], rigid, c["node_w"], k, c["verts"], c["normals"], empty, empty, intr, **kw)
    assert st["valid_first"] == 0 and st["initial_cost"] < 1e-12 and np.abs(out - rigid).max() < 1e-6
    rough = _perturbed(c, 3, rot=0.1, trans=0.02)
    out, st = O.solve6(c["node_pos"], rough, c["node_w"], k, c["verts"], c["normals"], empty, empty, intr, **kw)
    assert st["final_cost"] < 0.05 * st["initial_cost"]


def test_forcing_schedules_set_the_tolerance_each_linearisation_stops_at(scene):
    """Inexact Newton: the geometric schedule is pcg_tol_first * decay^i floored at pcg_tol; the
    Eisenstat-Walker one starts there and is then gamma * (rz0_i / rz0_(i-1)), clamped to [pcg_tol, pcg_tol_first].  Each PCG
    stops at the first iterate under its tolerance, and either schedule ends within a few percent of the tight solve."""
    cfg, c, intr, P, Nm = scene
    k = cfg["k"]
    base = dict(num_iter=2, gn_iter=3, linear_iter=400, lambda_=200.0, pcg_tol=1e-4)
    run = lambda **kw: O.solve6(c["node_pos"], c["node_dq"], c["node_w"], k, c["verts"], c["normals"], P, Nm, intr,
                                **dict(base, **kw))[1]
    tight = run()
    assert np.allclose(tight["pcg_tol_hist"], 1e-4, rtol=1e-6)
    geo = run(pcg_tol_first=0.1, pcg_tol_decay=0.1)  # the schedule restarts with every re-association (outer iteration)
    assert np.allclose(geo["pcg_tol_hist"], [0.1, 0.01, 1e-3] * 2, rtol=1e-5)
    assert np.allclose(run(pcg_tol_first=0.1, pcg_tol_decay=0.1, num_iter=1, gn_iter=6)["pcg_tol_hist"],
                       [0.1, 0.01, 1e-3, 1e-4, 1e-4, 1e-4], rtol=1e-5)
    ew = run(pcg_tol_first=0.1, pcg_tol_adapt=0.9)
    tol = np.array(ew["pcg_tol_hist"])
    assert tol[0] == pytest.approx(0.1, rel=1e-6) and np.all(tol >= 1e-4 * (1 - 1e-6)) and np.all(tol <= 0.1 * (1 + 1e-6))
    assert np.any((tol > 1e-4 * 1.01) & (tol < 0.1 * 0.99))  # it follows the residuals: a value strictly inside the clamp
    for st in (tight, geo, ew):
        rel, it, t = map(np.array, (st["pcg_rel_hist"], st["pcg_it_hist"], st["pcg_tol_hist"]))
        assert np.all((rel <= t * (1 + 1e-5)) | (it >= 400))
    assert geo["pcg_iters"] < tight["pcg_iters"] and ew["pcg_iters"] < tight["pcg_iters"]
    assert geo["final_cost"] < 1.05 * tight["final_cost"] and ew["final_cost"] < 1.05 * tight["final_cost"]


def test_projective_association_oracle_obeys_its_gates():
    """orc_correspond_projective (the gates of find_coresp, proj_icp.cu:72-98) on a hand-made map"""
    fx = fy = 2.0
    cx = cy = 1.5
    P = np.full((4, 4, 4), np.nan, np.float32)
    Nm = np.full((4, 4, 4), np.nan, np.float32)
    for y in range(4):
        for x in range(4):
            P[y, x] = [(x - cx) / fx, (y - cy) / fy, 1.0, 0.0]
            Nm[y, x] = [0, 0, -1, 0]
    P[0, 0, 0] = np.nan          # undefined pixel
    Nm[3, 3] = [1, 0, 0, 0]      # normal at right angles
    v = np.array([[-0.25, -0.25, 1.0],   # pixel (1, 1): association
                  [-0.75, -0.75, 1.0],   # pixel (0, 0): undefined
                  [0.75, 0.75, 1.0],     # pixel (3, 3): normal gate
                  [-0.25, -0.25, 1.5],   # lands on a pixel 0.5 m away: distance gate
                  [5.0, 0.0, 1.0],       # outside the image
                  [0.0, 0.0, -1.0]], np.float32)
    n = np.tile([[0, 0, -1]], (len(v), 1)).astype(np.float32)
    ov, on, pix = O.correspond_projective(v, n, P, Nm, fx, fy, cx, cy, 0.1, 0.5)
    assert pix.tolist() == [1 * 4 + 1, -1, -1, -1, -1, -1]
    np.testing.assert_array_equal(ov[0], P[1, 1, :3])
    np.testing.assert_array_equal(on[0], [0, 0, -1])
    assert np.isnan(ov[1:]).all() and np.isnan(on[1:]).all()
    # without normals the third vertex is associated too
    _, _, pix2 = O.correspond_projective(v, None, P, Nm, fx, fy, cx, cy, 0.1, 0.5)
    assert pix2.tolist() == [5, -1, 15, -1, -1, -1]


# ---------------------------------------------------------------- Gauss-Newton stopping rule (orc6_params.gn_tol)
_PCG = dict(linear_iter=64, lambda_=200.0, pcg_tol=1e-3, pcg_tol_first=0.1, pcg_tol_adapt=0.9)


def test_gn_stopping_rule_of_the_oracle(scene):
    """oracle.h: gn_tol.  With the rule the oracle (a) never lists an accepted linearisation above the one before it inside
    an outer iteration, (b) leaves, behind a rejected step, exactly the transforms of the solve that was only given the
    accepted iterations, (c) ends on the energy of its last accepted linearisation, (d) marks what it did per slot, and (e)
    with gn_tol = 0 is the fixed-iteration solve of before, slot for slot."""
    cfg, c, intr, P, Nm = scene
    k = cfg["k"]
    args = (c["node_pos"], c["node_dq"], c["node_w"], k, c["verts"], c["normals"], P, Nm, intr)
    dq_fix, st_fix = O.solve6(*args, num_iter=2, gn_iter=4, **_PCG)
    assert st_fix["stop_hist"] == [0] * 8 and st_fix["gn_solves"] == st_fix["gn_iters"] == 8 and st_fix["hist_n"] == 8
    assert st_fix["final_cost"] == st_fix["cost_hist"][-1] and st_fix["valid_hist"][0] == st_fix["valid_first"]
    dq, st = O.solve6(*args, num_iter=2, gn_iter=4, gn_tol=1e-3, **_PCG)
    codes = st["stop_hist"]
    assert len(codes) in (8, 9) and set(codes) <= {0, 1, 2, 3} and st["gn_solves"] == codes.count(0) < 8
    assert st["gn_rejected"] == codes.count(2) and st["gn_iters"] == len(codes) - codes.count(3)
    for o in range(2):
        slots = [i for i in range(4 * o, 4 * o + 4) if codes[i] in (0, 1)]
        costs = [st["cost_hist"][i] for i in slots]
        assert all(b <= a * (1 + 1e-3) for a, b in zip(costs, costs[1:])), costs
        if 2 in codes[4 * o:4 * o + 4]:  # everything behind a rejection is skipped, and the rejected energy was higher
            j = codes.index(2, 4 * o)
            assert codes[j + 1:4 * o + 4] == [3] * (4 * o + 3 - j) and st["cost_hist"][j] > costs[-1] * (1 + 1e-3)
    accepted = [st["cost_hist"][i] for i, cd in enumerate(codes) if cd in (0, 1)]
    assert st["final_cost"] == accepted[-1]
    # (b) one outer iteration: the solve with the rule == the fixed solve given only the accepted steps
    dq1, st1 = O.solve6(*args, num_iter=1, gn_iter=6, gn_tol=1e-3, **_PCG)
    c1 = st1["stop_hist"]
    if 2 in c1:
        j = c1.index(2)
        dq_ref, _ = O.solve6(*args, num_iter=1, gn_iter=j - 1, **_PCG)
    else:
        dq_ref, _ = O.solve6(*args, num_iter=1, gn_iter=st1["gn_solves"], **_PCG)
    assert np.array_equal(dq1, dq_ref)
    # the returned transforms are no worse than the fixed run's (energy re-associated at fresh weights)
    e_rule, _ = O.cost6(c["node_pos"], dq, c["node_w"], k, c["verts"], c["normals"], P, Nm, intr, lambda_=200.0)
    e_fix, _ = O.cost6(c["node_pos"], dq_fix, c["node_w"], k, c["verts"], c["normals"], P, Nm, intr, lambda_=200.0)
    assert e_rule <= 1.05 * e_fix


def test_gn_closing_check_and_matrix_reuse_of_the_oracle(scene):
    cfg, c, intr, P, Nm = scene
    args = (c["node_pos"], c["node_dq"], c["node_w"], cfg["k"], c["verts"], c["normals"], P, Nm, intr)
    # one iteration per outer iteration: only the closing slot can judge the step
    dq, st = O.solve6(*args, num_iter=1, gn_iter=1, gn_tol=1e-3, **_PCG)
    assert st["stop_hist"] == [0, 1] and st["final_cost"] == st["cost_hist"][1] < st["cost_hist"][0]
    dq0, st0 = O.solve6(*args, num_iter=1, gn_iter=1, **_PCG)
    assert np.array_equal(dq, dq0) and st0["stop_hist"] == [0]
    # reuse_matrix: iteration 0 is full Gauss-Newton, so a one-iteration solve is unchanged; later iterations differ
    dq_r1, _ = O.solve6(*args, num_iter=1, gn_iter=1, reuse_matrix=1, **_PCG)
    assert np.array_equal(dq_r1, dq0)
    dq_r3, st_r3 = O.solve6(*args, num_iter=1, gn_iter=3, reuse_matrix=1, **_PCG)
    dq_f3, st_f3 = O.solve6(*args, num_iter=1, gn_iter=3, **_PCG)
    assert st_r3["cost_hist"][:2] == st_f3["cost_hist"][:2] and not np.array_equal(dq_r3, dq_f3)
