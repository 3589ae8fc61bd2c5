"""-m gpu parity tests of the warp-field seam (k-NN graph, RBF weights, DQ warp) vs the oracle.

Bar: k-NN indices bit-exact (integer work); weights within 1 float ulp (double exp on two
different libms, rounded to float); warped vertices within 2e-6 m absolute (float DQ algebra
with identical operation order, exp as above)."""
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


def _ulp_diff(a, b):
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


@pytest.mark.parametrize("D,k,n", [(2048, 4, 20000), (500, 8, 5000), (3000, 16, 3000), (5, 8, 100), (1, 4, 10),
                                    (1025, 5, 777)])
def test_knn_bit_exact_and_weights(A, D, k, n):
    rng = np.random.default_rng(D + k)
    nodes = rng.uniform(-1, 1, (D, 3)).astype(np.float32)
    node_w = rng.uniform(0.05, 0.5, D).astype(np.float32)
    q = rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32)
    q[: min(n, D)] = nodes[: min(n, D)]  # queries sitting exactly on nodes (distance 0)
    idx, w = A.knn(dev(nodes), dev(node_w), dev(q), k)
    ref = O.knn(nodes, q, k, threads=8)
    assert np.array_equal(host(idx), ref)
    wref = np.zeros((n, k), np.float32)
    for v in range(0, n, max(1, n // 500)):
        for j in range(k):
            if ref[v, j] >= 0:
                wref[v, j] = O.transformation_weight(nodes[ref[v, j]], float(node_w[ref[v, j]]), q[v])
        assert _ulp_diff(host(w)[v], wref[v]).max() <= 1
    assert np.all(host(w)[ref < 0] == 0)


def test_knn_on_two_streams_from_one_thread(A):
    # the search grid of dfa_knn is internal scratch kept per (device, stream): two searches in flight on two streams,
    # driven by ONE host thread, over different node sets must not see each other's grid
    import torch
    rng = np.random.default_rng(3)
    sets = []
    for i in range(2):
        D, n = 1500 + 700 * i, 40000
        nodes = rng.uniform(-1 - i, 1 + i, (D, 3)).astype(np.float32)
        q = rng.uniform(-1 - i, 1 + i, (n, 3)).astype(np.float32)
        sets.append((dev(nodes), dev(np.full(D, 0.1, np.float32)), dev(q), O.knn(nodes, q, 4, threads=8)))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    out = [None, None]
    for rep in range(6):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                out[i] = A.knn(sets[i][0], sets[i][1], sets[i][2], 4)[0]
    torch.cuda.synchronize()
    for i in range(2):
        assert np.array_equal(host(out[i]), sets[i][3])


@pytest.mark.parametrize("kind", ["clustered", "planar", "far_queries", "duplicates", "line", "lattice"])
@pytest.mark.parametrize("k", [4, 8])
def test_knn_grid_path_is_exact_on_awkward_geometry(A, kind, k):
    # sizes above the grid threshold (D * n >= 2^22); the uniform grid must return exactly the
    # exhaustive (distance, index)-ordered answer whatever the node distribution
    rng = np.random.default_rng(hash(kind) % 1000 + k)
    D, n = 1500, 6000
    q = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    if kind == "clustered":
        centres = rng.uniform(-1, 1, (6, 3))
        nodes = (centres[rng.integers(0, 6, D)] + rng.normal(0, 0.01, (D, 3))).astype(np.float32)
    elif kind == "planar":
        nodes = rng.uniform(-1, 1, (D, 3)).astype(np.float32)
        nodes[:, 2] = 0.25  # zero extent along z
    elif kind == "far_queries":
        nodes = rng.uniform(-0.1, 0.1, (D, 3)).astype(np.float32)
        q = (q * 50).astype(np.float32)  # far outside the node bounding box
    elif kind == "duplicates":
        nodes = np.repeat(rng.uniform(-1, 1, (D // 10, 3)), 10, axis=0).astype(np.float32)
    elif kind == "line":
        nodes = np.zeros((D, 3), np.float32)
        nodes[:, 0] = np.linspace(-1, 1, D)
    else:  # integer lattice: massive exact ties
        g = np.stack(np.meshgrid(*[np.arange(12.0)] * 3, indexing="ij"), -1).reshape(-1, 3)
        nodes = g[:D].astype(np.float32)
        q = rng.integers(0, 12, (n, 3)).astype(np.float32) + np.float32(0.5)
    idx, _ = A.knn(dev(nodes), dev(np.full(len(nodes), 0.3, np.float32)), dev(q), k)
    assert np.array_equal(host(idx), O.knn(nodes, q, k, threads=8))


def test_knn_exact_ties_keep_lower_index(A):
    # integer lattice: many exactly equal distances
    g = np.stack(np.meshgrid(*[np.arange(6.0)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    q = (g[::5] + np.float32(0.5)).astype(np.float32)
    idx, _ = A.knn(dev(g), dev(np.ones(len(g), np.float32)), dev(q), 8)
    assert np.array_equal(host(idx), O.knn(g, q, 8))


@pytest.mark.parametrize("name", ["T0", "T1"])
def test_warp_to_live_matches_oracle(A, name):
    cfg = synth.CONFIGS[name]
    c = synth.canonical(cfg)
    D, k = cfg["D"], cfg["k"]
    rng = np.random.default_rng(5)
    # general rigid node transforms (rotation + translation), not only the solver's translations
    dq = np.stack([O.dq_from_euler(*rng.uniform(-0.2, 0.2, 3), *rng.uniform(-0.05, 0.05, 3)) for _ in range(D)])
    verts, normals = c["verts"][:20000], c["normals"][:20000]
    ov, on = A.warp_to_live(dev(c["node_pos"]), dev(dq), dev(c["node_w"]), k, dev(verts), dev(normals))
    rv, rn = O.warp_to_live(c["node_pos"], dq, c["node_w"], k, verts, normals, threads=8)
    np.testing.assert_allclose(host(ov), rv, atol=2e-6, rtol=0)
    np.testing.assert_allclose(host(on), rn, atol=2e-6, rtol=0)
    # identity transforms warp nothing
    ov, _ = A.warp_to_live(dev(c["node_pos"]), dev(c["node_dq"]), dev(c["node_w"]), k, dev(verts))
    assert np.array_equal(host(ov), verts)


@pytest.mark.parametrize("name", ["T1", "C1"])
def test_warp_with_a_given_graph_equals_the_warp_that_searches(A, name):
    cfg = synth.CONFIGS[name]
    c = synth.canonical(cfg)
    D, k = cfg["D"], cfg["k"]
    rng = np.random.default_rng(9)
    dq = np.stack([O.dq_from_euler(*rng.uniform(-0.2, 0.2, 3), *rng.uniform(-0.05, 0.05, 3)) for _ in range(D)])
    verts, normals = c["verts"], c["normals"]
    args = (dev(c["node_pos"]), dev(dq), dev(c["node_w"]))
    ov, on = A.warp_to_live(*args, k, dev(verts), dev(normals))
    idx, _ = A.knn(args[0], args[2], dev(verts), k)
    gv, gn = A.warp_to_live_graph(*args, idx, dev(verts), dev(normals))
    assert np.array_equal(bits(host(gv)), bits(host(ov))) and np.array_equal(bits(host(gn)), bits(host(on)))


def test_warp_empty_and_errors(A):
    import torch
    nodes = torch.zeros((4, 3), device="cuda")
    dq = torch.zeros((4, 8), device="cuda")
    w = torch.ones(4, device="cuda")
    ov, _ = A.warp_to_live(nodes, dq, w, 4, torch.zeros((0, 3), device="cuda"))
    assert ov.shape == (0, 3)
    with pytest.raises(A.DynfuAmdError):
        A.knn(nodes, w, torch.zeros((3, 3), device="cuda"), 17)


# ------------------------------------------------------------------ correspondence (dyn_fusion.cpp:212-242)
def _surface_clouds(n_canon, n_live, seed):
    """canonical = points on the synthetic sphere cap, live = a displaced, differently sampled set"""
    rng = np.random.default_rng(seed)
    def cap(n):
        d = rng.normal(size=(n, 3))
        d[:, 2] = -np.abs(d[:, 2])
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        return (synth.SPHERE_C + 0.5 * d).astype(np.float32), d.astype(np.float32)
    cv, cn = cap(n_canon)
    lv, _ = cap(n_live)
    lv += rng.normal(0, 0.004, lv.shape).astype(np.float32)
    return cv, cn, lv


@pytest.mark.parametrize("n_canon,n_live", [(65536, 65536), (3000, 10000), (17, 1000), (1, 5), (40000, 129)])
def test_correspond_bit_exact(A, n_canon, n_live):
    cv, cn, lv = _surface_clouds(n_canon, n_live, n_canon + n_live)
    lv[: min(n_live, n_canon) // 2] = cv[: min(n_live, n_canon) // 2]  # exact hits
    ov, on, idx = A.correspond(dev(cv), dev(cn), dev(lv))
    rv, rn, ridx = O.correspond(cv, cn, lv, threads=8)
    assert np.array_equal(host(idx), ridx)
    assert np.array_equal(host(ov), rv) and np.array_equal(host(on), rn)
    # the reference's own KD-tree (nanoflann, oracle/_ref) agrees wherever the nearest point is unique
    ref = O.ref_knn(cv, lv[:20000], 1)
    if ref is not None:
        d = ((lv[:20000].astype(np.float64) - cv[ref[0][:, 0]].astype(np.float64)) ** 2).sum(1)
        dm = ((lv[:20000].astype(np.float64) - rv[:20000].astype(np.float64)) ** 2).sum(1)
        assert np.all((ref[0][:, 0] == ridx[:20000]) | (np.abs(d - dm) <= 1e-12))


@pytest.mark.parametrize("n,shift_cells", [(120000, 1.6), (120000, 2.7), (600000, 2.2), (120000, 6.0)])
def test_correspond_a_surface_that_has_moved_by_cells(A, n, shift_cells):
    """A live surface one to a few grid cells away from the canonical one (along its normal: the 3 x 3 x 3 block of a query
    is then empty or holds only far points): the search leaves shells 0 / 1 and takes the exact ball of cells around the
    query (up to four cells of reach) or, beyond that, the shell walk.  Same indices as the oracle's exhaustive scan."""
    cv, cn, lv = _surface_clouds(n, 40000, n + int(10 * shift_cells))
    # the grid's cell size as pgrid_finalize_kernel picks it: the cap of 128 (256 above half a million points) cells per axis
    ext = cv.max(0) - cv.min(0)
    cs = max((0.5 if n > 500000 else 0.7) * float(np.cbrt(ext.prod() / n)), float(ext.max()) / (256 if n > 500000 else 128))
    d = lv - synth.SPHERE_C
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rng = np.random.default_rng(3)
    lv = (lv + d * (shift_cells * cs * rng.uniform(0.6, 1.0, (len(lv), 1)))).astype(np.float32)
    _, _, idx = A.correspond(dev(cv), None, dev(lv))
    _, _, ridx = O.correspond(cv, None, lv, threads=8)
    assert np.array_equal(host(idx), ridx)


def test_correspond_a_sparse_neighbourhood_grows_its_ball_without_scanning_it_again(A, devlib, monkeypatch):
    """Queries two to four cells away from the cloud find their 3 x 3 x 3 block EMPTY: the ball of cells starts at two cells
    and grows by one until it holds a point, and a grown ball scans only what the larger radius adds (the first
    form scanned the whole ball again on every growth — DFA_BALL_RESCAN=1 in the development library keeps it).  Same indices
    as the oracle's exhaustive scan and as the first form; not slower than it."""
    import torch
    n = 600000
    cv, cn, lv = _surface_clouds(n, 200000, n + 5)
    ext = cv.max(0) - cv.min(0)
    cs = max(0.5 * float(np.cbrt(ext.prod() / n)), float(ext.max()) / 256)
    d = lv - synth.SPHERE_C
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rng = np.random.default_rng(9)
    lv = (lv + d * (cs * rng.uniform(2.2, 4.4, (len(lv), 1)))).astype(np.float32)  # every block empty, two to three growths
    _, _, ridx = O.correspond(cv, None, lv, threads=8)
    dcv, dlv = dev(cv), dev(lv)
    ms = {}
    for form in ("grow", "rescan"):
        if form == "rescan":
            monkeypatch.setenv("DFA_BALL_RESCAN", "1")
        _, _, idx = A.correspond(dcv, None, dlv)
        assert np.array_equal(host(idx), ridx), form
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            A.correspond(dcv, None, dlv)
        e1.record()
        torch.cuda.synchronize()
        ms[form] = e0.elapsed_time(e1) / 5
    monkeypatch.delenv("DFA_BALL_RESCAN")
    assert ms["grow"] <= 1.1 * ms["rescan"], ms


def test_correspond_a_million_points_against_the_reference_kd_tree(A):
    """Clouds of more than half a million points switch to the finer grid (up to 256 cells per axis): a million canonical
    points on the synthetic surface, checked against the reference's own nanoflann KD-tree (oracle/_ref) on 60 000 queries
    and by the defining property on all of them (no canonical point of a random sample is closer than the one returned)."""
    cv, cn, lv = _surface_clouds(1 << 20, 300000, 77)
    ov, on, idx = A.correspond(dev(cv), dev(cn), dev(lv))
    idx_h, ov_h = host(idx), host(ov)
    assert np.array_equal(ov_h, cv[idx_h]) and np.array_equal(host(on), cn[idx_h])
    d_ret = ((lv.astype(np.float64) - ov_h.astype(np.float64)) ** 2).sum(1)
    rng = np.random.default_rng(1)
    for _ in range(4):  # random canonical points are never closer than the returned one
        cand = cv[rng.integers(0, len(cv), len(lv))]
        assert np.all(((lv.astype(np.float64) - cand.astype(np.float64)) ** 2).sum(1) >= d_ret - 1e-12)
    ref = O.ref_knn(cv, lv[:60000], 1)
    if ref is not None:
        r = ref[0][:, 0]
        d_ref = ((lv[:60000].astype(np.float64) - cv[r].astype(np.float64)) ** 2).sum(1)
        assert np.all((r == idx_h[:60000]) | (np.abs(d_ref - d_ret[:60000]) <= 1e-12))
        assert (r == idx_h[:60000]).mean() > 0.999


def test_correspond_duplicates_optional_outputs_and_errors(A):
    import torch
    cv = np.tile(np.array([[0, 0, 1], [0.5, 0, 1], [0, 0, 1]], np.float32), (50, 1))  # every point 50 (or 100) times
    lv = np.array([[0, 0, 1.01], [0.4, 0, 1], [9, 9, 9]], np.float32)
    ov, on, idx = A.correspond(dev(cv), None, dev(lv))
    assert on is None and host(idx).tolist() == [0, 1, 1]  # ties -> lowest index
    assert np.array_equal(host(ov), cv[[0, 1, 1]])
    ov, on, idx = A.correspond(dev(cv), dev(cv), torch.zeros((0, 3), device="cuda"))
    assert ov.shape == (0, 3) and idx.shape == (0,)
    with pytest.raises(A.DynfuAmdError):
        A.correspond(torch.zeros((0, 3), device="cuda"), None, dev(lv))


@pytest.mark.parametrize("kind", ["planar", "coincident", "clustered_far", "lattice", "line", "volume"])
def test_correspond_point_grid_is_exact_on_awkward_geometry(A, kind):
    rng = np.random.default_rng(11)
    n = 30000
    if kind == "planar":
        cv = np.c_[rng.uniform(-1, 1, (n, 2)), np.full(n, 1.25)]
    elif kind == "coincident":
        cv = np.tile([[0.25, -0.5, 2.0]], (n, 1))
    elif kind == "clustered_far":
        cv = np.r_[rng.normal(0, 0.01, (n - 10, 3)), rng.uniform(50, 60, (10, 3))]
    elif kind == "lattice":  # many exactly equidistant candidates
        g = np.arange(31) * 0.125
        cv = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    elif kind == "line":
        cv = np.c_[np.linspace(-3, 3, n), np.zeros(n), np.zeros(n)]
    else:
        cv = rng.uniform(-1, 1, (n, 3))
    cv = cv.astype(np.float32)
    lo, hi = cv.min(0), cv.max(0)
    lv = np.r_[rng.uniform(lo - 0.3, hi + 0.3, (3000, 3)), cv[::50] + 0.0625,  # lattice: cell centres
               rng.uniform(-100, 100, (200, 3))].astype(np.float32)
    _, _, idx = A.correspond(dev(cv), None, dev(lv))
    _, _, ridx = O.correspond(cv, None, lv, threads=8)
    assert np.array_equal(host(idx), ridx)


# ------------------------------------------------------------- node insertion pieces (warp_field.cpp:34-95)
@pytest.mark.parametrize("D,k,n", [(300, 8, 20000), (2048, 8, 60000), (3, 8, 500), (8192, 8, 150000), (4096, 4, 50000)])
def test_unsupported_vertices_and_calc_dqb_match_the_oracle(A, D, k, n):
    rng = np.random.default_rng(D)
    nodes = rng.uniform(-1, 1, (D, 3)).astype(np.float32)
    node_w = rng.uniform(0.05, 0.3, D).astype(np.float32)
    verts = rng.uniform(-1.3, 1.3, (n, 3)).astype(np.float32)
    verts[:D] = nodes  # distance 0: supported
    flags = host(A.unsupported_vertices(dev(nodes), dev(node_w), k, dev(verts)))
    ref = O.unsupported_flags(nodes, node_w, k, verts, threads=8)
    assert np.array_equal(flags, ref) and 0 < ref.mean() < 1 and not ref[:D].any()
    # blended transforms at the first points
    dq = np.zeros((D, 8), np.float32)
    for i in range(D):
        dq[i] = O.dq_from_euler(*rng.uniform(-0.3, 0.3, 3), *rng.uniform(-0.05, 0.05, 3))
    m = min(n, 400)
    got = host(A.calc_dqb(dev(nodes), dev(dq), dev(node_w), k, dev(verts[:m])))
    want = np.stack([O.calc_dqb(nodes, dq, node_w, k, verts[i]) for i in range(m)])
    assert np.abs(got - want).max() < 2e-6


def test_unsupported_vertices_without_nodes(A):
    import torch
    v = torch.zeros((7, 3), device="cuda")
    assert host(A.unsupported_vertices(None, None, 8, v)).tolist() == [1] * 7  # min stays HUGE_VALF (:40,:53)


# ------------------------------------------------------------- projective association
def _live_maps(cfg, frame):
    fx, fy, cx, cy = synth.intrinsics(cfg)
    P, Nm = O.points_normals(synth.depth_frame(cfg, frame), fx, fy, cx, cy)
    return (fx, fy, cx, cy), P, Nm


@pytest.mark.parametrize("name,with_normals", [("T1", True), ("T1", False), ("C1", True)])
def test_correspond_projective_bit_exact(A, name, with_normals):
    """dfa_correspond_projective vs the oracle's find_coresp gates: the canonical cloud of the synthetic scene against
    the vertex / normal maps of a later frame, plus vertices behind the camera, outside the image and on the far plane"""
    cfg = synth.CONFIGS[name]
    c = synth.canonical(cfg)
    intr, P, Nm = _live_maps(cfg, 5)
    extra = np.array([[0, 0, -1], [0, 0, 0], [50, 0, 1], [0, -50, 1], [0.01, 0.01, 2.5], [np.nan, 0, 1]], np.float32)
    v = np.concatenate([c["verts"], extra]).astype(np.float32)
    n = np.concatenate([c["normals"], np.tile([[0, 0, -1]], (len(extra), 1))]).astype(np.float32)
    args = (intr[0], intr[1], intr[2], intr[3], 0.05, 0.7)
    rv, rn, rp = O.correspond_projective(v, n if with_normals else None, P, Nm, *args)
    gv, gn, gp = A.correspond_projective(dev(v), dev(n) if with_normals else None, dev(P), dev(Nm), *args)
    assert np.array_equal(host(gp), rp)
    assert np.array_equal(bits(host(gv)), bits(rv)) and np.array_equal(bits(host(gn)), bits(rn))
    matched = rp >= 0
    assert 0.5 * len(c["verts"]) < matched.sum() < len(v)          # most of the visible surface is associated
    assert (rp[-6:-2] == -1).all() and rp[-1] == -1                # behind / outside / NaN: no association
    assert np.isnan(rv[~matched]).all() and np.isfinite(rv[matched]).all()
    # every association obeys the gates
    assert (np.linalg.norm(rv[matched] - v[matched], axis=1) <= 0.05 * (1 + 1e-6)).all()
    if with_normals:
        assert (np.abs(np.sum(rn[matched] * n[matched], axis=1)) >= 0.7 - 1e-6).all()
    # without a normal map: vertices only
    gv2, gn2, gp2 = A.correspond_projective(dev(v), None, dev(P), None, *args)
    rv2, _, rp2 = O.correspond_projective(v, None, P, None, *args)
    assert gn2 is None and np.array_equal(host(gp2), rp2) and np.array_equal(bits(host(gv2)), bits(rv2))


def test_correspond_projective_errors(A):
    import torch
    P = torch.zeros((4, 4, 4), device="cuda")
    v = torch.zeros((3, 3), device="cuda")
    with pytest.raises(A.DynfuAmdError):
        A.correspond_projective(v, None, P, None, 0.0, 1.0, 0.0, 0.0, 0.1, 0.5)  # fx = 0
    ov, on, pix = A.correspond_projective(torch.zeros((0, 3), device="cuda"), None, P, None, 1.0, 1.0, 0.0, 0.0, 0.1, 0.5)
    assert ov.shape == (0, 3) and pix.shape == (0,)


# ---- point-cloud plumbing between the seams (csrc/points.hip): bit copies, checked against numpy
@pytest.mark.parametrize("n", [0, 1, 63, 4096, 4097, 100003])
@pytest.mark.parametrize("ss,ds", [(4, 3), (3, 4), (3, 3), (4, 4), (5, 7)])
def test_repack_points_is_a_bit_copy_of_xyz(A, n, ss, ds):
    rng = np.random.default_rng(n + ss)
    src = rng.standard_normal((n, ss)).astype(np.float32)
    if n:
        src[0, :3] = [np.nan, -0.0, np.inf]
    out = host(A.repack_points(dev(src), ds, pad=1.0)) if n else np.zeros((0, ds), np.float32)
    assert out.shape == (n, ds)
    assert np.array_equal(bits(out[:, :3]), bits(src[:, :3]))
    assert np.all(out[:, 3:] == 1.0)


@pytest.mark.parametrize("n,density", [(0, 0.5), (1, 1.0), (15, 0.5), (4096, 0.0), (4096, 1.0), (4097, 0.01), (70001, 0.3),
                                        (1 << 20, 0.001), (3 * (1 << 20) + 5, 0.5)])
def test_compact_points_keeps_flagged_points_in_index_order(A, n, density):
    import torch
    rng = np.random.default_rng(n)
    pts = rng.standard_normal((n, 3)).astype(np.float32)
    flags = (rng.random(n) < density).astype(np.uint8) * rng.integers(1, 256, n).astype(np.uint8)  # any non-zero value counts
    out, idx = A.compact_points(dev(pts) if n else None, dev(flags) if n else torch.zeros(0, dtype=torch.uint8, device="cuda"))
    want = np.flatnonzero(flags)
    assert np.array_equal(host(idx), want.astype(np.int32))
    assert np.array_equal(bits(host(out)), bits(pts[want]))
    # a flag array that does not start on a 16-byte boundary (the kernel's wide loads must not assume it)
    if n > 100:
        f2 = dev(flags)[3:]
        out2, idx2 = A.compact_points(dev(pts[3:]), f2)
        assert np.array_equal(host(idx2), np.flatnonzero(flags[3:]).astype(np.int32))


@pytest.mark.parametrize("n", [1, 257, 100003])
def test_transform_points_matches_the_host_loop_bit_for_bit(A, n):
    from gpu_util import rot
    rng = np.random.default_rng(n)
    p = rng.standard_normal((n, 3)).astype(np.float32)
    R = rot([0.3, -1.0, 0.5], 0.7).astype(np.float32)
    t = np.array([-1.5, -1.5, 0.5], np.float32)
    aff = np.concatenate([R.reshape(-1), t])
    for with_t in (True, False):
        out = host(A.transform_points(dev(p), aff, with_t))
        ref = np.stack([(R[c, 0] * p[:, 0] + R[c, 1] * p[:, 1]) + R[c, 2] * p[:, 2] for c in range(3)], 1)
        if with_t:
            ref = ref + t
        assert np.array_equal(bits(out), bits(ref.astype(np.float32)))
