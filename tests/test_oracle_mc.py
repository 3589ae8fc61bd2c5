"""CPU tests of the marching-cubes row: the default case tables (host code of the library) and the
oracle restatement (oracle/mc_oracle.c) on analytic volumes.  The reference has no test for
marching cubes; the properties below are what a correct extraction must satisfy."""
import collections

import numpy as np
import pytest

import oracle as O
from mc_util import blob_volume, default_tables, pack

EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def _tables(which):
    if which == "default":
        return default_tables()
    t = O.ref_mc_tables()
    if t is None:
        pytest.skip("oracle/_ref/mc_tables.bin not built (no reference checkout)")
    return t


@pytest.mark.parametrize("which", ["default", "reference"])
def test_case_tables_are_consistent(which):
    tri, nv = _tables(which)
    assert tri.shape == (256, 16) and nv.shape == (256,)
    assert nv[0] == 0 and nv[255] == 0 and nv.max() <= 15
    for c in range(256):
        row = tri[c]
        assert nv[c] % 3 == 0 and np.all(row[nv[c]:] == -1) and np.all(row[:nv[c]] >= 0)
        crossed = {e for e, (a, b) in enumerate(EDGES) if ((c >> a) & 1) != ((c >> b) & 1)}
        assert set(row[:nv[c]].tolist()) == crossed  # exactly the crossed edges are used
        # every triangle side inside the cube is shared by two triangles of the case with opposite
        # directions, or lies on a cube face (closed oriented patches)
        sides = collections.Counter()
        for t in range(0, nv[c], 3):
            a, b, d = row[t:t + 3]
            for u, v in ((a, b), (b, d), (d, a)):
                sides[(u, v)] += 1
        for (u, v), n in sides.items():
            assert n == 1 and sides.get((v, u), 0) <= 1


def test_default_tables_match_the_reference_conventions():
    tri, nv = default_tables()
    assert tri[1, :3].tolist() == [0, 8, 3] and tri[254, :3].tolist() == [0, 3, 8]  # winding of the classic table
    ref = O.ref_mc_tables()
    if ref is not None:
        assert np.array_equal(nv, ref[1])  # same vertex count in every case
        for c in range(256):
            assert set(tri[c, :nv[c]].tolist()) == set(ref[0][c, :ref[1][c]].tolist())


@pytest.mark.parametrize("which", ["default", "reference"])
def test_oracle_mesh_is_closed_and_on_the_surface(which):
    tri, nv = _tables(which)
    dims = (40, 36, 44)
    X, Y, Z = dims
    z, y, x = np.meshgrid((np.arange(Z) + 0.5) / Z, (np.arange(Y) + 0.5) / Y, (np.arange(X) + 0.5) / X, indexing="ij")
    d = np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) - 0.3
    vol = pack(np.clip(d / 0.2, -1, 1), np.ones(d.shape, np.uint32))  # every voxel observed
    cell = np.array([1 / X, 1 / Y, 1 / Z], np.float32)
    pts, total, occ = O.marching_cubes(vol, cell, tri, nv)
    assert total == len(pts) and total % 3 == 0 and occ > 0 and np.all(pts[:, 3] == 1.0)
    # on the sphere (linear interpolation of a truncated distance: within a fraction of a cell)
    r = np.linalg.norm(pts[:, :3] - 0.5, axis=1)
    assert np.abs(r - 0.3).max() < 0.01
    # watertight: every undirected edge of the mesh belongs to exactly two triangles, with opposite directions
    # (a vertex on a shared cube edge is interpolated from either end by the two cubes, so its
    # coordinates may differ in the last bit: weld vertices closer than 1e-6)
    from scipy.spatial import cKDTree
    xyz = pts[:, :3].astype(np.float64)
    tree = cKDTree(xyz)
    label = np.array([min(nb) for nb in tree.query_ball_point(xyz, 1e-6)])
    t = pts[:, :3].reshape(-1, 3, 3)
    directed = collections.Counter()
    for a, b, c in label.reshape(-1, 3).tolist():
        if a == b or b == c or a == c:
            continue  # degenerate triangles (surface through a voxel centre) have zero area
        for p, q in ((a, b), (b, c), (c, a)):
            directed[(p, q)] += 1
    bad = sum(1 for (p, q), n in directed.items() if n != directed.get((q, p), 0))
    assert bad == 0
    # consistent orientation: with the reference's conventions (bit set where f < 0) the right-hand
    # normals point to the negative side, here the centre — for the reference's table and ours alike
    n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    c = t.mean(1) - 0.5
    big = np.linalg.norm(n, axis=1) > 1e-7
    assert np.all((n[big] * c[big]).sum(1) < 0)


def test_oracle_weight_zero_rule_bounds_and_capacity():
    tri, nv = default_tables()
    vol = blob_volume((24, 20, 28), seed=3)
    cell = np.array([0.01, 0.02, 0.03], np.float32)
    pts, total, occ = O.marching_cubes(vol, cell, tri, nv)
    assert total > 0 and len(pts) == total
    # truncated output = prefix of the full output
    part, total2, _ = O.marching_cubes(vol, cell, tri, nv, max_vertices=100)
    assert total2 == total and np.array_equal(part, pts[:100])
    # a volume with a zero weight everywhere produces nothing; so does an all-positive one
    assert O.marching_cubes(vol & np.uint32(0xffff), cell, tri, nv)[1] == 0
    assert O.marching_cubes(pack(np.full(vol.shape, 0.5), np.ones(vol.shape, np.uint32)), cell, tri, nv)[1] == 0
    # vertices stay inside the box of cell centres [0.5, dim - 0.5] * cell
    lo = 0.5 * cell
    hi = (np.array([24, 20, 28]) - 0.5) * cell
    assert np.all(pts[:, :3] >= lo - 1e-6) and np.all(pts[:, :3] <= hi + 1e-6)


def test_vertex_normals_of_a_sphere_point_outwards():
    """normals of the extracted vertices from the TSDF gradient (the raycaster's compute_normal).
    On the truncated distance field of a sphere they are the radial directions."""
    dims = (48, 48, 48)
    X, Y, Z = dims
    cell = np.array([2.0 / X, 2.0 / Y, 2.0 / Z], np.float32)
    z, y, x = np.meshgrid(np.arange(Z) * cell[2], np.arange(Y) * cell[1], np.arange(X) * cell[0], indexing="ij")
    centre, radius, trunc = np.array([1.0, 0.9, 1.1]), 0.55, 0.25
    d = np.sqrt((x - centre[0]) ** 2 + (y - centre[1]) ** 2 + (z - centre[2]) ** 2) - radius
    vol = pack(np.clip(d / trunc, -1, 1), np.full(d.shape, 5, np.uint32))
    tri, nv = default_tables()
    pts, total, _ = O.marching_cubes(vol, cell, tri, nv)
    assert total > 1000
    nrm = O.tsdf_vertex_normals(vol, cell, 0.5, pts)
    assert nrm.shape == pts.shape and np.isfinite(nrm[:, :3]).all() and (nrm[:, 3] == 0).all()
    np.testing.assert_allclose(np.linalg.norm(nrm[:, :3], axis=1), 1.0, atol=1e-5)
    radial = pts[:, :3] - centre
    radial /= np.linalg.norm(radial, axis=1, keepdims=True)
    assert (np.sum(radial * nrm[:, :3], axis=1) > 0.99).all()  # outward: the field grows away from the surface
    # a stencil that leaves the interpolation range [0, dim - 1) yields NaN, as in the raycaster
    edge = np.array([[0.0, 1.0, 1.0, 1.0], [(X - 1) * cell[0], 1.0, 1.0, 1.0]], np.float32)
    assert np.isnan(O.tsdf_vertex_normals(vol, cell, 0.5, edge)[:, 0]).all()
