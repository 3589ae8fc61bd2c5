"""-m gpu parity tests of the marching-cubes seam: HIP kernels (through the C ABI) vs the CPU oracle.

Bar: BIT-EXACT — vertex count, order (ascending linear voxel index) and the float bits of every
point, with the library's default case tables and with the reference's own tables
(oracle/_ref/mc_tables.bin, extracted at build time).  The oracle is unpinned (the reference has
no marching-cubes test), see oracle/mc_oracle.c."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from dynfu_amd import synth  # noqa: E402
from gpu_util import bits, dev, host  # noqa: E402
from mc_util import blob_volume, default_tables, pack  # noqa: E402


@pytest.fixture(scope="module")
def A():
    import dynfu_amd
    dynfu_amd.load()
    return dynfu_amd


def _tables(which):
    if which == "default":
        return default_tables()
    t = O.ref_mc_tables()
    if t is None:
        pytest.skip("oracle/_ref/mc_tables.bin not built")
    return t


def _run(A, vol, cell, tri, nv, cap):
    pts, total = A.marching_cubes(dev(vol), cell, dev(tri), dev(nv), cap)
    total = int(host(total)[0])
    return host(pts)[: min(total, cap)], total


@pytest.mark.parametrize("which", ["default", "reference"])
@pytest.mark.parametrize("dims", [(64, 64, 64), (128, 128, 128), (50, 38, 44), (256, 24, 40), (260, 9, 7), (2, 2, 2),
                                  (67, 5, 130)])
def test_marching_cubes_bit_exact(A, dims, which):
    tri, nv = _tables(which)
    vol = blob_volume(dims, seed=sum(dims))
    cell = np.array([3.0 / dims[0], 2.5 / dims[1], 3.5 / dims[2]], np.float32)
    ref, total, _ = O.marching_cubes(vol, cell, tri, nv)
    got, gtotal = _run(A, vol, cell, tri, nv, max(total, 1))
    assert gtotal == total
    assert np.array_equal(bits(got), bits(ref))


def test_marching_cubes_of_an_integrated_depth_frame(A):
    """the producer/consumer chain of DynFusion::operator(): integrate a depth frame, extract the mesh"""
    import torch
    cfg = synth.CONFIGS["T1"]
    fx, fy, cx, cy = synth.intrinsics(cfg)
    voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
    depth = synth.depth_frame(cfg, 0)
    dists = torch.empty(depth.shape, dtype=torch.uint16, device="cuda")
    A.compute_dists(dev(depth), dists, fx, fy, cx, cy)
    dim = cfg["dim"]
    vol = torch.empty((dim, dim, dim), dtype=torch.int32, device="cuda")
    A.tsdf_clear_integrate(vol, dists, voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy)
    tri, nv = default_tables()
    hv = host(vol).view(np.uint32)
    ref, total, occ = O.marching_cubes(hv, voxel, tri, nv)
    assert total > 10000 and occ > 3000
    pts, gtotal = A.marching_cubes(vol, voxel, dev(tri), dev(nv), total)
    assert int(host(gtotal)[0]) == total
    assert np.array_equal(bits(host(pts)), bits(ref))
    # the mesh is the depth surface: vertices (volume frame) -> camera frame lie on the sphere / plane
    p = ref[:, :3] + np.array(synth.VOLUME_POSE_T, np.float32)
    r = np.linalg.norm(p - synth.SPHERE_C, axis=1)
    on_sphere = np.abs(r - synth.SPHERE_R) < 0.03
    on_plane = np.abs(p[:, 2] - synth.PLANE_Z) < 0.03
    assert (on_sphere | on_plane).mean() > 0.97  # the rest: depth discontinuity at the silhouette


def test_marching_cubes_capacity_count_only_and_errors(A):
    import torch
    tri, nv = default_tables()
    vol = blob_volume((48, 40, 36), seed=5)
    cell = np.array([0.01, 0.01, 0.01], np.float32)
    ref, total, _ = O.marching_cubes(vol, cell, tri, nv)
    # count only
    _, t0 = A.marching_cubes(dev(vol), cell, dev(tri), dev(nv), 0)
    assert int(host(t0)[0]) == total
    # truncated: the first `cap` vertices, nothing written beyond
    cap = total // 3 + 1
    pts = torch.full((cap + 8, 4), -7.0, dtype=torch.float32, device="cuda")
    tot = torch.zeros(1, dtype=torch.int32, device="cuda")
    from dynfu_amd import _lib
    _lib._check(_lib.load().dfa_marching_cubes(_lib._dev(dev(vol)), 48, 40, 36, _lib._farr(cell, 3), _lib._dev(dev(tri)),
                                               _lib._dev(dev(nv)), _lib._dev(pts), cap, _lib._dev(tot), _lib._stream()))
    assert int(host(tot)[0]) == total
    assert np.array_equal(bits(host(pts)[:cap]), bits(ref[:cap])) and np.all(host(pts)[cap:] == -7.0)
    # empty volume, all-outside volume
    zero = np.zeros((8, 8, 8), np.uint32)
    assert int(host(A.marching_cubes(dev(zero), cell, dev(tri), dev(nv), 10)[1])[0]) == 0
    pos = pack(np.full((8, 8, 8), 0.25), np.ones((8, 8, 8), np.uint32))
    assert int(host(A.marching_cubes(dev(pos), cell, dev(tri), dev(nv), 10)[1])[0]) == 0
    with pytest.raises(A.DynfuAmdError):
        A.marching_cubes(dev(vol), cell, None, dev(nv), 10)


@pytest.mark.parametrize("dims,delta", [((64, 64, 64), 0.5), ((50, 38, 44), 0.75), ((128, 128, 128), 0.5)])
def test_vertex_normals_bit_exact(A, dims, delta):
    """dfa_tsdf_vertex_normals vs the oracle's compute_normal on the vertices marching cubes emits,
    plus points whose gradient stencil leaves the volume (NaN on both sides)"""
    tri, nv = default_tables()
    vol = blob_volume(dims, seed=7, holes=False)
    cell = np.array([3.0 / dims[0], 2.5 / dims[1], 3.5 / dims[2]], np.float32)
    pts, total, _ = O.marching_cubes(vol, cell, tri, nv)
    assert total > 100
    extra = np.array([[0, 0, 0, 1], [cell[0] * (dims[0] - 1), 0.5, 0.5, 1], [-1, 0.5, 0.5, 1]], np.float32)
    pts = np.concatenate([pts, extra]).astype(np.float32)
    ref = O.tsdf_vertex_normals(vol, cell, delta, pts)
    got = host(A.tsdf_vertex_normals(dev(vol), cell, delta, dev(pts)))
    assert np.isnan(ref[-3:, 0]).all()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(bits(got)[ok], bits(ref)[ok])
    assert host(A.tsdf_vertex_normals(dev(vol), cell, delta, dev(pts[:0]))).shape == (0, 4)
