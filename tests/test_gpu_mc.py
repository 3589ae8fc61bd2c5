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


def _boxes(hv):
    """the map the sweep must at least produce, per box of 32 x 2 x 8 voxels (x, y, z): (holds a voxel with a weight, holds
    a voxel with a weight and a negative distance)"""
    Z, Y, X = hv.shape
    w = (hv >> 16) != 0
    neg = w & ((hv & 0x8000) != 0) & ((hv & 0x7fff) != 0)

    def boxes(b):
        pad = np.zeros(((Z + 7) // 8 * 8, (Y + 1) // 2 * 2, (X + 31) // 32 * 32), bool)
        pad[:Z, :Y, :X] = b
        return pad.reshape(pad.shape[0] // 8, 8, pad.shape[1] // 2, 2, pad.shape[2] // 32, 32).any(axis=(1, 3, 5))

    return boxes(w), boxes(neg)


@pytest.mark.parametrize("name,dims", [("T1", None), ("T0", None), ("T1", (100, 77, 90)), ("C2", None)])
def test_occupancy_map_and_marching_cubes_without_the_empty_voxels(A, name, dims):
    """dfa_tsdf_clear_integrate_occ / dfa_tsdf_integrate_occ / dfa_tsdf_clear_occ keep a byte per box of 32 x 2 x 8 voxels;
    dfa_marching_cubes_occ reads it instead of the empty voxels (the reference's OccupiedVoxels pass reads all of them,
    src/kfusion/cuda/marching_cubes.cu:77-142).  The volume is the one the plain entry points write (bit for bit), the map
    covers every voxel with a weight and is sparse, the mesh is the plain entry point's and the oracle's, bit for bit —
    fused sweep, accumulating sweep over three frames, ragged dimensions, BASELINE size."""
    import torch
    cfg = synth.CONFIGS[name]
    fx, fy, cx, cy = synth.intrinsics(cfg)
    voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
    X, Y, Z = dims or (cfg["dim"],) * 3
    if dims:
        voxel = tuple(float(synth.VOLUME_SIZE / d) for d in dims)
    tri, nv = default_tables()
    dists = []
    for f in range(3):
        d = torch.empty((cfg["height"], cfg["width"]), dtype=torch.uint16, device="cuda")
        A.compute_dists(dev(synth.depth_frame(cfg, f)), d, fx, fy, cx, cy)
        dists.append(d)
    vol = torch.empty((Z, Y, X), dtype=torch.int32, device="cuda")
    plain = torch.empty_like(vol)
    occ = A.tsdf_occupancy(vol)
    assert tuple(occ.shape) == ((Z + 7) // 8, (Y + 1) // 2, (X + 31) // 32)
    occ.fill_(7)  # the fused sweep must write every byte

    def check(cap=None):
        hv = host(vol).view(np.uint32)
        assert np.array_equal(hv, host(plain).view(np.uint32))
        m, full = (host(occ) & 1) != 0, (host(occ) & 2) != 0
        need, need_neg = _boxes(hv)
        assert not (need & ~m).any() and not (need_neg & ~full).any()  # supersets of the boxes with weights / negative distances
        assert m.sum() < 3 * max(1, need.sum()) and not (full & ~m).any()  # ... and not much more
        # (boxes of 32 x 2 x 8 voxels against the frustum / a band of ~5 voxels around the surface: coarse at 64^3 and
        # 128^3, where most boxes hold something; at 512^3 a third of the volume has weights and a tenth may be negative)
        if X >= 512:
            assert m.mean() < 0.4 and full.mean() < 0.2, (m.mean(), full.mean())
        ref_pts, ref_total = A.marching_cubes(plain, voxel, dev(tri), dev(nv), cap or 1)
        total = int(host(ref_total)[0])
        ref_pts, _ = A.marching_cubes(plain, voxel, dev(tri), dev(nv), total)
        pts, got = A.marching_cubes(vol, voxel, dev(tri), dev(nv), total, occupancy=occ)
        assert int(host(got)[0]) == total > 1000
        assert np.array_equal(bits(host(pts)), bits(host(ref_pts)))
        return hv, total, host(pts)

    A.tsdf_clear_integrate(plain, dists[0], voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy)
    A.tsdf_clear_integrate(vol, dists[0], voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy, occupancy=occ)
    assert set(np.unique(host(occ))) <= {0, 1, 3}
    hv, total, pts = check()
    if X * Y * Z <= 128 ** 3:  # the CPU statement of marching cubes beside it
        ref, rtotal, _ = O.marching_cubes(hv, np.asarray(voxel, np.float32), tri, nv)
        assert rtotal == total and np.array_equal(bits(pts), bits(ref))
    # the accumulating sweep: two more frames into the same volume; the map only grows
    before = host(occ) != 0
    for f in (1, 2):
        A.tsdf_integrate(plain, dists[f], voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy)
        A.tsdf_integrate(vol, dists[f], voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy, occupancy=occ)
    assert not (before & ~(host(occ) != 0)).any()
    check()
    # clear: volume and map
    A.tsdf_clear(vol, occupancy=occ)
    assert not host(occ).any() and not host(vol).any()
    pts, got = A.marching_cubes(vol, voxel, dev(tri), dev(nv), 16, occupancy=occ)
    assert int(host(got)[0]) == 0
    # a map of ones is the plain sweep
    A.tsdf_clear_integrate(vol, dists[1], voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy)
    A.tsdf_clear_integrate(plain, dists[1], voxel, trunc, synth.MAX_WEIGHT, vol2cam, fx, fy, cx, cy)
    occ.fill_(3)
    _, t1 = A.marching_cubes(vol, voxel, dev(tri), dev(nv), 1, occupancy=occ)
    _, t0 = A.marching_cubes(plain, voxel, dev(tri), dev(nv), 1)
    assert int(host(t1)[0]) == int(host(t0)[0]) > 1000


@pytest.mark.parametrize("name,dims", [("T1", None), ("T1", (100, 77, 90)), ("C2", None), ("C4", None)])
def test_fused_sweep_over_a_known_occupancy_map_leaves_the_same_volume_and_map(A, name, dims):
    """dfa_tsdf_clear_integrate_known_occ: with a map that describes the volume on entry, boxes of zeros that stay zeros are
    not stored again.  Frame after frame with a moving camera — behind a clear, behind fused sweeps, behind accumulating
    sweeps — the volume and the map are those of dfa_tsdf_clear_integrate_occ (and so of the plain fused sweep, and at the
    small sizes of the oracle), bit for bit; the documented failure with a map that does NOT describe the volume is there
    too (old content survives where the map says "zeros")."""
    import torch
    cfg = synth.CONFIGS[name]
    fx, fy, cx, cy = synth.intrinsics(cfg)
    voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
    X, Y, Z = dims or (cfg["dim"],) * 3
    if dims:
        voxel = tuple(float(synth.VOLUME_SIZE / d) for d in dims)
    dists = []
    for f in range(4):
        d = torch.empty((cfg["height"], cfg["width"]), dtype=torch.uint16, device="cuda")
        A.compute_dists(dev(synth.depth_frame(cfg, f)), d, fx, fy, cx, cy)
        dists.append(d)

    def pose(f):  # the camera a little to the side and back every frame: other boxes come into the frustum
        m = np.array(vol2cam, np.float32).copy()
        m[9] += 0.04 * f
        m[11] += 0.03 * f
        return m

    vol = torch.empty((Z, Y, X), dtype=torch.int32, device="cuda").random_(0, 2 ** 31 - 1)
    ref = torch.empty_like(vol)
    occ, occ_ref = A.tsdf_occupancy(vol), A.tsdf_occupancy(vol)
    args = (voxel, trunc, synth.MAX_WEIGHT)

    def same():  # (on the device: 1024^3 is 4 GiB a volume)
        assert torch.equal(vol, ref) and torch.equal(occ, occ_ref)

    A.tsdf_clear(vol, occupancy=occ)  # from here on the map describes the volume
    occ_first = None
    for f in range(4):
        A.tsdf_clear_integrate(vol, dists[f], *args, pose(f), fx, fy, cx, cy, occupancy=occ, occupancy_known=True)
        A.tsdf_clear_integrate(ref, dists[f], *args, pose(f), fx, fy, cx, cy, occupancy=occ_ref)
        same()
        if f == 0:
            occ_first = occ.clone()
    if X >= 1024:
        # The one place where "old content survives" would show, against the ORACLE (the comparison above is with the plain
        # HIP sweep): the 32 slices with the most boxes that frame 0 wrote and frame 3 — the camera has moved — left alone
        # because their byte said zeros... which they are only if frames 1-3 put the zeros back when the boxes fell out of
        # the frustum.  orc_tsdf_integrate_slab (tsdf_volume.cu:11-22 clear + :43-96) on zeros = what a cleared volume
        # holds after frame 3.
        gone = ((occ_first != 0) & (occ == 0)).sum(dim=(1, 2))  # per layer of 8 slices
        kept = (occ != 0).sum(dim=(1, 2))
        win = gone.unfold(0, 4, 1).sum(1)                        # per window of four layers
        win_kept = kept.unfold(0, 4, 1).sum(1)
        win = torch.where(win_kept > 1000, win, torch.zeros_like(win))  # (a window frame 3 also writes in: not an empty slab)
        assert int(win.max()) > 1000, int(win.max())             # (thousands of boxes left the frustum)
        z0 = 8 * int(win.argmax())
        slab = np.zeros((32, Y, X), np.uint32)
        O.tsdf_integrate_slab(slab, z0, host(dists[3]), np.asarray(voxel, np.float32), trunc, synth.MAX_WEIGHT, pose(3), fx, fy, cx, cy,
                              threads=8)
        assert np.array_equal(host(vol[z0:z0 + 32]).view(np.uint32), slab), z0
        assert (slab >> 16 != 0).any()  # (not an empty slab: part of it is inside the frustum of frame 3)
    skipped = float((occ == 0).float().mean())
    if X >= 512:
        assert skipped > 0.6  # most of the volume is not written at all
    if X * Y * Z <= 128 ** 3:
        dn = host(dists[3])
        want = np.zeros((Z, Y, X), np.uint32)
        O.tsdf_integrate(want, dn, np.asarray(voxel, np.float32), trunc, synth.MAX_WEIGHT, pose(3), fx, fy, cx, cy)
        assert np.array_equal(host(vol).view(np.uint32), want)
    # behind accumulating sweeps (the map only grows there) ...
    for f in (1, 2):
        A.tsdf_integrate(vol, dists[f], *args, pose(f), fx, fy, cx, cy, occupancy=occ)
        A.tsdf_integrate(ref, dists[f], *args, pose(f), fx, fy, cx, cy, occupancy=occ_ref)
    A.tsdf_clear_integrate(vol, dists[0], *args, pose(0), fx, fy, cx, cy, occupancy=occ, occupancy_known=True)
    A.tsdf_clear_integrate(ref, dists[0], *args, pose(0), fx, fy, cx, cy, occupancy=occ_ref)
    same()
    # ... and what the contract warns of: a map of zeros over a volume that is not
    vol.fill_(0x00070001)  # (weight 7: nothing one frame writes)
    occ.zero_()
    A.tsdf_clear_integrate(vol, dists[0], *args, pose(0), fx, fy, cx, cy, occupancy=occ, occupancy_known=True)
    left = vol == 0x00070001
    assert bool(left.any()) and not bool((left & (ref != 0)).any())  # only where the frame writes zeros
    A.tsdf_clear_integrate(vol, dists[0], *args, pose(0), fx, fy, cx, cy, occupancy=occ)  # the sweep without the promise repairs it
    same()
