"""-m gpu parity tests of the TSDF seam: HIP kernels (through the C ABI) vs the CPU oracle.

Bar: BIT-EXACT — voxel indices touched, 16-bit weights and half-float tsdf bits; raycast
outputs compared as raw float bits (misses are the 0x7fffffff NaN of the reference).
The TSDF oracle itself is unpinned (the reference has no TSDF tests), see oracle/oracle.h.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from dynfu_amd import synth  # noqa: E402
from gpu_util import aff12, bits, dev, host, rot  # noqa: E402


@pytest.fixture(scope="module")
def A():
    import dynfu_amd
    dynfu_amd.load()
    return dynfu_amd


def _scene(name, frame=0):
    cfg = synth.CONFIGS[name]
    fx, fy, cx, cy = synth.intrinsics(cfg)
    voxel, trunc, vol2cam, cam2vol, rinv = synth.volume_params(cfg)
    depth = synth.depth_frame(cfg, frame)
    return cfg, (fx, fy, cx, cy), voxel, trunc, vol2cam, cam2vol, rinv, depth


@pytest.mark.parametrize("shape", [(480, 640), (37, 53), (1, 1), (8, 130)])
def test_compute_dists_bit_exact(A, shape):
    import torch
    rng = np.random.default_rng(1)
    rows, cols = shape
    depth = rng.integers(0, 6000, (rows, cols)).astype(np.uint16)
    depth[rng.random(depth.shape) < 0.1] = 0
    # pitched input and output (PtrStep semantics): wider row stride than cols
    pitch = cols + 6
    d_in = torch.zeros((rows, pitch), dtype=torch.uint16, device="cuda")
    d_in[:, :cols] = dev(depth)
    d_out = torch.full((rows, pitch), 0x5555, dtype=torch.uint16, device="cuda")
    fx, fy, cx, cy = 525.0, 520.0, cols / 2 - 0.5, rows / 2 - 0.5
    A.compute_dists(d_in[:, :cols], d_out[:, :cols], fx, fy, cx, cy)
    got = host(d_out)
    assert np.array_equal(got[:, :cols], O.compute_dists(depth, fx, fy, cx, cy))
    assert np.all(got[:, cols:] == 0x5555)  # padding untouched


@pytest.mark.parametrize("dims", [(64, 64, 64), (50, 38, 44), (4, 4, 4), (128, 32, 72)])
def test_clear(A, dims):
    import torch
    X, Y, Z = dims
    vol = torch.full((Z, Y, X), 0x12345678, dtype=torch.int32, device="cuda")
    A.tsdf_clear(vol)
    assert int(vol.abs().max()) == 0


def _integrate_both(A, vol_np, dists, voxel, trunc, maxw, vol2cam, intr, fused=False):
    import torch
    v = dev(vol_np)
    d = dev(dists)
    (A.tsdf_clear_integrate if fused else A.tsdf_integrate)(v, d, voxel, trunc, maxw, vol2cam, *intr)
    torch.cuda.synchronize()
    ref = np.zeros_like(vol_np) if fused else vol_np.copy()
    n = O.tsdf_integrate(ref, dists, voxel, trunc, maxw, vol2cam, *intr, threads=8)
    return host(v, np.uint32), ref, n


@pytest.mark.parametrize("name", ["T0", "T1"])
def test_integrate_bit_exact_synthetic_frames(A, name):
    cfg, intr, voxel, trunc, vol2cam, _, _, _ = _scene(name)
    dim = cfg["dim"]
    vol = np.zeros((dim, dim, dim), np.uint32)
    for f in range(3):  # running average over three different frames, weight clamp at 2
        dists = O.compute_dists(synth.depth_frame(cfg, f * 7, noise_mm=1.0), *intr)
        got, ref, n = _integrate_both(A, vol, dists, voxel, trunc, 2, vol2cam, intr)
        assert n > 0.05 * vol.size
        assert np.array_equal(got, ref), "frame %d: %d voxels differ" % (f, int((got != ref).sum()))
        vol = ref
    assert set(np.unique(vol >> 16)) == {0, 1, 2}


@pytest.mark.parametrize("dims", [(50, 38, 44), (36, 20, 9), (64, 8, 130), (4, 4, 4)])
def test_integrate_bit_exact_ragged_dims_and_rotated_camera(A, dims):
    # X not a multiple of 4 (scalar path), tiny volumes, camera inside the volume looking
    # along a tilted axis (voxels behind the camera, rays leaving the image on all sides)
    X, Y, Z = dims
    rng = np.random.default_rng(X * 1000 + Y)
    rows, cols = 61, 83
    depth = (rng.uniform(400, 2500, (rows, cols))).astype(np.uint16)
    depth[rng.random(depth.shape) < 0.2] = 0
    intr = (70.0, 65.0, cols / 2 - 0.5, rows / 2 - 0.5)
    dists = O.compute_dists(depth, *intr)
    voxel = np.array([0.04, 0.05, 0.03], np.float32)
    R = rot([0.3, 1.0, 0.2], 0.4)
    # camera sits inside / just behind the volume whatever its size: part of the voxels have
    # vc.z <= 0, part project outside the image, part land on invalid (0) depth
    centre = 0.5 * voxel * np.array([X, Y, Z], np.float32)
    vol2cam = aff12(R, np.array([0.05, -0.03, 0.9], np.float32) - (R @ centre).astype(np.float32))
    vol = rng.integers(0, 2 ** 32, (Z, Y, X), dtype=np.uint64).astype(np.uint32)
    vol = (vol & 0x0003FFFF) | 0x3000  # plausible half tsdf + small weights
    got, ref, n = _integrate_both(A, vol, dists, voxel, 0.1, 5, vol2cam, intr)
    assert n > 0
    assert np.array_equal(got, ref)
    got, ref, _ = _integrate_both(A, vol, dists, voxel, 0.1, 5, vol2cam, intr, fused=True)
    assert np.array_equal(got, ref)


def test_fused_clear_integrate_equals_clear_then_integrate(A):
    cfg, intr, voxel, trunc, vol2cam, _, _, depth = _scene("T1")
    dim = cfg["dim"]
    dists = O.compute_dists(depth, *intr)
    junk = np.full((dim, dim, dim), 0xDEADBEEF, np.uint32)
    got, ref, _ = _integrate_both(A, junk, dists, voxel, trunc, 64, vol2cam, intr, fused=True)
    assert np.array_equal(got, ref)


def _odd_dists(cfg, intr, seed):
    """dists of a synthetic frame with patches of every kind of pixel the reference skips or saturates on:
    +0, -0, negative, NaN, +inf, the smallest subnormal"""
    rng = np.random.default_rng(seed)
    dists = O.compute_dists(synth.depth_frame(cfg, 3, noise_mm=1.0), *intr)
    H, W = dists.shape
    vals = np.array([0x0000, 0x8000, 0xC000, 0x7E00, 0x7C00, 0x0001, 0x8001, 0xA11F, 0xFC00], np.uint16)  # .. -tiny, -0.01, -inf
    for _ in range(300):
        x0, y0 = rng.integers(0, W), rng.integers(0, H)
        w, h = rng.integers(1, 14), rng.integers(1, 14)
        dists[y0:y0 + h, x0:x0 + w] = vals[rng.integers(0, len(vals))]
    return dists


@pytest.mark.parametrize("tilt", [0.0, 0.35])
def test_integrate_bit_exact_with_holes_and_special_halves(A, tilt):
    # the run classification (csrc/tsdf_classify.hpp) bounds the dists a run can meet by 8x8 tiles: invalid pixels,
    # NaN and infinities must come out exactly as the per-voxel reference treats them
    cfg, intr, voxel, trunc, vol2cam, _, _, _ = _scene("T1")
    dim = cfg["dim"]
    dists = _odd_dists(cfg, intr, 5)
    if tilt:
        R = rot([0.2, 1.0, -0.3], tilt)
        centre = 0.5 * voxel * dim
        vol2cam = aff12(R, (vol2cam[9:] + centre) - (R @ centre).astype(np.float32))
    vol = np.zeros((dim, dim, dim), np.uint32)
    got, ref, n = _integrate_both(A, vol, dists, voxel, trunc, 64, vol2cam, intr, fused=True)
    assert n > 0.03 * vol.size
    assert np.array_equal(got, ref), "%d voxels differ" % int((got != ref).sum())
    for _ in range(2):
        got, ref, _ = _integrate_both(A, ref, dists, voxel, trunc, 2, vol2cam, intr)
        assert np.array_equal(got, ref), "%d voxels differ" % int((got != ref).sum())


def test_integrate_close_up_and_camera_plane_through_the_volume(A):
    # several pixels per voxel (runs spanning many tiles) and runs that cross z = 0
    cfg, intr, voxel, trunc, _, _, _, depth = _scene("T1")
    dists = O.compute_dists(depth, *intr)
    for size, t in ((0.5, (-0.25, -0.25, 0.02)), (3.0, (-1.5, -1.5, -1.0))):
        dim = 64
        vx = np.full(3, size / dim, np.float32)
        R = rot([1.0, 0.5, 0.1], 0.25)
        v2c = aff12(R, np.array(t, np.float32))
        vol = np.zeros((dim, dim, dim), np.uint32)
        got, ref, _ = _integrate_both(A, vol, dists, vx, max(0.04, 2.1 * size / dim), 64, v2c, intr, fused=True)
        assert np.array_equal(got, ref)
        got, ref, _ = _integrate_both(A, ref, dists, vx, max(0.04, 2.1 * size / dim), 64, v2c, intr)
        assert np.array_equal(got, ref)


def test_integrate_on_two_streams_with_different_frames(A):
    # the tile table of a sweep is internal scratch, cached per stream: two sweeps in flight must not share it
    import torch
    cfg, intr, voxel, trunc, vol2cam, _, _, _ = _scene("T1")
    dim = cfg["dim"]
    d = [O.compute_dists(synth.depth_frame(cfg, f), *intr) for f in (0, 25)]
    d[1][:, : d[1].shape[1] // 2] = 0
    refs = []
    for k in range(2):
        r = np.zeros((dim, dim, dim), np.uint32)
        O.tsdf_integrate(r, d[k], voxel, trunc, 64, vol2cam, *intr, threads=8)
        refs.append(r)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    vols = [torch.empty((dim, dim, dim), dtype=torch.int32, device="cuda") for _ in range(2)]
    dd = [dev(x) for x in d]
    torch.cuda.synchronize()
    for rep in range(8):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                A.tsdf_clear_integrate(vols[k], dd[k], voxel, trunc, 64, vol2cam, *intr)
    torch.cuda.synchronize()
    for k in range(2):
        assert np.array_equal(host(vols[k], np.uint32), refs[k])


def test_run_classified_sweep_equals_per_voxel_sweep_at_512(A):
    """The default sweep against the per-voxel kernel of round 1 (DFA_TSDF_LEGACY=1 in a child process), C2 volume,
    tilted camera: same volume, bit for bit, fused and accumulating."""
    import subprocess, sys, torch
    cfg, intr, voxel, trunc, vol2cam, _, _, depth = _scene("C2")
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import dynfu_amd as A, oracle as O
from dynfu_amd import synth
from gpu_util import aff12, rot, dev
cfg = synth.CONFIGS["C2"]; intr = synth.intrinsics(cfg)
voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
dim = cfg["dim"]
R = rot([0.1, 1.0, 0.2], 0.3); centre = 0.5 * voxel * dim
v2c = aff12(R, (vol2cam[9:] + centre) - (R @ centre).astype(np.float32))
dists = dev(O.compute_dists(synth.depth_frame(cfg, 0), *intr))
v = torch.empty((dim, dim, dim), dtype=torch.int32, device="cuda")
A.tsdf_clear_integrate(v, dists, voxel, trunc, 64, v2c, *intr)
A.tsdf_integrate(v, dists, voxel, trunc, 64, v2c, *intr)
s = v.view(dim, -1).long()
print("CHK", int(s.sum()), int((s * torch.arange(1, s.shape[1] + 1, device="cuda")).sum() %% (2**61 - 1)), int(((v >> 16) == 2).sum()))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for legacy in (False, True):
        env = dict(os.environ)
        env.pop("DFA_TSDF_LEGACY", None)
        if legacy:
            env["DFA_TSDF_LEGACY"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("CHK")][0])
    assert outs[0] == outs[1], outs
    assert int(outs[0].split()[3]) > 0.02 * cfg["dim"] ** 3


@pytest.mark.parametrize("seed", range(12))
def test_integrate_random_small_configurations(A, seed):
    """Randomised sweep over what the run classification depends on: volume dims (1 .. 70 per axis, tails shorter than a
    run, partial waves), anisotropic voxels, image sizes from 1x1 to partial tiles, focal lengths (a run spanning from a
    fraction of a pixel to many tiles), poses (camera in front of, inside and behind the volume; any rotation),
    truncation distances, weights, and depth images with invalid / special pixels.  Fused and accumulating sweeps
    against the oracle, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    X, Y, Z = (int(v) for v in rng.integers(1, 71, 3))
    rows, cols = (int(v) for v in rng.choice([1, 2, 7, 8, 9, 31, 64, 97], 2))
    focal = float(rng.choice([3.0, 20.0, 80.0, 400.0]))
    intr = (focal, focal * float(rng.uniform(0.8, 1.2)), cols / 2 - 0.5 + float(rng.uniform(-3, 3)), rows / 2 - 0.5)
    voxel = rng.uniform(0.01, 0.08, 3).astype(np.float32)
    trunc = float(rng.uniform(0.03, 0.4))
    maxw = int(rng.choice([0, 1, 2, 64, 65535]))
    depth = rng.uniform(300, 4000, (rows, cols)).astype(np.uint16)
    depth[rng.random(depth.shape) < 0.25] = 0
    dists = O.compute_dists(depth, *intr)
    specials = np.array([0x8000, 0xC000, 0x7E00, 0x7C00, 0x0001, 0x03FF, 0x8001, 0xA11F, 0xFC00], np.uint16)
    mask = rng.random(dists.shape) < 0.08
    dists[mask] = specials[rng.integers(0, len(specials), int(mask.sum()))]
    R = rot(rng.normal(size=3), float(rng.uniform(0, np.pi)))
    centre = 0.5 * voxel * np.array([X, Y, Z], np.float32)
    where = rng.choice(["front", "inside", "behind", "far"])
    offset = {"front": [0.1, -0.1, 1.5], "inside": [0.0, 0.05, 0.1], "behind": [0.0, 0.0, -2.0], "far": [9.0, -7.0, 30.0]}[where]
    vol2cam = aff12(R, np.array(offset, np.float32) - (R @ centre).astype(np.float32))
    vol = np.zeros((Z, Y, X), np.uint32)
    got, ref, _ = _integrate_both(A, vol, dists, voxel, trunc, maxw, vol2cam, intr, fused=True)
    assert np.array_equal(got, ref), (seed, where, int((got != ref).sum()))
    junk = rng.integers(0, 2 ** 32, (Z, Y, X), dtype=np.uint64).astype(np.uint32)
    junk = (junk & 0x003FFFFF) | 0x3000
    for _ in range(2):
        got, ref, _ = _integrate_both(A, junk, dists, voxel, trunc, maxw, vol2cam, intr)
        assert np.array_equal(got, ref), (seed, where, int((got != ref).sum()))
        junk = ref


@pytest.mark.parametrize("zchunk", ["4", "7", "20", "1000"])
def test_integrate_independent_of_z_chunking(A, devlib, zchunk, monkeypatch):
    # the kernel replays the running `vc += zstep` additions for chunks that start at z0 > 0
    cfg, intr, voxel, trunc, vol2cam, _, _, depth = _scene("T0")
    dim = cfg["dim"]
    dists = O.compute_dists(depth, *intr)
    R = rot([1, 0.2, 0], 0.15)
    vol2cam = aff12(R, vol2cam[9:])
    monkeypatch.setenv("DFA_TSDF_ZCHUNK", zchunk)
    vol = np.zeros((dim, dim, dim), np.uint32)
    got, ref, _ = _integrate_both(A, vol, dists, voxel, trunc, 64, vol2cam, intr)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("name", ["T0", "T1"])
def test_raycast_bit_exact(A, name):
    import torch
    cfg, intr, voxel, trunc, vol2cam, cam2vol, rinv, depth = _scene(name)
    dim, W, H = cfg["dim"], cfg["width"], cfg["height"]
    vol = np.zeros((dim, dim, dim), np.uint32)
    dists = O.compute_dists(depth, *intr)
    for _ in range(2):
        O.tsdf_integrate(vol, dists, voxel, trunc, 64, vol2cam, *intr, threads=8)
    v = dev(vol)
    # slightly moved camera so rays are not axis aligned with the integration frame
    R = rot([0, 1, 0], 0.05)
    cam2vol_m = aff12(R, cam2vol[9:] + np.array([0.02, -0.01, 0.0], np.float32))
    rinv_m = np.linalg.inv(R).astype(np.float32).reshape(-1)
    for c2v, ri in ((cam2vol, rinv), (cam2vol_m, rinv_m)):
        pts = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        nrm = torch.zeros_like(pts)
        A.tsdf_raycast_points(v, voxel, trunc, c2v, ri, *intr, synth.RAYCAST_STEP_FACTOR,
                              synth.GRADIENT_DELTA_FACTOR, pts, nrm)
        rp, rn = O.tsdf_raycast_points(vol, voxel, trunc, c2v, ri, *intr, synth.RAYCAST_STEP_FACTOR,
                                       synth.GRADIENT_DELTA_FACTOR, W, H, threads=8)
        hits = ~np.isnan(rp[..., 0])
        assert hits.mean() > 0.5
        assert np.array_equal(bits(host(pts)), bits(rp))
        assert np.array_equal(bits(host(nrm)), bits(rn))
        dep = torch.full((H, W), 7, dtype=torch.uint16, device="cuda")
        nrm2 = torch.zeros_like(pts)
        A.tsdf_raycast_depth(v, voxel, trunc, c2v, ri, *intr, synth.RAYCAST_STEP_FACTOR,
                             synth.GRADIENT_DELTA_FACTOR, dep, nrm2)
        rd, rn2 = O.tsdf_raycast_depth(vol, voxel, trunc, c2v, ri, *intr, synth.RAYCAST_STEP_FACTOR,
                                       synth.GRADIENT_DELTA_FACTOR, W, H, threads=8)
        assert np.array_equal(host(dep), rd)
        assert np.array_equal(bits(host(nrm2)), bits(rn2))


def _raycast_both_variants_bit_exact(A, v, vol_host, voxel, trunc, c2v, ri, intr, W, H, min_hits):
    import torch
    pts = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    nrm = torch.zeros_like(pts)
    A.tsdf_raycast_points(v, voxel, trunc, c2v, ri, *intr, synth.RAYCAST_STEP_FACTOR, synth.GRADIENT_DELTA_FACTOR, pts, nrm)
    rp, rn = O.tsdf_raycast_points(vol_host, voxel, trunc, c2v, ri, *intr, synth.RAYCAST_STEP_FACTOR,
                                   synth.GRADIENT_DELTA_FACTOR, W, H, threads=8)
    hits = ~np.isnan(rp[..., 0])
    assert hits.mean() > min_hits, hits.mean()
    assert np.array_equal(bits(host(pts)), bits(rp))
    assert np.array_equal(bits(host(nrm)), bits(rn))
    dep = torch.full((H, W), 7, dtype=torch.uint16, device="cuda")
    nrm2 = torch.zeros_like(pts)
    A.tsdf_raycast_depth(v, voxel, trunc, c2v, ri, *intr, synth.RAYCAST_STEP_FACTOR, synth.GRADIENT_DELTA_FACTOR, dep, nrm2)
    rd, rn2 = O.tsdf_raycast_depth(vol_host, voxel, trunc, c2v, ri, *intr, synth.RAYCAST_STEP_FACTOR,
                                   synth.GRADIENT_DELTA_FACTOR, W, H, threads=8)
    assert np.array_equal(host(dep), rd)
    assert np.array_equal(bits(host(nrm2)), bits(rn2))
    return int(hits.sum())


@pytest.mark.parametrize("name", ["C1", "C2", "C4"])
def test_raycast_bit_exact_at_baseline_sizes(A, name):
    """Both raycast variants at the BASELINE configurations' own sizes — 256^3 / VGA, 512^3 / VGA, 1024^3 / 720p (a
    4 GiB volume: voxel offsets beyond 2^31 bytes, 2-4x the march steps of the small cases, `tmax -= step` and the
    NaN edges of the trilinear stencil in other places) — against oracle/tsdf_oracle.c, bit for bit, from the
    integration pose and from a moved, rotated camera.  The volume is the HIP sweep's (two sweeps: weights 2, averaged
    values), downloaded once; the oracle casts its rays through that very copy."""
    import torch
    cfg, intr, voxel, trunc, vol2cam, cam2vol, rinv, depth = _scene(name)
    dim, W, H = cfg["dim"], cfg["width"], cfg["height"]
    dists = torch.empty((H, W), dtype=torch.uint16, device="cuda")
    A.compute_dists(dev(depth), dists, *intr)
    v = torch.empty((dim, dim, dim), dtype=torch.int32, device="cuda")
    A.tsdf_clear_integrate(v, dists, voxel, trunc, 64, vol2cam, *intr)
    depth2 = _scene(name, frame=12)[7]
    A.compute_dists(dev(depth2), dists, *intr)
    A.tsdf_integrate(v, dists, voxel, trunc, 64, vol2cam, *intr)
    vol_host = host(v, np.uint32)
    R = rot([0.1, 1, 0.05], 0.07)
    cam2vol_m = aff12(R, cam2vol[9:] + np.array([0.03, -0.02, 0.01], np.float32))
    rinv_m = np.linalg.inv(R).astype(np.float32).reshape(-1)
    for c2v, ri in ((cam2vol, rinv), (cam2vol_m, rinv_m)):
        nhits = _raycast_both_variants_bit_exact(A, v, vol_host, voxel, trunc, c2v, ri, intr, W, H, 0.5)
        # the work counters of the measurement entry point describe the same rays
        t = A.tsdf_raycast_tally(v, voxel, trunc, c2v, ri, *intr, synth.RAYCAST_STEP_FACTOR, synth.GRADIENT_DELTA_FACTOR, W, H,
                                 unique=(name != "C4"))
        assert t["hits"] == nhits and t["rays_entered"] <= W * H and t["march_fetches"] >= 2 * t["hits"]
        assert t["trilinear_fetches"] >= 64 * t["hits"] - 8 * 8 * 64  # 8 samples of 8 voxels per hit (edge samples fall out)
        if t["unique_voxels"] is not None:
            assert t["hits"] < t["unique_voxels"] <= t["march_fetches"] + t["trilinear_fetches"]


def test_raycast_through_a_volume_with_far_side_surfaces(A):
    """rays that leave a surface again (tsdf - -> +: the reference's early exit, tsdf_volume.cu:234) and rays that graze:
    a 256^3 volume fused from TWO poses so that back faces exist, cast from a third pose"""
    import torch
    cfg, intr, voxel, trunc, vol2cam, cam2vol, rinv, depth = _scene("C1")
    dim, W, H = cfg["dim"], cfg["width"], cfg["height"]
    dists = torch.empty((H, W), dtype=torch.uint16, device="cuda")
    A.compute_dists(dev(depth), dists, *intr)
    v = torch.empty((dim, dim, dim), dtype=torch.int32, device="cuda")
    A.tsdf_clear_integrate(v, dists, voxel, trunc, 64, vol2cam, *intr)
    R2 = rot([0, 1, 0], 0.6)
    centre = np.array([0.0, 0.0, 1.5])
    t2 = (centre - R2 @ centre) + (R2 @ np.array(vol2cam[9:], np.float64))
    A.tsdf_integrate(v, dists, voxel, trunc, 64, aff12(R2, t2), *intr)
    vol_host = host(v, np.uint32)
    R3 = rot([0.2, 1, 0], -0.8)
    cam_pos_vol = np.array([1.5, 1.5, 1.0]) + R3 @ np.array([0.0, 0.0, -1.4])
    c2v = aff12(R3, cam_pos_vol)
    ri = np.linalg.inv(R3).astype(np.float32).reshape(-1)
    _raycast_both_variants_bit_exact(A, v, vol_host, voxel, trunc, c2v, ri, intr, W, H, 0.05)


def test_full_size_sweep_256_bit_exact(A):
    """BASELINE config C1's volume (256^3) against the oracle, whole: two frames fused, bit for bit"""
    import torch
    cfg, intr, voxel, trunc, vol2cam, _, _, depth = _scene("C1")
    dim = cfg["dim"]
    ref = np.zeros((dim, dim, dim), np.uint32)
    v = torch.full((dim, dim, dim), -1, dtype=torch.int32, device="cuda")
    for i, frame in enumerate((0, 9)):
        d = O.compute_dists(_scene("C1", frame=frame)[7], *intr)
        (A.tsdf_clear_integrate if i == 0 else A.tsdf_integrate)(v, dev(d), voxel, trunc, 64, vol2cam, *intr)
        O.tsdf_integrate(ref, d, voxel, trunc, 64, vol2cam, *intr, threads=8)
    assert np.array_equal(host(v, np.uint32), ref)
    assert 0.05 < float((ref >> 16 == 2).mean()) < 0.9


@pytest.mark.parametrize("size", [(150, 101), (17, 9), (1040, 24)])
def test_raycast_ragged_image_sizes(A, size):
    """image sizes that are not multiples of the 16 x 16-pixel workgroup tile nor of the 64-tile round of the XCD-aware
    tile order (partial tiles at the right / bottom edge, a partial last round, a single row of tiles)"""
    cfg, intr, voxel, trunc, vol2cam, cam2vol, rinv, depth = _scene("T0")
    dim = cfg["dim"]
    vol = np.zeros((dim, dim, dim), np.uint32)
    O.tsdf_integrate(vol, O.compute_dists(depth, *intr), voxel, trunc, 64, vol2cam, *intr, threads=8)
    W, H = size
    f = 0.82 * W
    intr2 = (f, f * 1.01, W / 2 - 0.5, H / 2 - 0.5)
    R = rot([0.3, 1, 0], 0.04)
    c2v = aff12(R, cam2vol[9:] + np.array([0.01, 0.02, -0.01], np.float32))
    ri = np.linalg.inv(R).astype(np.float32).reshape(-1)
    _raycast_both_variants_bit_exact(A, dev(vol), vol, voxel, trunc, c2v, ri, intr2, W, H, 0.02)


def test_raycast_with_64_bit_voxel_indices(A, devlib, monkeypatch):
    """volumes beyond 2^32 voxels (16 GiB) take the kernels' 64-bit voxel index; the development flavour runs that
    instantiation on a small volume (DFA_RAY_IDX64=1): same bits as the oracle"""
    monkeypatch.setenv("DFA_RAY_IDX64", "1")
    cfg, intr, voxel, trunc, vol2cam, cam2vol, rinv, depth = _scene("T1")
    dim, W, H = cfg["dim"], cfg["width"], cfg["height"]
    vol = np.zeros((dim, dim, dim), np.uint32)
    O.tsdf_integrate(vol, O.compute_dists(depth, *intr), voxel, trunc, 64, vol2cam, *intr, threads=8)
    R = rot([0, 1, 0.1], 0.05)
    c2v = aff12(R, cam2vol[9:] + np.array([0.02, -0.01, 0.0], np.float32))
    ri = np.linalg.inv(R).astype(np.float32).reshape(-1)
    _raycast_both_variants_bit_exact(A, dev(vol), vol, voxel, trunc, c2v, ri, intr, W, H, 0.5)


@pytest.mark.parametrize("dims,vox", [((50, 38, 44), (0.05, 0.06, 0.055)), ((96, 20, 33), (0.03, 0.12, 0.08))])
def test_raycast_non_cubic_volume_and_anisotropic_voxels(A, dims, vox):
    """the raycaster's per-axis voxel sizes, volume extents (`box_max = size - voxel` per axis) and gradient deltas on a
    volume that is neither cubic nor isotropic, camera off-axis"""
    X, Y, Z = dims
    voxel = np.array(vox, np.float32)
    trunc = float(max(0.04, 2.1 * voxel.max()))
    W, H = 160, 120
    intr = (131.25, 131.25, W / 2 - 0.5, H / 2 - 0.5)
    size = voxel * np.array(dims, np.float32)
    vol2cam = aff12(np.eye(3), np.array([-size[0] / 2, -size[1] / 2, 0.6], np.float32))
    depth = synth.depth_frame(synth.CONFIGS["T0"], 3)
    vol = np.zeros((Z, Y, X), np.uint32)
    O.tsdf_integrate(vol, O.compute_dists(depth, *intr), voxel, trunc, 64, vol2cam, *intr, threads=8)
    R = rot([0.2, 1, 0.1], 0.09)
    c2v = aff12(R, -vol2cam[9:] + np.array([0.04, -0.03, 0.02], np.float32))
    ri = np.linalg.inv(R).astype(np.float32).reshape(-1)
    _raycast_both_variants_bit_exact(A, dev(vol), vol, voxel, trunc, c2v, ri, intr, W, H, 0.05)


def test_raycast_miss_everywhere_on_empty_volume(A):
    import torch
    cfg, intr, voxel, trunc, _, cam2vol, rinv, _ = _scene("T0")
    dim, W, H = cfg["dim"], cfg["width"], cfg["height"]
    v = torch.zeros((dim, dim, dim), dtype=torch.int32, device="cuda")
    pts = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    nrm = torch.zeros_like(pts)
    A.tsdf_raycast_points(v, voxel, trunc, cam2vol, rinv, *intr, 0.75, 0.5, pts, nrm)
    assert np.all(bits(host(pts)) == 0x7FFFFFFF) and np.all(bits(host(nrm)) == 0x7FFFFFFF)


def test_full_size_properties_512(A):
    """BASELINE config C2 volume (512^3): size-independent properties instead of the oracle
    (which needs ~10 s per sweep): fused == clear+integrate; W sweeps of one frame give weight
    min(W, max_weight) on exactly the voxels of the first sweep with the tsdf unchanged
    (running average of identical samples); checksum independent of the z chunking."""
    import torch
    cfg, intr, voxel, trunc, vol2cam, _, _, depth = _scene("C2")
    dim = cfg["dim"]
    dists = dev(O.compute_dists(depth, *intr))
    a = torch.full((dim, dim, dim), -1, dtype=torch.int32, device="cuda")
    A.tsdf_clear_integrate(a, dists, voxel, trunc, 3, vol2cam, *intr)
    b = torch.full((dim, dim, dim), -1, dtype=torch.int32, device="cuda")
    A.tsdf_clear(b)
    A.tsdf_integrate(b, dists, voxel, trunc, 3, vol2cam, *intr)
    assert torch.equal(a, b)
    touched = (a >> 16) == 1
    frac = float(touched.float().mean())
    assert 0.05 < frac < 0.9
    assert int((a[~touched]).abs().max()) == 0
    for sweep in range(2, 6):
        A.tsdf_integrate(b, dists, voxel, trunc, 3, vol2cam, *intr)
        w = b >> 16
        assert torch.equal(w == min(sweep, 3), touched) and int(w[~touched].max()) == 0
        # averaging identical samples: (F*W + F)/(W+1) stays within one half ulp step of F
        diff = ((b & 0xFFFF) - (a & 0xFFFF)).abs()
        assert int(diff.max()) <= 1
    os.environ["DFA_TSDF_ZCHUNK"] = "48"  # (a switch of the development flavour: the product picks its chunks itself)
    try:
        from dynfu_amd import _lib
        with _lib.use_library(_lib.dev_lib_path()):
            c = torch.empty_like(a)
            A.tsdf_clear_integrate(c, dists, voxel, trunc, 3, vol2cam, *intr)
    finally:
        del os.environ["DFA_TSDF_ZCHUNK"]
    assert torch.equal(a, c)
    # spot-check one z-slab of the 512^3 result against the oracle (bit exact)
    ref = np.zeros((dim, dim, dim), np.uint32)
    O.tsdf_integrate(ref, host(dists), voxel, trunc, 3, vol2cam, *intr, threads=8)
    assert np.array_equal(host(a, np.uint32), ref)


def test_full_size_properties_1024(A):
    """BASELINE config C4 (1024^3, 4 GiB volume, 720p depth): fused == clear + integrate, a second
    sweep doubles the weights of exactly the touched voxels, and the per-z-slab XOR/sum checksum of
    the volume is independent of the z chunking."""
    import torch
    cfg, intr, voxel, trunc, vol2cam, _, _, depth = _scene("C4")
    dim = cfg["dim"]
    dists = torch.empty((cfg["height"], cfg["width"]), dtype=torch.uint16, device="cuda")
    A.compute_dists(dev(depth), dists, *intr)
    a = torch.empty((dim, dim, dim), dtype=torch.int32, device="cuda")
    A.tsdf_clear_integrate(a, dists, voxel, trunc, 64, vol2cam, *intr)
    b = torch.full((dim, dim, dim), 0x7F7F7F7F, dtype=torch.int32, device="cuda")
    A.tsdf_clear(b)
    A.tsdf_integrate(b, dists, voxel, trunc, 64, vol2cam, *intr)
    assert torch.equal(a, b)
    checksum = lambda v: (v.view(dim, -1).long().sum(1))
    ca = checksum(a)
    touched = int(((a >> 16) == 1).sum())
    assert 0.02 * dim ** 3 < touched < 0.9 * dim ** 3
    del b
    A.tsdf_integrate(a, dists, voxel, trunc, 64, vol2cam, *intr)
    assert int(((a >> 16) == 2).sum()) == touched and int(((a >> 16) == 1).sum()) == 0
    os.environ["DFA_TSDF_ZCHUNK"] = "200"
    try:
        from dynfu_amd import _lib
        with _lib.use_library(_lib.dev_lib_path()):
            c = torch.empty_like(a)
            A.tsdf_clear_integrate(c, dists, voxel, trunc, 64, vol2cam, *intr)
    finally:
        del os.environ["DFA_TSDF_ZCHUNK"]
    assert torch.equal(checksum(c), ca)
    del c
    # ---- bit for bit against the oracle, slab by slab (orc_tsdf_integrate_slab replays the z additions below the slab):
    # the front of the volume, the slices through the sphere's surface, the slices around the background plane and the
    # last slices — two sweeps each, as `a` has had
    dists_h = host(dists)
    vs = float(voxel[2])
    z_sphere = int(round((1.0 - float(vol2cam[11])) / vs))   # the sphere's front, z = 1.0 m
    z_plane = int(round((synth.PLANE_Z - float(vol2cam[11])) / vs))
    for z0 in (0, z_sphere - 16, z_plane - 16, dim - 32):
        slab = np.zeros((32, dim, dim), np.uint32)
        for _ in range(2):
            O.tsdf_integrate_slab(slab, z0, dists_h, voxel, trunc, 64, vol2cam, *intr, threads=8)
        assert np.array_equal(host(a[z0:z0 + 32], np.uint32), slab), z0
        if z0 in (z_sphere - 16, z_plane - 16):
            assert (slab >> 16 == 2).mean() > 0.01  # the slab is not an empty one


def test_argument_errors_are_loud(A):
    import torch
    v = torch.zeros((8, 8, 8), dtype=torch.int32, device="cuda")
    d = torch.zeros((4, 4), dtype=torch.uint16, device="cuda")
    with pytest.raises(A.DynfuAmdError):
        A.tsdf_integrate(v, d, [0.1, 0.1, 0.1], -1.0, 64, np.eye(4)[:3], 1, 1, 0, 0)  # trunc <= 0
    with pytest.raises(A.DynfuAmdError):
        A.tsdf_integrate(v, d, [0.1, 0.1, 0.1], 0.1, 70000, np.eye(4)[:3], 1, 1, 0, 0)  # weight overflow
    with pytest.raises(A.DynfuAmdError):
        A.tsdf_integrate(v.cpu(), d, [0.1, 0.1, 0.1], 0.1, 64, np.eye(4)[:3], 1, 1, 0, 0)  # host pointer
