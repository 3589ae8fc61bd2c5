"""Deterministic synthetic inputs for the BASELINE.json configurations (numpy, host side).

Harness data only — used by bench.py, __graft_entry__.smoke() and the parity tests; the
kernels never see this module.  Everything follows SURVEY.md §8(d), with the reference's
constants cited where they come from:
  camera    640x480, fx = fy = 525, cx = 319.5, cy = 239.5       (src/kfusion/kinfu.cpp:16-18)
  volume    3 m cube, pose translate(-1.5, -1.5, 0.5), trunc 0.04 m, max weight 64,
            raycast step factor 0.75, gradient delta factor 0.5  (src/kfusion/kinfu.cpp:20-38)
  nodes     every 128th canonical vertex, dg_w = 3 * epsilon    (src/dynfu/dyn_fusion.cpp:151-161)
  solver    tukeyOffset 4.652, psi_data 0.01, lambda 200, psi_reg 1e-4 (src/dynfu/dyn_fusion.cpp:13-20)
"""
import math

import numpy as np

CONFIGS = {
    # name: volume dim, image (w, h), focal, nodes D, k, GN ("outer") iterations
    "C1": dict(dim=256, width=640, height=480, focal=525.0, D=512, k=8, gn_iters=5),
    "C2": dict(dim=512, width=640, height=480, focal=525.0, D=2048, k=4, gn_iters=5),
    "C3": dict(dim=512, width=640, height=480, focal=525.0, D=4096, k=8, gn_iters=10),
    "C4": dict(dim=1024, width=1280, height=720, focal=1050.0, D=8192, k=8, gn_iters=10),
    # small variants for tests / smoke (same geometry, fewer voxels / vertices)
    "T0": dict(dim=64, width=160, height=120, focal=131.25, D=64, k=4, gn_iters=3),
    "T1": dict(dim=128, width=320, height=240, focal=262.5, D=256, k=8, gn_iters=3),
}

VOLUME_SIZE = 3.0
VOLUME_POSE_T = (-1.5, -1.5, 0.5)
TRUNC_DIST = 0.04
MAX_WEIGHT = 64
RAYCAST_STEP_FACTOR = 0.75
GRADIENT_DELTA_FACTOR = 0.5
EPSILON = 0.025
VERTS_PER_NODE = 128
SPHERE_C = np.array([0.0, 0.0, 1.5])
SPHERE_R = 0.5
PLANE_Z = 2.5
N_FRAMES = 50

SOLVER = dict(tukey_offset=4.652, psi_data=0.01, lambda_=200.0, psi_reg=1e-4)


def intrinsics(cfg):
    return cfg["focal"], cfg["focal"], cfg["width"] / 2 - 0.5, cfg["height"] / 2 - 0.5


def volume_params(cfg):
    """voxel size (3,), clamped trunc distance (tsdf_volume.cpp:57-61), vol2cam as 12 floats."""
    dim = cfg["dim"]
    vs = np.float32(VOLUME_SIZE) / np.float32(dim)
    voxel = np.array([vs, vs, vs], np.float32)
    trunc = max(np.float32(TRUNC_DIST), np.float32(2.1) * vs)
    # camera pose = identity  =>  vol2cam = pose, cam2vol = pose^-1
    vol2cam = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, *VOLUME_POSE_T], np.float32)
    cam2vol = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, *(-np.array(VOLUME_POSE_T))], np.float32)
    rinv = np.eye(3, dtype=np.float32).reshape(-1)
    return voxel, float(trunc), vol2cam, cam2vol, rinv


def _radius(dirs, frame):
    """bulged sphere radius for unit directions from the sphere centre"""
    theta = np.arctan2(dirs[..., 1], dirs[..., 0])
    phi = np.arcsin(np.clip(dirs[..., 2], -1, 1))
    return SPHERE_R + 0.01 * np.sin(3 * theta + 2 * math.pi * frame / N_FRAMES) * np.cos(2 * phi)


_DEPTH_CACHE = {}


def depth_frame(cfg, frame, noise_mm=0.0):
    """u16 depth in millimetres: bulged sphere in front of the plane z = 2.5 m, 5 px invalid border (memoised: a
    frame costs 0.3-1.4 s of numpy and bench.py builds several sequences of the same configuration)."""
    key = (cfg["width"], cfg["height"], cfg["focal"], frame, noise_mm)
    if key not in _DEPTH_CACHE:
        _DEPTH_CACHE[key] = _depth_frame(cfg, frame, noise_mm)
        _DEPTH_CACHE[key].setflags(write=False)
    return _DEPTH_CACHE[key]


def _depth_frame(cfg, frame, noise_mm):
    fx, fy, cx, cy = intrinsics(cfg)
    W, H = cfg["width"], cfg["height"]
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    d = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    # ray / sphere with a direction dependent radius: fixed-point on the radius (bulge is 2 %)
    r = np.full(u.shape, SPHERE_R)
    hit = np.zeros(u.shape, bool)
    s = np.zeros(u.shape)
    for _ in range(6):
        b = d @ SPHERE_C
        disc = b * b - (SPHERE_C @ SPHERE_C - r * r)
        hit = disc > 0
        s = np.where(hit, b - np.sqrt(np.where(hit, disc, 0.0)), 0.0)
        p = d * s[..., None] - SPHERE_C
        n = np.linalg.norm(p, axis=-1, keepdims=True)
        r = np.where(hit, _radius(p / np.where(n > 0, n, 1.0), frame), r)
    z_sphere = s * d[..., 2]
    z = np.where(hit, z_sphere, PLANE_Z)
    mm = np.rint(z * 1000.0)
    if noise_mm > 0:
        rng = np.random.default_rng(frame + 1)
        mm = mm + np.rint(rng.normal(0.0, noise_mm, mm.shape))
    depth = np.clip(mm, 0, 65535).astype(np.uint16)
    depth[:5, :] = 0
    depth[-5:, :] = 0
    depth[:, :5] = 0
    depth[:, -5:] = 0
    return depth


def canonical(cfg):
    """N = 128 D vertices on the camera-facing hemisphere of the frame-0 surface (Fibonacci
    spiral, fixed order), analytic normals, nodes = every 128th vertex."""
    D = cfg["D"]
    N = VERTS_PER_NODE * D
    i = np.arange(N, dtype=np.float64)
    golden = math.pi * (3.0 - math.sqrt(5.0))
    zc = -(i + 0.5) / N  # camera looks down +z: facing hemisphere has direction z < 0
    rad = np.sqrt(1.0 - zc * zc)
    ang = golden * i
    dirs = np.stack([rad * np.cos(ang), rad * np.sin(ang), zc], -1)
    r = _radius(dirs, 0)
    verts = (SPHERE_C + dirs * r[:, None]).astype(np.float32)
    normals = dirs.astype(np.float32)
    node_pos = np.ascontiguousarray(verts[::VERTS_PER_NODE])
    node_w = np.full(D, 3 * EPSILON, np.float32)
    node_dq = np.zeros((D, 8), np.float32)
    node_dq[:, 0] = 1.0
    return dict(verts=verts, normals=normals, node_pos=node_pos, node_w=node_w, node_dq=node_dq)


def true_translations(node_pos, frame, k=4):
    """ground-truth node translations of frame `frame` (metres).  The reference model sums the
    UN-normalised RBF weights of the k neighbours (energy.t:50-53), so a vertex moves by about
    (sum of weights ~ 0.93 k) x |t|: the amplitude is 1 cm at k = 4 and scaled by 4/k above, which
    keeps the displacement inside the Tukey cut-off (4.652 x 0.01 m) for most vertices."""
    ph = 2 * math.pi * frame / N_FRAMES
    p = node_pos.astype(np.float64)
    amp = 0.01 * min(1.0, 4.0 / k)
    t = amp * np.stack([np.sin(7 * p[:, 0] + ph), np.cos(5 * p[:, 1] + ph), np.sin(3 * p[:, 2] + ph)], -1)
    return t.astype(np.float32)


def live_vertices(verts, idx, weights, t_true):
    """reference-parity model: live = canon + sum_j w_j t*_j (energy.t:50-55)"""
    idx = np.asarray(idx)
    w = np.where(idx >= 0, np.asarray(weights, np.float64), 0.0)
    contrib = (w[..., None] * t_true.astype(np.float64)[np.maximum(idx, 0)]).sum(1)
    return (verts.astype(np.float64) + contrib).astype(np.float32)
