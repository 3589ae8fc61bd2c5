"""dev tool: launches every TSDF-seam kernel of one configuration a few times (for rocprofv3
--kernel-trace / --pmc runs) and prints hipEvent timings + algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = synth.CONFIGS[name]
fx, fy, cx, cy = synth.intrinsics(cfg)
voxel, trunc, vol2cam, cam2vol, rinv = synth.volume_params(cfg)
dim, W, H = cfg["dim"], cfg["width"], cfg["height"]
depth = torch.from_numpy(synth.depth_frame(cfg, 0)).cuda()
dists = torch.empty((H, W), dtype=torch.uint16, device="cuda")
vol = torch.zeros((dim, dim, dim), dtype=torch.int32, device="cuda")
pts = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
nrm = torch.empty_like(pts)
dep = torch.empty((H, W), dtype=torch.uint16, device="cuda")
V = dim ** 3


def timeit(label, fn, nbytes):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    print("%-28s %9.4f ms (min %9.4f)  %8.1f GB/s algorithmic" % (label, ms.mean(), ms.min(), nbytes / ms.mean() / 1e6))


timeit("compute_dists", lambda: A.compute_dists(depth, dists, fx, fy, cx, cy), 4 * W * H)
timeit("clear", lambda: A.tsdf_clear(vol), 4 * V)
A.tsdf_clear(vol)
timeit("integrate (read+write)", lambda: A.tsdf_integrate(vol, dists, voxel, trunc, 64, vol2cam, fx, fy, cx, cy), 8 * V + 2 * W * H)
touched = int(((vol >> 16) > 0).sum())
print("   voxels updated per sweep: %d of %d (%.1f %%) -> touched-voxel traffic %.3f GB" % (touched, V, 100 * touched / V, (4 * V + 4 * touched) / 1e9))
timeit("clear+integrate fused", lambda: A.tsdf_clear_integrate(vol, dists, voxel, trunc, 64, vol2cam, fx, fy, cx, cy), 4 * V + 2 * W * H)
timeit("raycast points", lambda: A.tsdf_raycast_points(vol, voxel, trunc, cam2vol, rinv, fx, fy, cx, cy, 0.75, 0.5, pts, nrm), 32 * W * H)
timeit("raycast depth", lambda: A.tsdf_raycast_depth(vol, voxel, trunc, cam2vol, rinv, fx, fy, cx, cy, 0.75, 0.5, dep, nrm), 18 * W * H)
hits = int((~torch.isnan(pts[..., 0])).sum())
print("   raycast hits: %d of %d pixels" % (hits, W * H))
