"""dev tool: times the fused and the read+write TSDF sweep over the A/B switches of tsdf.hip in ONE process
(DFA_TSDF_RUN / DFA_TSDF_WAVE / DFA_TSDF_ZCHUNK / DFA_TSDF_LEGACY / DFA_TSDF_ABLATE are read per call)."""
import os as _os
_os.environ.setdefault("DFA_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dynfu_amd", "libdynfu_amd_dev.so"))  # the DFA_* switches exist in the development flavour only
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = synth.CONFIGS[name]
intr = synth.intrinsics(cfg)
voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
dim, W, H = cfg["dim"], cfg["width"], cfg["height"]
depth = torch.from_numpy(synth.depth_frame(cfg, 0)).cuda()
dists = torch.empty((H, W), dtype=torch.uint16, device="cuda")
A.compute_dists(depth, dists, *intr)
vol = torch.zeros((dim, dim, dim), dtype=torch.int32, device="cuda")


def timeit(fn):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def variant(env):
    for k in ("DFA_TSDF_RUN", "DFA_TSDF_WAVE", "DFA_TSDF_ZCHUNK", "DFA_TSDF_LEGACY", "DFA_TSDF_ABLATE"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items() if v is not None})
    f = timeit(lambda: A.tsdf_clear_integrate(vol, dists, voxel, trunc, 64, vol2cam, *intr))
    A.tsdf_clear(vol)
    r = timeit(lambda: A.tsdf_integrate(vol, dists, voxel, trunc, 64, vol2cam, *intr))
    print("%-60s fused %.4f ms (%.0f GB/s)   read+write %.4f ms" % (env, f, 4 * dim ** 3 / f / 1e6, r), flush=True)


variant({"DFA_TSDF_LEGACY": 1})
variant({"DFA_TSDF_LEGACY": 1, "DFA_TSDF_ZCHUNK": 128})
variant({"DFA_TSDF_ABLATE": 3})
variant({"DFA_TSDF_ABLATE": 3, "DFA_TSDF_ZCHUNK": 128})
for u, w, zc in itertools.product((4, 8), (64, 32, 16), (None, 256, 128, 64)):
    variant({"DFA_TSDF_RUN": u, "DFA_TSDF_WAVE": w, "DFA_TSDF_ZCHUNK": zc})
