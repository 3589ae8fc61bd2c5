"""dev tool: the C++ adaptor's per-frame time (DynFusion::operator() through runSequence: PNG in, pcd / vtk out) on a
synthetic VGA sequence — what a caller of the reference's classes (INTEGRATION.md option A) gets, next to bench.py's
device-resident path."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from dynfu_amd import synth, build as B

exe = B.build_cpp_tests()["test_host_io"]
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
with tempfile.TemporaryDirectory() as d:
    os.makedirs(d + "/depth"), os.makedirs(d + "/color")
    for f in range(n):
        Image.fromarray(synth.depth_frame(cfg, f)).save("%s/depth/frame-%06d.depth.png" % (d, f))
        Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save("%s/color/frame-%06d.color.png" % (d, f))
    env = dict(os.environ, DFA_SEQ_DIM=str(cfg["dim"]), DFA_SEQ_FRAME_MS="1")
    r = subprocess.run([exe, "sequence", d], capture_output=True, text=True, env=env)
    print(r.stdout, r.stderr[-3000:])
    for line in r.stdout.splitlines():
        if line.startswith("frame_ms"):
            ms = [float(v) for v in line.split()[1:]]
            if len(ms) > 2:
                print("operator() per frame, frames 2..: median %.2f ms, min %.2f ms" % (float(np.median(ms[2:])), min(ms[2:])))
