"""Writes tests/golden/depth_t0_frame7.png: frame 7 of the synthetic T0 sequence (dynfu_amd/synth.py, 160x120, millimetres)
as a 16-bit greyscale PNG, encoded by Pillow (an encoder that is not ours: adaptive filters, its own deflate settings).
tests/test_host_cpp.py decodes it with dfa::io::readDepthPng and compares with synth.depth_frame.
    python tests/golden/make_depth_png.py"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynfu_amd import synth  # noqa: E402

if __name__ == "__main__":
    d = synth.depth_frame(synth.CONFIGS["T0"], 7, noise_mm=1.0)
    Image.fromarray(d.astype(np.uint16)).save(os.path.join(ROOT, "tests", "golden", "depth_t0_frame7.png"), optimize=True)
    print(d.shape, d.dtype, int(d.max()))
