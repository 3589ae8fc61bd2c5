"""-m gpu parity tests of the rigid-ICP seam (src/kfusion/cuda/proj_icp.cu) vs the oracle.

Bar: the set of matched pixels is identical (integer count), the 27 sums agree to 1e-5 relative to the largest
sum (the reference itself tree-reduces floats, so its sums depend on the reduction order; the HIP kernel adds
float wave totals and double partials, the oracle adds doubles in pixel order).  Oracle unpinned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from dynfu_amd import synth  # noqa: E402
from gpu_util import dev, host, rot  # noqa: E402


@pytest.fixture(scope="module")
def A():
    import dynfu_amd
    dynfu_amd.load()
    return dynfu_amd


def _frames(name="T1"):
    cfg = synth.CONFIGS[name]
    intr = synth.intrinsics(cfg)
    d0, d1 = synth.depth_frame(cfg, 0), synth.depth_frame(cfg, 4)
    return cfg, intr, d0, d1


@pytest.mark.parametrize("variant", ["depth", "points"])
@pytest.mark.parametrize("level", [0, 1])
def test_icp_sums_match_the_oracle(A, variant, level):
    cfg, intr, d0, d1 = _frames()
    fx, fy, cx, cy = intr
    # the reference's inputs: masked depth + normals per pyramid level (kinfu.cpp:150-175)
    m0, n0 = O.normals_mask_depth(d0, *intr)
    m1, n1 = O.normals_mask_depth(d1, *intr)
    P0, N0 = O.points_normals(d0, *intr)
    P1, N1 = O.points_normals(d1, *intr)
    if level == 1:
        m0, n0 = O.resize_depth_normals(m0, n0)
        m1, n1 = O.resize_depth_normals(m1, n1)
        P0, N0 = O.resize_points_normals(P0, N0)
        P1, N1 = O.resize_points_normals(P1, N1)
    div = 1 << level
    li = (fx / div, fy / div, cx / div, cy / div)
    aff = np.concatenate([rot([0.2, 1.0, 0.1], 0.01).astype(np.float32).reshape(-1), np.array([0.004, -0.003, 0.006], np.float32)])
    if variant == "depth":
        args = (m1, n1, m0, n0)
    else:
        args = (P1, N1, P0, N0)
    ref, matched = O.icp_sums(*args, aff, li)
    sums, m = A.icp_sums(*(dev(a) for a in args), aff, *li)
    assert int(host(m)[0]) == matched and matched > 0.5 * m0.size
    assert np.abs(host(sums).astype(np.float64) - ref).max() <= 1e-5 * np.abs(ref).max()


def test_icp_sums_no_overlap_and_errors(A):
    import torch
    cfg, intr, d0, d1 = _frames("T0")
    m0, n0 = O.normals_mask_depth(d0, *intr)
    far = np.concatenate([np.eye(3, dtype=np.float32).reshape(-1), np.array([5.0, 0, 0], np.float32)])  # nothing projects
    sums, m = A.icp_sums(dev(m0), dev(n0), dev(m0), dev(n0), far, *intr)
    assert int(host(m)[0]) == 0 and not host(sums).any()
    ident = np.concatenate([np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)])
    sums, m = A.icp_sums(dev(m0), dev(n0), dev(m0), dev(n0), ident, *intr)
    h = host(sums)
    assert int(host(m)[0]) > 0.97 * int((m0 != 0).sum())  # a frame matches itself (the re-projected pixel may round to a neighbour)
    assert np.abs(h[[6, 12, 17, 21, 24, 26]]).max() <= 1e-3 * np.abs(h).max()  # b ~ 0: already aligned (up to the re-projection rounding)
    with pytest.raises(A.DynfuAmdError):
        A.icp_sums(dev(m0), dev(n0), dev(m0), dev(n0), ident, 0.0, 1.0, 0.0, 0.0)
