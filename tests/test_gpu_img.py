"""-m gpu parity tests of the depth pre-processing seam (src/kfusion/cuda/imgproc.cu) vs the oracle.
Bar: BIT-EXACT, pitched inputs included; the oracle is unpinned (no reference tests)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from dynfu_amd import synth  # noqa: E402
from gpu_util import bits, dev, host  # noqa: E402


@pytest.fixture(scope="module")
def A():
    import dynfu_amd
    dynfu_amd.load()
    return dynfu_amd


def _depth(H, W, seed):
    rng = np.random.default_rng(seed)
    d = (1200 + 400 * np.sin(np.arange(W) / 11.0)[None, :] + 300 * np.cos(np.arange(H) / 7.0)[:, None]
         + rng.normal(0, 6, (H, W))).astype(np.uint16)
    d[rng.random((H, W)) < 0.04] = 0
    d[rng.random((H, W)) < 0.01] = 60000
    return d


SHAPES = [(480, 640), (37, 53), (8, 32), (1, 1), (2, 2), (9, 200)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("ksz,ss,sd", [(7, 4.5, 0.04), (3, 1.0, 0.2), (17, 6.0, 0.05)])
def test_bilateral_bit_exact(A, shape, ksz, ss, sd):
    d = _depth(*shape, seed=ksz)
    assert np.array_equal(host(A.depth_bilateral_filter(dev(d), ksz, ss, sd)), O.bilateral(d, ksz, ss, sd))


def test_bilateral_on_the_synthetic_frame_and_pitched_rows(A):
    import torch
    cfg = synth.CONFIGS["C2"]
    d = synth.depth_frame(cfg, 7, noise_mm=1.0)
    H, W = d.shape
    buf = torch.zeros((H, W + 10), dtype=torch.uint16, device="cuda")
    buf[:, :W] = dev(d)
    out = A.depth_bilateral_filter(buf[:, :W], 7, 4.5, 0.04)   # row stride != cols
    assert np.array_equal(host(out), O.bilateral(d, 7, 4.5, 0.04))


@pytest.mark.parametrize("shape", SHAPES)
def test_truncate_pyramid_normals_resizers_bit_exact(A, shape):
    d = _depth(*shape, seed=5)
    H, W = shape
    fx, fy, cx, cy = 525.0 * W / 640, 525.0 * W / 640, W / 2 - 0.5, H / 2 - 0.5
    t = dev(d)
    A.depth_truncate(t, 1.4)
    assert np.array_equal(host(t), O.truncate_depth(d, 1.4))
    assert np.array_equal(host(A.depth_build_pyramid(dev(d), 0.04)), O.depth_pyr(d, 0.04))
    m = dev(d)
    n = A.compute_normals_mask_depth(m, fx, fy, cx, cy)
    rm, rn = O.normals_mask_depth(d, fx, fy, cx, cy)
    assert np.array_equal(host(m), rm) and np.array_equal(bits(host(n)), bits(rn))
    d2, n2 = A.resize_depth_normals(m, n)
    rd2, rn2 = O.resize_depth_normals(rm, rn)
    assert np.array_equal(host(d2), rd2) and np.array_equal(bits(host(n2)), bits(rn2))
    P, Nm = A.compute_points_normals(dev(d), fx, fy, cx, cy)
    v2, m2 = A.resize_points_normals(P, Nm)
    rv2, rm2 = O.resize_points_normals(host(P), host(Nm))
    assert np.array_equal(bits(host(v2)), bits(rv2)) and np.array_equal(bits(host(m2)), bits(rm2))


def test_errors(A):
    import torch
    d = torch.zeros((4, 4), dtype=torch.uint16, device="cuda")
    with pytest.raises(A.DynfuAmdError):
        A.depth_bilateral_filter(d, 0, 1.0, 1.0)
    with pytest.raises(A.DynfuAmdError):
        A.depth_truncate(d, 100.0)
