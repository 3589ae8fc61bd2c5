"""CPU tests of the depth pre-processing oracle (oracle/img_oracle.c, restating
src/kfusion/cuda/imgproc.cu).  The reference has no tests for these kernels; checked here are the
properties the reference's code implies, including its quirks."""
import numpy as np

import oracle as O


def _depth(H, W, seed=0):
    rng = np.random.default_rng(seed)
    d = (1200 + 400 * np.sin(np.arange(W) / 11.0)[None, :] + 300 * np.cos(np.arange(H) / 7.0)[:, None]
         + rng.normal(0, 3, (H, W))).astype(np.uint16)
    d[rng.random((H, W)) < 0.03] = 0
    return d


def test_exp_stand_in_is_accurate_and_monotone():
    xs = -np.logspace(-6, 1.9, 3000)
    e = np.array([O._libimg().orc_exp_neg(float(x)) for x in xs])
    ref = np.exp(xs.astype(np.float64))
    assert np.abs(e / ref - 1).max() < 1e-5  # same order as the hardware __expf it replaces
    assert np.all(np.diff(e) <= 0) and O._libimg().orc_exp_neg(0.0) == 1.0 and O._libimg().orc_exp_neg(-200.0) == 0.0


def test_bilateral_smooths_noise_keeps_edges_and_follows_the_window_quirk():
    H, W = 40, 60
    rng = np.random.default_rng(1)
    d = np.full((H, W), 1000, np.uint16)
    d[:, 30:] = 1500                                   # a 0.5 m step
    noisy = (d + rng.integers(-4, 5, d.shape)).astype(np.uint16)
    out = O.bilateral(noisy, 7, 4.5, 0.04)             # kinfu.cpp:26-28 defaults
    inner = (slice(5, 35), slice(5, 25))
    assert np.abs(out[inner].astype(int) - 1000).mean() < np.abs(noisy[inner].astype(int) - 1000).mean() * 0.6
    assert abs(int(out[20, 28]) - 1000) <= 4 and abs(int(out[20, 31]) - 1500) <= 4  # the step survives
    # :17-18,23-24: the window stops one short of the last row / column, so the last column's own pixel is
    # outside its window; with a constant image the result is still that constant
    c = np.full((9, 9), 777, np.uint16)
    assert np.all(O.bilateral(c, 5, 3.0, 0.05) == 777)
    # a 1 x 1 image has an empty window: 0 / 0 -> 0 (the device's float->int conversion of NaN)
    assert O.bilateral(np.array([[500]], np.uint16), 3, 1.0, 0.1)[0, 0] == 0


def test_truncate_and_pyramid():
    d = _depth(31, 45, 2)
    t = O.truncate_depth(d, 1.3)
    assert np.array_equal(t, np.where(d > 1300, 0, d))
    p = O.depth_pyr(d, 0.04)
    assert p.shape == (15, 22)
    # each output is the integer mean of the window values within 3 sigma (120 mm) of the centre
    y, x = 6, 9
    win = d[max(0, 2 * y - 2):min(2 * y + 3, 30), max(0, 2 * x - 2):min(2 * x + 3, 44)].astype(int)
    sel = win[np.abs(win - int(d[2 * y, 2 * x])) < 120.0]
    assert p[y, x] == (0 if len(sel) == 0 else sel.sum() // len(sel))
    assert O.depth_pyr(np.zeros((1, 7), np.uint16), 0.04).shape == (0, 3)


def test_normals_mask_and_resizers():
    d = _depth(24, 32, 3)
    fx = fy = 40.0
    cx, cy = 15.5, 11.5
    dm, n = O.normals_mask_depth(d, fx, fy, cx, cy)
    P, Nm = O.points_normals(d, fx, fy, cx, cy)       # the sibling kernel computes the same normals
    assert np.array_equal(np.isnan(n[..., 0]), np.isnan(Nm[..., 0]))
    ok = ~np.isnan(n[..., 0])
    assert np.array_equal(n[ok][:, :3], Nm[ok][:, :3]) and np.all(n[..., 3] == 0)
    assert np.all(dm[~ok] == 0) and np.array_equal(dm[ok], d[ok])
    d2, n2 = O.resize_depth_normals(dm, n)
    assert d2.shape == (12, 16)
    q = dm.reshape(12, 2, 16, 2).astype(int)
    full = (q[:, 0, :, 0] * q[:, 0, :, 1] != 0) & (q[:, 1, :, 0] * q[:, 1, :, 1] != 0)
    assert np.array_equal(d2[full], (q.sum((1, 3)) // 4)[full]) and np.all(d2[~full] == 0)
    assert np.all(np.isnan(n2[~full])) and np.all(np.isnan(n2[full][:, 3]))  # w stays NaN (:271)
    v2, m2 = O.resize_points_normals(P, Nm)
    valid = ~np.isnan(P[..., 0]).reshape(12, 2, 16, 2).any((1, 3))
    assert np.array_equal(~np.isnan(v2[..., 0]), valid)
    assert np.allclose(v2[valid][:, :3], P[..., :3].reshape(12, 2, 16, 2, 3).sum((1, 3))[valid] / 4, rtol=1e-6)
