"""shared by the marching-cubes tests: analytic packed TSDF volumes (numpy) and the case tables"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def default_tables():
    """dfa_mc_default_tables is host code: callable without a GPU (and without torch)"""
    lib = ctypes.CDLL(os.path.join(ROOT, "dynfu_amd", "libdynfu_amd.so"))
    tri, nv = np.zeros((256, 16), np.int32), np.zeros(256, np.int32)
    assert lib.dfa_mc_default_tables(tri.ctypes.data_as(ctypes.c_void_p), nv.ctypes.data_as(ctypes.c_void_p)) == 0
    return tri, nv


def pack(tsdf, weight):
    """float tsdf in [-1, 1] -> half bits (numpy's conversion is round-to-nearest-even) | weight << 16"""
    h = np.asarray(tsdf, np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
    return h | (np.asarray(weight, np.uint32) << 16)


def blob_volume(dims, seed=0, holes=True, trunc=0.1):
    """(Z, Y, X) uint32 volume: truncated signed distance of two overlapping spheres, weight 0
    outside the truncation band and in random holes (exercises the "any weight == 0" rule)"""
    X, Y, Z = dims
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid((np.arange(Z) + 0.5) / Z, (np.arange(Y) + 0.5) / Y, (np.arange(X) + 0.5) / X, indexing="ij")
    d1 = np.sqrt((x - 0.4) ** 2 + (y - 0.45) ** 2 + (z - 0.5) ** 2) - 0.27
    d2 = np.sqrt((x - 0.65) ** 2 + (y - 0.6) ** 2 + (z - 0.45) ** 2) - 0.2
    d = np.minimum(d1, d2)
    tsdf = np.clip(d / trunc, -1, 1)
    w = (np.abs(d) < trunc).astype(np.uint32) * rng.integers(1, 65, d.shape).astype(np.uint32)
    if holes:
        w[rng.random(d.shape) < 0.02] = 0
        tsdf = np.where(rng.random(d.shape) < 0.01, 0.0, tsdf)  # exact zeros: f < iso is false
        tsdf = np.where(rng.random(d.shape) < 0.005, -0.0, tsdf)
    return pack(tsdf, w)
