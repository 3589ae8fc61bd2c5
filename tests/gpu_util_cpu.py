"""the torch-free helpers of gpu_util (rotation / affine packing) for the CPU suite"""
import numpy as np


def rot(axis, ang):
    axis = np.asarray(axis, float)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def aff12(R, t):
    return np.concatenate([np.asarray(R, np.float32).reshape(-1), np.asarray(t, np.float32)]).astype(np.float32)
