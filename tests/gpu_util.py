"""helpers shared by the -m gpu parity tests"""
import numpy as np
import torch


def dev(a):
    """numpy -> CUDA tensor (uint32 volumes travel as int32, same bits)"""
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).cuda()
    return torch.from_numpy(a).cuda()


def host(t, dtype=None):
    a = t.detach().cpu().numpy()
    if dtype is not None:
        a = a.view(dtype)
    return a


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rot(axis, ang):
    axis = np.asarray(axis, float)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def aff12(R, t):
    return np.concatenate([np.asarray(R, np.float32).reshape(-1), np.asarray(t, np.float32)]).astype(np.float32)
