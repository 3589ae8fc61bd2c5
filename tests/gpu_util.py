"""helpers shared by the -m gpu parity tests"""
import numpy as np
import torch


def dev(a):
    """numpy -> CUDA tensor (uint32 volumes travel as int32, same bits)"""
    a = np.ascontiguousarray(a)
    if not a.flags.writeable:  # (memoised synthetic frames are read-only; torch wants to own writable memory)
        a = a.copy()
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


from gpu_util_cpu import aff12, rot  # noqa: E402,F401
