"""CPU checks of the TSDF restatement's own plumbing (the TSDF oracle is UNPINNED — the reference holds no vector for
it, oracle/oracle.h): a slab of slices computed by orc_tsdf_integrate_slab is bit-for-bit the corresponding part of the
full sweep.  That property is what lets the -m gpu tests compare a 1024^3 volume slab by slab."""
import numpy as np

import oracle as O
from dynfu_amd import synth
from gpu_util_cpu import aff12, rot


def test_slab_sweep_equals_the_slices_of_the_full_sweep():
    cfg = synth.CONFIGS["T0"]
    intr = synth.intrinsics(cfg)
    voxel, trunc, vol2cam, _, _ = synth.volume_params(cfg)
    vol2cam = aff12(rot([1, 0.2, 0], 0.15), vol2cam[9:])  # a zstep with three non-zero components
    d = O.compute_dists(synth.depth_frame(cfg, 0), *intr)
    dim = cfg["dim"]
    full = np.zeros((dim, dim, dim), np.uint32)
    O.tsdf_integrate(full, d, voxel, trunc, 64, vol2cam, *intr, threads=4)
    O.tsdf_integrate(full, d, voxel, trunc, 64, vol2cam, *intr, threads=4)  # second sweep: previous values are read
    assert (full != 0).mean() > 0.05
    for z0, z1 in ((0, 5), (17, 40), (dim - 9, dim)):
        slab = np.zeros((z1 - z0, dim, dim), np.uint32)
        for _ in range(2):
            O.tsdf_integrate_slab(slab, z0, d, voxel, trunc, 64, vol2cam, *intr, threads=4)
        assert np.array_equal(slab, full[z0:z1])
