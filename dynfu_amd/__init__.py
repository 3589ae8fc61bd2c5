"""dynfu_amd — MI355X-native (gfx950) TSDF fusion + warp-field solve behind a C ABI.

This package is plumbing only: it loads ``libdynfu_amd.so`` (hand-written HIP kernels +
the C ABI of ``include/dynfu_amd.h``) and exposes the entry points on torch CUDA tensors
(torch is used for device memory and streams, nothing else).  There is NO CPU fallback: if
the library is missing or no GPU is visible every compute call raises.
"""
from . import _lib
from ._lib import (icp_sums, calc_dqb, unsupported_vertices, repack_points, compact_points, transform_points, warp_to_live_graph, DynfuAmdError, Solve6Params, SolveParams, Solver, Solver6, compute_points_normals, compute_normals_mask_depth, depth_bilateral_filter,
                   depth_build_pyramid, depth_truncate, resize_depth_normals, resize_points_normals, compute_dists, correspond, knn, marching_cubes, mc_default_tables, lib_path, load, tsdf_clear,
                   tsdf_clear_integrate, tsdf_integrate, tsdf_occupancy, tsdf_raycast_depth, tsdf_raycast_points, tsdf_raycast_tally, tsdf_vertex_normals, correspond_projective, version,
                   warp_to_live)

__all__ = ["icp_sums", "calc_dqb", "unsupported_vertices", "repack_points", "compact_points", "transform_points", "warp_to_live_graph", "DynfuAmdError", "Solve6Params", "SolveParams", "Solver", "Solver6", "compute_points_normals", "compute_normals_mask_depth", "depth_bilateral_filter", "depth_build_pyramid",
           "depth_truncate", "resize_depth_normals", "resize_points_normals", "compute_dists", "correspond", "knn", "marching_cubes", "mc_default_tables", "lib_path", "load", "tsdf_clear",
           "tsdf_clear_integrate", "tsdf_integrate", "tsdf_occupancy", "tsdf_raycast_depth", "tsdf_raycast_points", "tsdf_raycast_tally", "tsdf_vertex_normals", "correspond_projective", "version",
           "warp_to_live", "_lib"]
