// kfusion/cuda/imgproc.hpp — the depth-frame functions of the reference's include/kfusion/cuda/imgproc.hpp
// (:7-24) as thin wrappers of the dynfu_amd C ABI: each allocates its output like the reference's wrapper in
// src/kfusion/imgproc.cpp and makes ONE dfa_* call on the default stream.
//
//   function                      C ABI call                          reference kernel (imgproc.cu)
//   depthBilateralFilter          dfa_depth_bilateral_filter          bilateral_kernel :8
//   depthTruncation               dfa_depth_truncate                  truncate_depth_kernel :60
//   depthBuildPyramid             dfa_depth_build_pyramid             pyramid_kernel :84
//   computeNormalsAndMaskDepth    dfa_compute_normals_mask_depth      compute_normals_kernel :129 + mask_depth_kernel :159
//   computePointNormals           dfa_compute_points_normals          points_normals_kernel :187
//   resizeDepthNormals            dfa_resize_depth_normals            resize_depth_normals_kernel :258
//   resizePointsNormals           dfa_resize_points_normals           resize_points_normals_kernel :314
//   (computeDists, waitAllDefaultStream: kfusion/types.hpp; renderImage / renderTangentColors: visualisation, absent)
#pragma once
#include <kfusion/types.hpp>

namespace kfusion {
namespace cuda {

// whole-frame filters (full resolution in, full resolution out)
void depthBilateralFilter(const Depth& in, Depth& out, int ksz, float sigma_spatial, float sigma_depth);
void depthTruncation(Depth& depth, float threshold /* metres; in place */);

// geometry of a depth frame
void computePointNormals(const Intr& intr, const Depth& depth, Cloud& points, Normals& normals);
void computeNormalsAndMaskDepth(const Intr& intr, Depth& depth /* masked in place */, Normals& normals);

// half-resolution levels of the tracking pyramid
void depthBuildPyramid(const Depth& depth, Depth& pyramid, float sigma_depth);
void resizePointsNormals(const Cloud& points, const Normals& normals, Cloud& points_out, Normals& normals_out);
void resizeDepthNormals(const Depth& depth, const Normals& normals, Depth& depth_out, Normals& normals_out);

}  // namespace cuda
}  // namespace kfusion
