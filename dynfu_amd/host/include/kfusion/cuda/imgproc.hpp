// kfusion/cuda/imgproc.hpp — the reference's include/kfusion/cuda/imgproc.hpp:7-24 (depth pre-processing
// and image utilities; the render* functions are visualisation and not provided) on the dynfu_amd C ABI.
#pragma once
#include <kfusion/types.hpp>

namespace kfusion {
namespace cuda {
void depthBilateralFilter(const Depth& in, Depth& out, int ksz, float sigma_spatial, float sigma_depth);
void depthTruncation(Depth& depth, float threshold);
void depthBuildPyramid(const Depth& depth, Depth& pyramid, float sigma_depth);
void computeNormalsAndMaskDepth(const Intr& intr, Depth& depth, Normals& normals);
void computePointNormals(const Intr& intr, const Depth& depth, Cloud& points, Normals& normals);
// computeDists and waitAllDefaultStream: kfusion/types.hpp
void resizeDepthNormals(const Depth& depth, const Normals& normals, Depth& depth_out, Normals& normals_out);
void resizePointsNormals(const Cloud& points, const Normals& normals, Cloud& points_out, Normals& normals_out);
}  // namespace cuda
}  // namespace kfusion
