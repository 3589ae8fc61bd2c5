// imgproc.cpp — kfusion::cuda image functions on the dynfu_amd C ABI (reference: src/kfusion/imgproc.cpp:3-66;
// every function allocates its output like the reference does, then makes one C-ABI call).
#include <kfusion/cuda/imgproc.hpp>

#include "../../../include/dynfu_amd.h"

namespace kfusion {
namespace cuda {

void depthBilateralFilter(const Depth& in, Depth& out, int ksz, float sigma_spatial, float sigma_depth) {
    out.create(in.rows(), in.cols());  // :5
    dfa::check(dfa_depth_bilateral_filter(in.ptr(), (int)in.step(), out.ptr(), (int)out.step(), in.cols(), in.rows(), ksz,
                                          sigma_spatial, sigma_depth, nullptr),
               "depthBilateralFilter");
}

void depthTruncation(Depth& depth, float threshold) {
    dfa::check(dfa_depth_truncate(depth.ptr(), (int)depth.step(), depth.cols(), depth.rows(), threshold, nullptr),
               "depthTruncation");
}

void depthBuildPyramid(const Depth& depth, Depth& pyramid, float sigma_depth) {
    pyramid.create(depth.rows() / 2, depth.cols() / 2);  // :12
    dfa::check(dfa_depth_build_pyramid(depth.ptr(), (int)depth.step(), depth.cols(), depth.rows(), pyramid.ptr(),
                                       (int)pyramid.step(), sigma_depth, nullptr),
               "depthBuildPyramid");
}

void computeNormalsAndMaskDepth(const Intr& intr, Depth& depth, Normals& normals) {
    normals.create(depth.rows(), depth.cols());  // :19
    dfa::check(dfa_compute_normals_mask_depth(depth.ptr(), (int)depth.step(), depth.cols(), depth.rows(), intr.fx, intr.fy,
                                              intr.cx, intr.cy, (float*)normals.ptr(), (int)normals.step(), nullptr),
               "computeNormalsAndMaskDepth");
}

void computePointNormals(const Intr& intr, const Depth& depth, Cloud& points, Normals& normals) {
    points.create(depth.rows(), depth.cols());  // :28-29
    normals.create(depth.rows(), depth.cols());
    dfa::check(dfa_compute_points_normals(depth.ptr(), (int)depth.step(), depth.cols(), depth.rows(), intr.fx, intr.fy,
                                          intr.cx, intr.cy, (float*)points.ptr(), (int)points.step(), (float*)normals.ptr(),
                                          (int)normals.step(), nullptr),
               "computePointNormals");
}

void resizeDepthNormals(const Depth& depth, const Normals& normals, Depth& depth_out, Normals& normals_out) {
    depth_out.create(depth.rows() / 2, depth.cols() / 2);  // :45-46
    normals_out.create(normals.rows() / 2, normals.cols() / 2);
    dfa::check(dfa_resize_depth_normals(depth.ptr(), (int)depth.step(), (const float*)normals.ptr(), (int)normals.step(),
                                        depth.cols(), depth.rows(), depth_out.ptr(), (int)depth_out.step(),
                                        (float*)normals_out.ptr(), (int)normals_out.step(), nullptr),
               "resizeDepthNormals");
}

void resizePointsNormals(const Cloud& points, const Normals& normals, Cloud& points_out, Normals& normals_out) {
    points_out.create(points.rows() / 2, points.cols() / 2);  // :56-57
    normals_out.create(normals.rows() / 2, normals.cols() / 2);
    dfa::check(dfa_resize_points_normals((const float*)points.ptr(), (int)points.step(), (const float*)normals.ptr(),
                                         (int)normals.step(), points.cols(), points.rows(), (float*)points_out.ptr(),
                                         (int)points_out.step(), (float*)normals_out.ptr(), (int)normals_out.step(), nullptr),
               "resizePointsNormals");
}

}  // namespace cuda
}  // namespace kfusion
