// tsdf_volume.cpp — kfusion::cuda::TsdfVolume on the dynfu_amd C ABI
// (host logic follows src/kfusion/tsdf_volume.cpp:18-129; kernels are in libdynfu_amd.so).
#include <algorithm>

#include <kfusion/cuda/tsdf_volume.hpp>

#include "../../../include/dynfu_amd.h"

using namespace kfusion;
using namespace kfusion::cuda;

void kfusion::cuda::computeDists(const Depth& depth, Dists& dists, const Intr& intr) {
    dists.create(depth.rows(), depth.cols());  // imgproc.cpp:39
    dfa::check(dfa_compute_dists(depth.ptr(), (int)depth.step(), dists.ptr(), (int)dists.step(), depth.cols(),
                                 depth.rows(), intr.fx, intr.fy, intr.cx, intr.cy, nullptr),
               "computeDists");
}

float TsdfVolume::Entry::half2float(half) { throw "Not implemented"; }               // tsdf_volume.cpp:9
TsdfVolume::Entry::half TsdfVolume::Entry::float2half(float) { throw "Not implemented"; }  // :11-13

void TsdfVolume::create(const Vec3i& dims) {  // :32-38
    cfg_.dims = dims;
    blob_.create((size_t)dims[0] * dims[1] * dims[2] * sizeof(int));
    occ_.create(dfa_tsdf_occupancy_bytes(dims[0], dims[1], dims[2]));
    setTruncDist(cfg_.trunc);
    clear();
}

void TsdfVolume::clear() {  // :74-80
    dfa::check(dfa_tsdf_clear_occ(blob_.ptr<uint32_t>(), cfg_.dims[0], cfg_.dims[1], cfg_.dims[2], occ_.ptr<uint8_t>(), nullptr),
               "TsdfVolume::clear");
    occ_known_ = soleOwner();  // (a handle that is alive now may write the voxels later: no promise then)
}

void TsdfVolume::integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr) {  // :82-93
    Affine3f vol2cam = camera_pose.inv() * cfg_.pose;
    float aff[12];
    vol2cam.to12(aff);
    const Vec3f vsz = getVoxelSize();
    if (mapTrusted())  // (the accumulating sweep only ADDS to the map: it has to be right before)
        dfa::check(dfa_tsdf_integrate_occ(dists.ptr(), (int)dists.step(), dists.cols(), dists.rows(), blob_.ptr<uint32_t>(),
                                          cfg_.dims[0], cfg_.dims[1], cfg_.dims[2], vsz.v, cfg_.trunc, cfg_.max_weight, aff, intr.fx,
                                          intr.fy, intr.cx, intr.cy, occ_.ptr<uint8_t>(), nullptr),
                   "TsdfVolume::integrate");
    else
        dfa::check(dfa_tsdf_integrate(dists.ptr(), (int)dists.step(), dists.cols(), dists.rows(), blob_.ptr<uint32_t>(),
                                      cfg_.dims[0], cfg_.dims[1], cfg_.dims[2], vsz.v, cfg_.trunc, cfg_.max_weight, aff, intr.fx,
                                      intr.fy, intr.cx, intr.cy, nullptr),
                   "TsdfVolume::integrate");
    dfa::device_synchronize();  // the reference's device::integrate blocks (tsdf_volume.cu:120)
}

void TsdfVolume::clearAndIntegrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr) {
    Affine3f vol2cam = camera_pose.inv() * cfg_.pose;
    float aff[12];
    vol2cam.to12(aff);
    const Vec3f vsz = getVoxelSize();
    // (a map that describes the volume: boxes of zeros that stay zeros are not written again)
    dfa::check((mapTrusted() ? dfa_tsdf_clear_integrate_known_occ : dfa_tsdf_clear_integrate_occ)(
                   dists.ptr(), (int)dists.step(), dists.cols(), dists.rows(), blob_.ptr<uint32_t>(), cfg_.dims[0], cfg_.dims[1],
                   cfg_.dims[2], vsz.v, cfg_.trunc, cfg_.max_weight, aff, intr.fx, intr.fy, intr.cx, intr.cy, occ_.ptr<uint8_t>(), nullptr),
               "TsdfVolume::clearAndIntegrate");
    occ_known_ = soleOwner();  // (the fused sweep leaves volume and map describing each other — while nobody else can write)
    dfa::device_synchronize();
}

void TsdfVolume::raycast(const Affine3f& camera_pose, const Intr& intr, Depth& depth, Normals& normals) {  // :95-110
    Affine3f cam2vol = cfg_.pose.inv() * camera_pose;
    float aff[12], rinv[9];
    cam2vol.to12(aff);
    cam2vol.inverse_rotation(rinv);
    const Vec3f vsz = getVoxelSize();
    dfa::check(dfa_tsdf_raycast_depth(blob_.ptr<uint32_t>(), cfg_.dims[0], cfg_.dims[1], cfg_.dims[2], vsz.v, cfg_.trunc, aff, rinv,
                                      intr.fx, intr.fy, intr.cx, intr.cy, cfg_.ray_step, cfg_.grad_delta,
                                      depth.ptr(), (int)depth.step(), (float*)normals.ptr(), (int)normals.step(),
                                      depth.cols(), depth.rows(), nullptr),
               "TsdfVolume::raycast(depth)");
}

void TsdfVolume::raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals) {  // :112-129
    Affine3f cam2vol = cfg_.pose.inv() * camera_pose;
    float aff[12], rinv[9];
    cam2vol.to12(aff);
    cam2vol.inverse_rotation(rinv);
    const Vec3f vsz = getVoxelSize();
    dfa::check(dfa_tsdf_raycast_points(blob_.ptr<uint32_t>(), cfg_.dims[0], cfg_.dims[1], cfg_.dims[2], vsz.v, cfg_.trunc, aff,
                                       rinv, intr.fx, intr.fy, intr.cx, intr.cy, cfg_.ray_step,
                                       cfg_.grad_delta, (float*)points.ptr(), (int)points.step(),
                                       (float*)normals.ptr(), (int)normals.step(), points.cols(), points.rows(),
                                       nullptr),
               "TsdfVolume::raycast(points)");
}
