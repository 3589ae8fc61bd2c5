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

TsdfVolume::TsdfVolume(const Vec3i& dims)  // :18-28
    : data_(), trunc_dist_(0.03f), max_weight_(128), dims_(dims), size_(Vec3f::all(3.f)),
      pose_(Affine3f::Identity()), gradient_delta_factor_(0.75f), raycast_step_factor_(0.75f) {
    create(dims_);
}
TsdfVolume::~TsdfVolume() {}

void TsdfVolume::create(const Vec3i& dims) {  // :32-38
    dims_ = dims;
    const size_t voxels = (size_t)dims_[0] * dims_[1] * dims_[2];
    data_.create(voxels * sizeof(int));
    setTruncDist(trunc_dist_);
    clear();
}
Vec3i TsdfVolume::getDims() const { return dims_; }
Vec3f TsdfVolume::getVoxelSize() const { return Vec3f(size_[0] / dims_[0], size_[1] / dims_[1], size_[2] / dims_[2]); }
const CudaData TsdfVolume::data() const { return data_; }
CudaData TsdfVolume::data() { return data_; }
Vec3f TsdfVolume::getSize() const { return size_; }
void TsdfVolume::setSize(const Vec3f& size) {
    size_ = size;
    setTruncDist(trunc_dist_);
}
float TsdfVolume::getTruncDist() const { return trunc_dist_; }
void TsdfVolume::setTruncDist(float distance) {  // :57-61
    Vec3f vsz       = getVoxelSize();
    float max_coeff = std::max<float>(std::max<float>(vsz[0], vsz[1]), vsz[2]);
    trunc_dist_     = std::max(distance, 2.1f * max_coeff);
}
int TsdfVolume::getMaxWeight() const { return max_weight_; }
void TsdfVolume::setMaxWeight(int weight) { max_weight_ = weight; }
Affine3f TsdfVolume::getPose() const { return pose_; }
void TsdfVolume::setPose(const Affine3f& pose) { pose_ = pose; }
float TsdfVolume::getRaycastStepFactor() const { return raycast_step_factor_; }
void TsdfVolume::setRaycastStepFactor(float factor) { raycast_step_factor_ = factor; }
float TsdfVolume::getGradientDeltaFactor() const { return gradient_delta_factor_; }
void TsdfVolume::setGradientDeltaFactor(float factor) { gradient_delta_factor_ = factor; }
void TsdfVolume::swap(CudaData& data) { data_.swap(data); }
void TsdfVolume::applyAffine(const Affine3f& affine) { pose_ = affine * pose_; }

void TsdfVolume::clear() {  // :74-80
    dfa::check(dfa_tsdf_clear(data_.ptr<uint32_t>(), dims_[0], dims_[1], dims_[2], nullptr), "TsdfVolume::clear");
}

void TsdfVolume::integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr) {  // :82-93
    Affine3f vol2cam = camera_pose.inv() * pose_;
    float aff[12];
    vol2cam.to12(aff);
    const Vec3f vsz = getVoxelSize();
    dfa::check(dfa_tsdf_integrate(dists.ptr(), (int)dists.step(), dists.cols(), dists.rows(), data_.ptr<uint32_t>(),
                                  dims_[0], dims_[1], dims_[2], vsz.v, trunc_dist_, max_weight_, aff, intr.fx, intr.fy,
                                  intr.cx, intr.cy, nullptr),
               "TsdfVolume::integrate");
    dfa::device_synchronize();  // the reference's device::integrate blocks (tsdf_volume.cu:120)
}

void TsdfVolume::clearAndIntegrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr) {
    Affine3f vol2cam = camera_pose.inv() * pose_;
    float aff[12];
    vol2cam.to12(aff);
    const Vec3f vsz = getVoxelSize();
    dfa::check(dfa_tsdf_clear_integrate(dists.ptr(), (int)dists.step(), dists.cols(), dists.rows(),
                                        data_.ptr<uint32_t>(), dims_[0], dims_[1], dims_[2], vsz.v, trunc_dist_,
                                        max_weight_, aff, intr.fx, intr.fy, intr.cx, intr.cy, nullptr),
               "TsdfVolume::clearAndIntegrate");
    dfa::device_synchronize();
}

void TsdfVolume::raycast(const Affine3f& camera_pose, const Intr& intr, Depth& depth, Normals& normals) {  // :95-110
    Affine3f cam2vol = pose_.inv() * camera_pose;
    float aff[12], rinv[9];
    cam2vol.to12(aff);
    cam2vol.inverse_rotation(rinv);
    const Vec3f vsz = getVoxelSize();
    dfa::check(dfa_tsdf_raycast_depth(data_.ptr<uint32_t>(), dims_[0], dims_[1], dims_[2], vsz.v, trunc_dist_, aff, rinv,
                                      intr.fx, intr.fy, intr.cx, intr.cy, raycast_step_factor_, gradient_delta_factor_,
                                      depth.ptr(), (int)depth.step(), (float*)normals.ptr(), (int)normals.step(),
                                      depth.cols(), depth.rows(), nullptr),
               "TsdfVolume::raycast(depth)");
}

void TsdfVolume::raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals) {  // :112-129
    Affine3f cam2vol = pose_.inv() * camera_pose;
    float aff[12], rinv[9];
    cam2vol.to12(aff);
    cam2vol.inverse_rotation(rinv);
    const Vec3f vsz = getVoxelSize();
    dfa::check(dfa_tsdf_raycast_points(data_.ptr<uint32_t>(), dims_[0], dims_[1], dims_[2], vsz.v, trunc_dist_, aff,
                                       rinv, intr.fx, intr.fy, intr.cx, intr.cy, raycast_step_factor_,
                                       gradient_delta_factor_, (float*)points.ptr(), (int)points.step(),
                                       (float*)normals.ptr(), (int)normals.step(), points.cols(), points.rows(),
                                       nullptr),
               "TsdfVolume::raycast(points)");
}
