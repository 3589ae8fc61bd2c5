// kinfu.cpp — kfusion::KinFu on the adaptor functions (reference: src/kfusion/kinfu.cpp:46-262).
#include <kfusion/kinfu.hpp>

#include <cmath>

#include "../../../include/dynfu_amd.h"

namespace kfusion {

KinFu::KinFu(const KinFuParams& params) : params_(params) {
    if (params.volume_dims[0] % 32 != 0) throw dfa::Error(DFA_ERR_INVALID, "KinFu: volume_dims[0] must be a multiple of 32");  // :47
    volume_ = std::make_shared<cuda::TsdfVolume>(params_.volume_dims);  // :49-56
    volume_->setTruncDist(params_.tsdf_trunc_dist);
    volume_->setMaxWeight(params_.tsdf_max_weight);
    volume_->setSize(params_.volume_size);
    volume_->setPose(params_.volume_pose);
    volume_->setRaycastStepFactor(params_.raycast_step_factor);
    volume_->setGradientDeltaFactor(params_.gradient_delta_factor);
    icp_ = std::make_shared<cuda::ProjectiveICP>();  // :58-61
    icp_->setDistThreshold(params_.icp_dist_thres);
    icp_->setAngleThreshold(params_.icp_angle_thres);
    icp_->setIterationsNum(params_.icp_iter_num);
    mc_ = std::make_shared<cuda::MarchingCubes>();  // :63
    const int levels = cuda::ProjectiveICP::MAX_PYRAMID_LEVELS;  // allocate_buffers, :87-115 (images size themselves)
    for (Frame* f : {&curr_, &prev_}) f->depth_pyr.resize(levels), f->points_pyr.resize(levels), f->normals_pyr.resize(levels);
    reset();
}

void KinFu::reset() {  // :117-126
    frame_counter_ = 0;
    poses_.clear();
    poses_.push_back(Affine3f());
    volume_->clear();
}

Affine3f KinFu::getCameraPose(int time) const {  // :128-134
    // (the reference tests `time > size()`, which lets time == size() read one pose past the end: not reproduced)
    if (time >= (int)poses_.size() || time < 0) time = (int)poses_.size() - 1;
    return poses_[(size_t)time];
}

bool KinFu::operator()(const cuda::Depth& depth) {
    const KinFuParams& p = params_;
    const int LEVELS     = icp_->getUsedLevelsNum();
    cuda::computeDists(depth, dists_, p.intr);  // :144
    cuda::depthBilateralFilter(depth, curr_.depth_pyr[0], p.bilateral_kernel_size, p.bilateral_sigma_spatial,
                               p.bilateral_sigma_depth);  // :145-146
    if (p.icp_truncate_depth_dist > 0) cuda::depthTruncation(curr_.depth_pyr[0], p.icp_truncate_depth_dist);  // :148-149
    for (int i = 1; i < LEVELS; ++i) cuda::depthBuildPyramid(curr_.depth_pyr[i - 1], curr_.depth_pyr[i], p.bilateral_sigma_depth);  // :151-152
    for (int i = 0; i < LEVELS; ++i) cuda::computePointNormals(p.intr(i), curr_.depth_pyr[i], curr_.points_pyr[i], curr_.normals_pyr[i]);  // :154-159
    cuda::waitAllDefaultStream();  // :161

    if (frame_counter_ == 0) {  // :164-174: can't do more with the first frame
        volume_->integrate(dists_, poses_.back(), p.intr);
        curr_.points_pyr.swap(prev_.points_pyr);
        curr_.normals_pyr.swap(prev_.normals_pyr);
        return ++frame_counter_, false;
    }

    Affine3f affine;  // current -> previous (:179-194)
    if (!icp_->estimateTransform(affine, p.intr, curr_.points_pyr, curr_.normals_pyr, prev_.points_pyr, prev_.normals_pyr))
        return reset(), false;
    poses_.push_back(poses_.back() * affine);  // curr -> global (:196)

    // :202-210 — the reference computes whether the camera moved (tsdf_min_camera_movement) and then integrates
    // regardless (the `if (integrate)` is commented out), into a volume it clears first: the model is the last frame
    volume_->clearAndIntegrate(dists_, poses_.back(), p.intr);

    volume_->raycast(poses_.back(), p.intr, prev_.points_pyr[0], prev_.normals_pyr[0]);  // :222
    for (int i = 1; i < LEVELS; ++i)
        cuda::resizePointsNormals(prev_.points_pyr[i - 1], prev_.normals_pyr[i - 1], prev_.points_pyr[i], prev_.normals_pyr[i]);  // :223-225
    cuda::waitAllDefaultStream();  // :227

    if (frame_counter_ == 1) return ++frame_counter_, false;  // :230-232
    return ++frame_counter_, true;
}

std::shared_ptr<dfa::PolygonMesh> KinFu::extractMesh() {
    dfa::DeviceArray<cuda::MarchingCubes::PointType> buffer;
    auto triangles = mc_->run(*volume_, buffer);
    std::vector<cuda::MarchingCubes::PointType> host;
    if (!triangles.empty()) triangles.download(host);
    return std::make_shared<dfa::PolygonMesh>(dfa::convertToMesh(host));
}

}  // namespace kfusion
