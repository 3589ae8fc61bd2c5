// kfusion/cuda/projective_icp.hpp — kfusion::cuda::ProjectiveICP (reference interface:
// include/kfusion/cuda/projective_icp.hpp:7-45; behaviour: src/kfusion/projective_icp.cpp).
//
// Split of the work: every iteration's per-pixel rows and their 27 sums are ONE dfa_icp_sums call (GPU); the
// 6x6 system, its determinant test and the pose update are host code, as in the reference.  OpenCV is not
// available to this build: a pivoted LU in double stands in for cv::determinant + cv::solve(DECOMP_SVD), the
// Rodrigues formula for cv::Affine3f(rvec, t).  The Frame overload of estimateTransform is a CV_Assert(false) in the
// reference (:98-113) and is not declared here.
#pragma once
#include <vector>

#include <kfusion/types.hpp>

namespace kfusion {
namespace cuda {

class ProjectiveICP {
public:
    enum { MAX_PYRAMID_LEVELS = 4 };
    typedef std::vector<Depth> DepthPyr;      // masked depth per level (depth != 0 => normal defined)
    typedef std::vector<Normals> NormalsPyr;
    typedef std::vector<Cloud> PointsPyr;

    ProjectiveICP();  // 20 degrees, 0.1 m, iterations {10, 5, 4, 0} (:62-65)
    virtual ~ProjectiveICP();

    // coarse-to-fine Gauss-Newton on the point-to-plane error; false when a level's system is singular (:136-142)
    virtual bool estimateTransform(Affine3f& affine, const Intr& intr, const DepthPyr& dcurr, const NormalsPyr ncurr,
                                   const DepthPyr dprev, const NormalsPyr nprev);
    virtual bool estimateTransform(Affine3f& affine, const Intr& intr, const PointsPyr& vcurr, const NormalsPyr ncurr,
                                   const PointsPyr vprev, const NormalsPyr nprev);

    // gates and schedule
    void setIterationsNum(const std::vector<int>& iters);  // per level, level 0 = full resolution
    int getUsedLevelsNum() const;                          // levels up to the last non-zero iteration count
    void setAngleThreshold(float angle) { gate_angle_ = angle; }
    void setDistThreshold(float distance) { gate_dist_ = distance; }
    float getAngleThreshold() const { return gate_angle_; }
    float getDistThreshold() const { return gate_dist_; }

private:
    float gate_angle_, gate_dist_;
    std::vector<int> schedule_;
    dfa::DeviceArray<float> sums_;  // the 27 sums of one linearisation (device)
    bool runLevel(Affine3f& affine, const Intr& intr, int level, bool depth_variant, const void* curr, int curr_step,
                  const float* ncurr, int ncurr_step, const void* prev, int prev_step, const float* nprev, int nprev_step,
                  int cols, int rows);
};

}  // namespace cuda
}  // namespace kfusion
