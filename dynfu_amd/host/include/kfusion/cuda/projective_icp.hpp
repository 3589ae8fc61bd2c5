// kfusion/cuda/projective_icp.hpp — class kfusion::cuda::ProjectiveICP with the reference's interface
// (include/kfusion/cuda/projective_icp.hpp:7-45, src/kfusion/projective_icp.cpp) on dfa_icp_sums: the
// per-pixel rows and their 27 sums run on the GPU, the 6x6 solve and the pose update on the host as in the
// reference.  cv::determinant / cv::solve(DECOMP_SVD) / cv::Affine3f(rvec, t) (OpenCV is not available to this
// build) are replaced by a pivoted LU in double and the Rodrigues formula.
#pragma once
#include <vector>

#include <kfusion/types.hpp>

namespace kfusion {
namespace cuda {
class ProjectiveICP {
public:
    enum { MAX_PYRAMID_LEVELS = 4 };
    typedef std::vector<Depth> DepthPyr;
    typedef std::vector<Cloud> PointsPyr;
    typedef std::vector<Normals> NormalsPyr;

    ProjectiveICP();
    virtual ~ProjectiveICP();

    float getDistThreshold() const { return dist_thres_; }
    void setDistThreshold(float distance) { dist_thres_ = distance; }
    float getAngleThreshold() const { return angle_thres_; }
    void setAngleThreshold(float angle) { angle_thres_ = angle; }
    void setIterationsNum(const std::vector<int>& iters);
    int getUsedLevelsNum() const;

    // masked depth (depth != 0 implies a defined normal), projective_icp.cpp:118-150
    virtual bool estimateTransform(Affine3f& affine, const Intr& intr, const DepthPyr& dcurr, const NormalsPyr ncurr,
                                   const DepthPyr dprev, const NormalsPyr nprev);
    // vertex maps, projective_icp.cpp:152-200
    virtual bool estimateTransform(Affine3f& affine, const Intr& intr, const PointsPyr& vcurr, const NormalsPyr ncurr,
                                   const PointsPyr vprev, const NormalsPyr nprev);

private:
    std::vector<int> iters_;
    float angle_thres_;
    float dist_thres_;
    dfa::DeviceArray<float> sums_;
    bool iterate(Affine3f& affine, const Intr& intr, int level, bool depth_variant, const void* curr, int curr_step,
                 const float* ncurr, int ncurr_step, const void* prev, int prev_step, const float* nprev, int nprev_step,
                 int cols, int rows);
};
}  // namespace cuda
}  // namespace kfusion
