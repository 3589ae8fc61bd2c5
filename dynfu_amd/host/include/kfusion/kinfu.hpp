// kfusion/kinfu.hpp — class kfusion::KinFu, the rigid KinectFusion loop the reference's DynFusion derives from
// (include/kfusion/kinfu.hpp:24-108, src/kfusion/kinfu.cpp:46-262): per frame
//   dists -> bilateral filter -> [truncation] -> depth pyramid -> point / normal maps per level
//   -> projective ICP against the previous frame's model maps (coarse to fine, kinfu.cpp:185-200)
//   -> pose chain -> clear + integrate -> raycast the model from the new pose -> down-sampled model maps.
// Every device step is one call of the dynfu_amd C ABI through the adaptor functions of kfusion/cuda/*.hpp; the class
// itself is host control flow.  Built is the reference's default compile path (point maps; `USE_DEPTH` undefined).
// Not here: renderImage / the light pose (visualisation) and the colour image argument (never read, kinfu.cpp:140).
#pragma once
#include <memory>
#include <vector>

#include <dfa_host/io.hpp>
#include <kfusion/cuda/imgproc.hpp>
#include <kfusion/cuda/marching_cubes.hpp>
#include <kfusion/cuda/projective_icp.hpp>
#include <kfusion/cuda/tsdf_volume.hpp>

namespace kfusion {

// kfusion::KinFuParams (include/kfusion/kinfu.hpp:33-61, defaults src/kfusion/kinfu.cpp:10-44) without the light pose
// (rendering)
struct KinFuParams {
    static KinFuParams default_params();
    int cols, rows;
    Intr intr;
    Vec3i volume_dims;
    Vec3f volume_size;
    Affine3f volume_pose;
    float bilateral_sigma_depth, bilateral_sigma_spatial;
    int bilateral_kernel_size;
    float icp_truncate_depth_dist;
    float icp_dist_thres, icp_angle_thres;  // gates of the rigid tracker (kfusion::KinFu)
    std::vector<int> icp_iter_num;          // iterations per pyramid level, level 0 = full resolution
    float tsdf_min_camera_movement;
    float tsdf_trunc_dist;
    int tsdf_max_weight;
    float raycast_step_factor, gradient_delta_factor;
};

class KinFu {
public:
    explicit KinFu(const KinFuParams& params);  // kinfu.cpp:46-67 (volume dims must be a multiple of 32, :47)

    const KinFuParams& params() const { return params_; }
    KinFuParams& params() { return params_; }
    cuda::TsdfVolume& tsdf() { return *volume_; }
    cuda::ProjectiveICP& icp() { return *icp_; }
    cuda::MarchingCubes& mc() { return *mc_; }

    void reset();                              // :117-126: pose chain back to the identity, volume cleared
    Affine3f getCameraPose(int time = -1) const;  // :128-134
    int frameCounter() const { return frame_counter_; }

    // kinfu.cpp:140-234.  false for the first two frames and after a lost track (which resets), true afterwards.
    bool operator()(const cuda::Depth& depth);

    // marching cubes of the current volume in KinFu::convertToMesh's layout (:236-262)
    std::shared_ptr<dfa::PolygonMesh> extractMesh();

protected:  // as in the reference (kinfu.hpp:88-108): DynFusion derives from this class and drives these directly
    struct Frame {
        std::vector<cuda::Depth> depth_pyr;
        std::vector<cuda::Cloud> points_pyr;
        std::vector<cuda::Normals> normals_pyr;
    };
    int frame_counter_ = 0;
    KinFuParams params_;
    std::vector<Affine3f> poses_;
    cuda::Dists dists_;
    Frame curr_, prev_;
    std::shared_ptr<cuda::TsdfVolume> volume_;
    std::shared_ptr<cuda::ProjectiveICP> icp_;
    std::shared_ptr<cuda::MarchingCubes> mc_;
};

}  // namespace kfusion
