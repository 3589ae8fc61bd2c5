// kfusion/cuda/tsdf_volume.hpp — kfusion::cuda::TsdfVolume with the reference's public interface
// (include/kfusion/cuda/tsdf_volume.hpp:7-73, src/kfusion/tsdf_volume.cpp:18-129), implemented on
// the dynfu_amd C ABI (dfa_tsdf_*).  fetchCloud / fetchNormals are out of the hot path
// (not called by DynFusion) and not provided.
#pragma once
#include <kfusion/types.hpp>

namespace kfusion {
namespace cuda {
class TsdfVolume {
public:
    explicit TsdfVolume(const Vec3i& dims);
    virtual ~TsdfVolume();

    void create(const Vec3i& dims);

    Vec3i getDims() const;
    Vec3f getVoxelSize() const;

    const CudaData data() const;
    CudaData data();

    Vec3f getSize() const;
    void setSize(const Vec3f& size);

    float getTruncDist() const;
    void setTruncDist(float distance);

    int getMaxWeight() const;
    void setMaxWeight(int weight);

    Affine3f getPose() const;
    void setPose(const Affine3f& pose);

    float getRaycastStepFactor() const;
    void setRaycastStepFactor(float factor);

    float getGradientDeltaFactor() const;
    void setGradientDeltaFactor(float factor);

    virtual void clear();
    virtual void applyAffine(const Affine3f& affine);
    virtual void integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr);
    virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Depth& depth, Normals& normals);
    virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals);

    // clear() + integrate() in one sweep: what DynFusion::operator() does every frame
    // (src/dynfu/dyn_fusion.cpp:113-116); bit-identical result, half the HBM traffic.  Extension.
    void clearAndIntegrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr);

    void swap(CudaData& data);

    struct Entry {  // tsdf_volume.hpp:53-62 (the reference's converters throw "Not implemented")
        typedef unsigned short half;
        half tsdf;
        unsigned short weight;
        static float half2float(half value);
        static half float2half(float value);
    };

private:
    CudaData data_;
    float trunc_dist_;
    int max_weight_;
    Vec3i dims_;
    Vec3f size_;
    Affine3f pose_;
    float gradient_delta_factor_;
    float raycast_step_factor_;
};
}  // namespace cuda
}  // namespace kfusion
