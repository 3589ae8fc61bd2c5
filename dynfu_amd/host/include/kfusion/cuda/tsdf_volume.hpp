// kfusion/cuda/tsdf_volume.hpp — kfusion::cuda::TsdfVolume: the public interface of the reference's class
// (include/kfusion/cuda/tsdf_volume.hpp:7-73; behaviour src/kfusion/tsdf_volume.cpp:18-129) on the dynfu_amd C ABI.
//
// Layout of this adaptor: the volume's settings live in one plain struct, trivial accessors are inline, the
// three operations that touch the GPU (clear / integrate / raycast) are one dfa_tsdf_* call each.
// Not provided: fetchCloud / fetchNormals / get,setGridOrigin (not on the DynFusion path).
#pragma once
#include <algorithm>

#include <kfusion/types.hpp>

namespace kfusion {
namespace cuda {

class TsdfVolume {
    struct Settings {
        Vec3i dims;
        Vec3f size          = Vec3f::all(3.f);      // metres            (tsdf_volume.cpp:22)
        Affine3f pose       = Affine3f::Identity(); // volume -> world   (:23)
        float trunc         = 0.03f;                // metres            (:19), clamped by setTruncDist
        int max_weight      = 128;                  //                   (:20)
        float ray_step      = 0.75f;                // in truncation distances (:25)
        float grad_delta    = 0.75f;                // in voxels         (:24)
    } cfg_;
    CudaData blob_;  // X*Y*Z packed voxels {fp16 tsdf, u16 weight}
    // Occupancy map of the volume (dynfu_amd.h: dfa_tsdf_occupancy_bytes): which boxes of 32 x 2 x 8 voxels the sweeps of THIS
    // object may have left a weight in — kept by clear / integrate / clearAndIntegrate, read by cuda::MarchingCubes::run
    // instead of the empty voxels.  The map is only trusted while THIS object is the sole owner of the voxels and of the
    // map: CudaData handles and copies of a TsdfVolume share storage (ref-counted, as the reference's DeviceMemory), so a
    // writable data() handle, a swap, a copy of the object (either side may write afterwards) or a sweep run while another
    // handle is alive all make it unknown until the next clear / fused sweep by a sole owner.
    CudaData occ_;
    mutable bool occ_known_ = false;  // (mutable: copying a volume makes the SOURCE's map unknown too)
    bool soleOwner() const { return blob_.unique() && occ_.unique(); }
    bool mapTrusted() const { return occ_known_ && soleOwner(); }

public:
    // --- construction / storage -----------------------------------------------------------------------------
    explicit TsdfVolume(const Vec3i& dims) { cfg_.dims = dims, create(dims); }
    TsdfVolume(const TsdfVolume& o) : cfg_(o.cfg_), blob_(o.blob_), occ_(o.occ_) { o.occ_known_ = false; }
    TsdfVolume& operator=(const TsdfVolume& o) {
        if (this != &o) cfg_ = o.cfg_, blob_ = o.blob_, occ_ = o.occ_, occ_known_ = false, o.occ_known_ = false;
        return *this;
    }
    virtual ~TsdfVolume() {}
    void create(const Vec3i& dims);            // allocates and clears
    void swap(CudaData& data) { blob_.swap(data), occ_known_ = false; }
    CudaData data() { return occ_known_ = false, blob_; }
    const CudaData data() const { return blob_; }
    // the occupancy map of the voxels (device) or nullptr when the voxels may have been written from outside this object
    const unsigned char* occupancy() const { return mapTrusted() ? occ_.ptr<unsigned char>() : nullptr; }

    // --- the GPU work -----------------------------------------------------------------------------------------
    virtual void clear();
    virtual void integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr);
    // clear() then integrate() as ONE sweep (what DynFusion::operator() does every frame, dyn_fusion.cpp:113-116):
    // same voxels bit for bit, half the HBM traffic.  Extension of the reference's interface.
    void clearAndIntegrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr);
    virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Depth& depth, Normals& normals);
    virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals);
    virtual void applyAffine(const Affine3f& affine) { cfg_.pose = affine * cfg_.pose; }

    // --- settings ---------------------------------------------------------------------------------------------
    Vec3i getDims() const { return cfg_.dims; }
    Vec3f getSize() const { return cfg_.size; }
    Vec3f getVoxelSize() const {
        return Vec3f(cfg_.size[0] / cfg_.dims[0], cfg_.size[1] / cfg_.dims[1], cfg_.size[2] / cfg_.dims[2]);
    }
    Affine3f getPose() const { return cfg_.pose; }
    float getTruncDist() const { return cfg_.trunc; }
    int getMaxWeight() const { return cfg_.max_weight; }
    float getRaycastStepFactor() const { return cfg_.ray_step; }
    float getGradientDeltaFactor() const { return cfg_.grad_delta; }

    void setPose(const Affine3f& pose) { cfg_.pose = pose; }
    void setMaxWeight(int weight) { cfg_.max_weight = weight; }
    void setRaycastStepFactor(float factor) { cfg_.ray_step = factor; }
    void setGradientDeltaFactor(float factor) { cfg_.grad_delta = factor; }
    // never below 2.1 voxel edges (tsdf_volume.cpp:57-61); re-applied when the size changes (:47-50)
    void setTruncDist(float distance) {
        const Vec3f v = getVoxelSize();
        cfg_.trunc    = std::max(distance, 2.1f * std::max(std::max(v[0], v[1]), v[2]));
    }
    void setSize(const Vec3f& size) { cfg_.size = size, setTruncDist(cfg_.trunc); }

    // the packed voxel (tsdf_volume.hpp:53-62); the reference's two converters throw "Not implemented"
    struct Entry {
        typedef unsigned short half;
        half tsdf;
        unsigned short weight;
        static half float2half(float value);
        static float half2float(half value);
    };
};

}  // namespace cuda
}  // namespace kfusion
