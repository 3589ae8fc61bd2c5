// frame.cpp — the two representations of a dynfu::Frame and the traffic between them (see dynfu/utils/frame.hpp).
// Host clouds are arrays of 16-byte points (pcl::PointXYZ / pcl::Normal layout), the seams take packed N x 3 floats:
// the 16-byte records cross PCIe as they are and dfa_repack_points converts on the device — no per-point host loops.
#include <dynfu/utils/frame.hpp>

#include "../../../include/dynfu_amd.h"

namespace dynfu {

static_assert(sizeof(dfa::PointXYZ) == 16 && sizeof(dfa::Normal) == 16, "clouds are arrays of float4 records");

std::shared_ptr<Frame> Frame::fromDevice(int frame_id, dfa::DeviceArray<float> vertices3, dfa::DeviceArray<float> normals3,
                                         size_t n) {
    if (vertices3.size() < 3 * n || normals3.size() < 3 * n)
        throw dfa::Error(DFA_ERR_INVALID, "Frame::fromDevice: arrays shorter than n x 3");
    std::shared_ptr<Frame> f(new Frame());
    f->id_ = frame_id, f->n_points_ = n;
    f->dv_ = vertices3, f->dn_ = normals3;
    f->dev_valid_ = true;
    return f;
}

void Frame::syncHost() const {
    if (host_valid_) return;
    const size_t n = n_points_;
    v_.points.resize(n), n_.points.resize(n);
    if (n) {
        dfa::DeviceArray<float> staging(4 * n);
        dfa::check(dfa_repack_points(dv_.ptr(), 3, staging.ptr(), 4, (int)n, 1.f, nullptr), "Frame: vertices to host");
        staging.DeviceMemory::download(v_.points.data(), 16 * n);  // hipMemcpy: ordered behind the kernel
        dfa::check(dfa_repack_points(dn_.ptr(), 3, staging.ptr(), 4, (int)n, 0.f, nullptr), "Frame: normals to host");
        staging.DeviceMemory::download(n_.points.data(), 16 * n);
    }
    host_valid_ = true;
}

void Frame::syncDevice() const {
    if (dev_valid_ && !host_master_) return;
    // a host-master frame is uploaded on every use: its clouds may have been written through getVertices()
    // (fresh arrays every time: arrays handed out earlier stay the snapshot they were)
    const size_t n = v_.size();
    dv_ = dfa::DeviceArray<float>(3 * n), dn_ = dfa::DeviceArray<float>(3 * n);
    if (n) {
        dfa::DeviceArray<float> staging(4 * n);
        staging.DeviceMemory::upload(v_.points.data(), 16 * n);
        dfa::check(dfa_repack_points(staging.ptr(), 4, dv_.ptr(), 3, (int)n, 0.f, nullptr), "Frame: vertices to device");
        if (n_.size() >= n) {
            staging.DeviceMemory::upload(n_.points.data(), 16 * n);
        } else {  // fewer normals than vertices: the missing ones are default-constructed (0, 0, 0)
            std::vector<dfa::Normal> padded(n_.points);
            padded.resize(n);
            staging.DeviceMemory::upload(padded.data(), 16 * n);
        }
        dfa::check(dfa_repack_points(staging.ptr(), 4, dn_.ptr(), 3, (int)n, 0.f, nullptr), "Frame: normals to device");
    }
    dev_valid_ = true;
}

}  // namespace dynfu
