// dynfu/utils/frame.hpp — dynfu::Frame (reference: include/dynfu/utils/frame.hpp:15-33): an id plus the vertex
// and normal clouds of one surface (canonical, live, or canonical warped to live).
//
// The reference holds the two pcl clouds by value on the host and every stage that needs them on the GPU stages and
// uploads them again (resetGPUMemory, opt_solver.cpp:149-202).  Here a Frame owns BOTH representations and keeps them
// coherent lazily:
//   * built from host clouds (the reference's constructor): the host clouds are the master; the packed N x 3 device
//     arrays are made on first use and made again whenever the host side may have changed;
//   * built from device arrays (Frame::fromDevice — what the adaptor's own stages produce: marching cubes, warpToLive,
//     findCorrespondingFrame): the device arrays are the master and nothing crosses PCIe until somebody asks for the
//     clouds.  getVertices() / getNormals() return MUTABLE references as the reference does, so from the first such call
//     on the host clouds are the master (the caller may write through the reference at any later time); the const
//     accessors vertices() / normals() download without giving write access and leave the device arrays valid.
#pragma once
#include <cstddef>
#include <memory>
#include <utility>

#include <dfa_host/device.hpp>
#include <dfa_host/types.hpp>

namespace dynfu {

class Frame {
    typedef dfa::PointCloud<dfa::PointXYZ> Vertices;
    typedef dfa::PointCloud<dfa::Normal> Normals;

public:
    Frame(int frame_id, Vertices vertices, Normals normals)
        : id_(frame_id), n_points_(vertices.size()), v_(std::move(vertices)), n_(std::move(normals)), host_valid_(true),
          host_master_(true), dev_valid_(false) {}

    // device-resident frame: vertices3 / normals3 are packed n x 3 float arrays (shared, not copied)
    static std::shared_ptr<Frame> fromDevice(int frame_id, dfa::DeviceArray<float> vertices3,
                                             dfa::DeviceArray<float> normals3, size_t n);

    int getId() { return id_; }
    Vertices& getVertices() { return exposeHost(), v_; }  // mutable reference, as the reference returns
    Normals& getNormals() { return exposeHost(), n_; }

    // ---- extensions of the adaptor
    size_t size() const { return host_master_ ? v_.size() : n_points_; }  // vertices, without materialising either side
    const Vertices& vertices() const { return syncHost(), v_; }
    const Normals& normals() const { return syncHost(), n_; }
    // packed n x 3 device arrays (a vertex without a normal gets (0, 0, 0)); the view is valid while the frame lives
    // and, for a host-master frame, until its next device() call (which uploads the clouds again)
    struct DeviceView {
        const float *vertices, *normals;
        size_t n;
    };
    DeviceView device() const { return syncDevice(), DeviceView{dv_.ptr(), dn_.ptr(), size()}; }
    // the arrays themselves (shared ownership: they outlive the frame for as long as the caller keeps them)
    void deviceArrays(dfa::DeviceArray<float>& vertices3, dfa::DeviceArray<float>& normals3) const {
        syncDevice();
        vertices3 = dv_, normals3 = dn_;
    }
    bool deviceResident() const { return !host_master_; }

private:
    Frame() = default;
    int id_ = 0;
    size_t n_points_ = 0;
    mutable Vertices v_;
    mutable Normals n_;
    mutable dfa::DeviceArray<float> dv_, dn_;
    mutable bool host_valid_ = false, host_master_ = false, dev_valid_ = false;
    void exposeHost() {
        syncHost();
        host_master_ = true;
    }
    void syncHost() const;    // device -> host clouds when the host side is not valid
    void syncDevice() const;  // host clouds -> device arrays: always for a host-master frame (it may have been written)
};

}  // namespace dynfu
