// dynfu/warp_field.hpp — class Warpfield with the reference's interface
// (include/dynfu/warp_field.hpp:32-78, src/dynfu/warp_field.cpp).  The k-NN and the bulk warp
// run on the GPU through dfa_knn / dfa_warp_to_live; the reference's nanoflann KD-tree is gone
// (getKdTree() has no counterpart).  Node insertion (getUnsupportedVertices / update, warp_field.cpp:34-95): the
// support test runs on the GPU (dfa_unsupported_vertices), the pcl::VoxelGrid subsampling of the few unsupported
// vertices is dfa::voxelGridFilter (host C++, restating PCL's algorithm), the new nodes' transforms come from
// dfa_calc_dqb.
#pragma once
#include <memory>
#include <vector>

#include <dynfu/utils/frame.hpp>
#include <dynfu/utils/node.hpp>

#define KNN 8  // warp_field.hpp:27

class Warpfield {
public:
    Warpfield();
    ~Warpfield();

    void init(float epsilon, std::vector<std::shared_ptr<Node>> nodes);
    void addNode(std::shared_ptr<Node> newNode);
    std::vector<std::shared_ptr<Node>> getNodes();
    // the same nodes without the copy (8 k shared pointers: ~0.05 ms of reference counting per call) — for the adaptor's own
    // per-frame loops; the reference's accessor above returns by value
    const std::vector<std::shared_ptr<Node>>& nodesRef() const { return *nodes; }

    std::vector<std::shared_ptr<Node>> findNeighbors(int numNeighbor, dfa::PointXYZ vertex);
    std::vector<size_t> findNeighborsIndex(int numNeighbor, dfa::PointXYZ vertex);

    std::shared_ptr<DualQuaternion<float>> calcDQB(dfa::PointXYZ point);
    std::shared_ptr<dynfu::Frame> warpToLive(std::shared_ptr<dynfu::Frame> canonicalFrame);

    // warp_field.cpp:34-62 (returns the cloud by value instead of a pcl Ptr)
    dfa::PointCloud<dfa::PointXYZ> getUnsupportedVertices(std::shared_ptr<dynfu::Frame> frame);
    // warp_field.cpp:64-95: insert a node per 5 cm voxel of unsupported vertices
    void update(std::shared_ptr<dynfu::Frame> frame);

    // number of neighbours (the reference's compile-time KNN; run-time here, default 8)
    void setKnn(int k) { knn_ = k; }
    int getKnn() const { return knn_; }

private:
    float epsilon = 0.f;
    int knn_      = KNN;
    // The node list is shared by the copies of a warp field until one of them changes it (copy on write): the reference
    // hands Warpfield around BY VALUE every frame (CombinedSolver's constructor, opt_solver.cpp:3-13), and a copy of 8.5 k
    // shared pointers — and their release when the solver goes — was 0.1 ms of each frame.  Each copy still sees its own
    // list, the Nodes themselves were always shared.
    using NodeList = std::vector<std::shared_ptr<Node>>;
    std::shared_ptr<NodeList> nodes = std::make_shared<NodeList>();
    NodeList& ownNodes() {  // before any change of the list
        if (nodes.use_count() > 1) nodes = std::make_shared<NodeList>(*nodes);
        return *nodes;
    }
    struct DeviceNodes;  // device copies of node positions / weights (positions are immutable)
    std::shared_ptr<DeviceNodes> dev;
    void syncPositions();
    void syncTransforms();

public:
    // device copies of the node arrays for the adaptor's own stages (D x 3 positions, D weights, D x 8 transforms as
    // of the last syncTransforms); nullptr / 0 before init
    struct DeviceNodeView {
        const float *pos, *w, *dq;
        int D;
    };
    DeviceNodeView deviceNodes(bool refresh_transforms = true);
    // the same arrays on the host: D x 3 positions, D weights, D x 8 transforms (real w,x,y,z ; dual w,x,y,z) — what
    // the solver adaptors upload (resetGPUMemory, opt_solver.cpp:149-202)
    void hostArrays(std::vector<float>& pos, std::vector<float>& w, std::vector<float>& dq);
};
