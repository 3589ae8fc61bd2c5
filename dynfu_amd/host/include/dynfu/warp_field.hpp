// dynfu/warp_field.hpp — class Warpfield with the reference's interface
// (include/dynfu/warp_field.hpp:32-78, src/dynfu/warp_field.cpp).  The k-NN and the bulk warp
// run on the GPU through dfa_knn / dfa_warp_to_live; the reference's nanoflann KD-tree is gone
// (getKdTree() has no counterpart).  Node insertion (update / getUnsupportedVertices,
// warp_field.cpp:34-95) is a "next" row of the scope table and not provided.
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

    std::vector<std::shared_ptr<Node>> findNeighbors(int numNeighbor, dfa::PointXYZ vertex);
    std::vector<size_t> findNeighborsIndex(int numNeighbor, dfa::PointXYZ vertex);

    std::shared_ptr<DualQuaternion<float>> calcDQB(dfa::PointXYZ point);
    std::shared_ptr<dynfu::Frame> warpToLive(std::shared_ptr<dynfu::Frame> canonicalFrame);

    // number of neighbours (the reference's compile-time KNN; run-time here, default 8)
    void setKnn(int k) { knn_ = k; }
    int getKnn() const { return knn_; }

private:
    float epsilon = 0.f;
    int knn_      = KNN;
    std::vector<std::shared_ptr<Node>> nodes;
    struct DeviceNodes;  // device copies of node positions / weights (positions are immutable)
    std::shared_ptr<DeviceNodes> dev;
    void syncPositions();
};
