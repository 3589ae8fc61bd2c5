// dynfu/utils/frame.hpp — dynfu::Frame (reference: include/dynfu/utils/frame.hpp:15-33): an id plus the vertex
// and normal clouds of one surface (canonical, live, or canonical warped to live), held by value.
#pragma once
#include <utility>

#include <dfa_host/types.hpp>

namespace dynfu {

class Frame {
    typedef dfa::PointCloud<dfa::PointXYZ> Vertices;
    typedef dfa::PointCloud<dfa::Normal> Normals;
    int id_;
    Vertices v_;
    Normals n_;

public:
    Frame(int frame_id, Vertices vertices, Normals normals) : id_(frame_id), v_(std::move(vertices)), n_(std::move(normals)) {}

    int getId() { return id_; }
    Vertices& getVertices() { return v_; }  // mutable reference, as the reference returns
    Normals& getNormals() { return n_; }
};

}  // namespace dynfu
