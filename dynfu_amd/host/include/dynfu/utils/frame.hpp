// dynfu/utils/frame.hpp — dynfu::Frame (include/dynfu/utils/frame.hpp:15-33)
#pragma once
#include <dfa_host/types.hpp>

namespace dynfu {
class Frame {
public:
    Frame(int id, dfa::PointCloud<dfa::PointXYZ> vertices, dfa::PointCloud<dfa::Normal> normals)
        : id(id), vertices(vertices), normals(normals) {}
    ~Frame() = default;
    int getId() { return id; }
    dfa::PointCloud<dfa::PointXYZ>& getVertices() { return vertices; }
    dfa::PointCloud<dfa::Normal>& getNormals() { return normals; }

private:
    int id;
    dfa::PointCloud<dfa::PointXYZ> vertices;
    dfa::PointCloud<dfa::Normal> normals;
};
}  // namespace dynfu
