// dynfu/utils/node.hpp — deformation node (include/dynfu/utils/node.hpp:33-59, src/dynfu/utils/node.cpp)
#pragma once
#include <cmath>
#include <memory>

#include <dynfu/utils/dual_quaternion.hpp>

class Node {
public:
    Node(dfa::PointXYZ position, std::shared_ptr<DualQuaternion<float>> transformation, float radialBasisWeight)
        : dg_v(position), dg_se3(transformation), dg_w(radialBasisWeight) {}
    ~Node() = default;

    dfa::PointXYZ getPosition() { return dg_v; }
    std::shared_ptr<DualQuaternion<float>>& getTransformation() { return dg_se3; }
    // dg_se3 <- new * dg_se3 (node.cpp:19-23)
    void updateTransformation(std::shared_ptr<DualQuaternion<float>> new_dg_se3) {
        dg_se3 = std::make_shared<DualQuaternion<float>>(*new_dg_se3 * *dg_se3);
    }
    // the same composition from a value (one allocation per call instead of two: a solve updates every node)
    void updateTransformation(const DualQuaternion<float>& new_dg_se3) {
        dg_se3 = std::make_shared<DualQuaternion<float>>(new_dg_se3 * *dg_se3);
    }
    void setTransformation(std::shared_ptr<DualQuaternion<float>> new_dg_se3) { dg_se3 = new_dg_se3; }
    float getRadialBasisWeight() { return dg_w; }
    // exp(-|dg_v - v|^2 / (2 dg_w^2)), double arithmetic (node.cpp:29-36)
    float getTransformationWeight(dfa::PointXYZ v) {
        const double dx = (double)(dg_v.x - v.x), dy = (double)(dg_v.y - v.y), dz = (double)(dg_v.z - v.z);
        return (float)std::exp(-(dx * dx + dy * dy + dz * dz) / (2 * ((double)dg_w * (double)dg_w)));
    }

private:
    dfa::PointXYZ dg_v;
    std::shared_ptr<DualQuaternion<float>> dg_se3;
    float dg_w;
};
