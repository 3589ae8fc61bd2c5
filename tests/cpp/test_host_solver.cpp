// test_host_solver.cpp — the reference's 8 OptTest scenes (test/opt_optimisation_test.cpp:212-698)
// through the host adaptor classes Warpfield / CombinedSolver on the GPU.  Fixture values
// :36-122, assertion form :229-239 (|warp(source) - target| <= 1e-3 per axis).
#include <dynfu/utils/opt_solver.hpp>
#include <dynfu/warp_field.hpp>

#include "minitest.hpp"

namespace {
typedef std::vector<std::shared_ptr<Node>> Nodes;
typedef dfa::PointCloud<dfa::PointXYZ> Cloud;

struct Fixture {
    CombinedSolverParameters params;
    Nodes g1, g2, all;
    Warpfield warpfield;
    dfa::Affine3f affine;
    float maxError = 1e-3f, epsilon_dynfu = 0.0015f, tukeyOffset = 4.652f, psi_data = 1e-2f, lambda = 0.f,
          psi_reg = 1e-4f;
    Fixture() {
        params.numIter = 32, params.nonLinearIter = 16, params.linearIter = 256;  // :38-44
        params.useOpt = false, params.useOptLM = true, params.earlyOut = true, params.optDoublePrecision = true;
        auto dq  = std::make_shared<DualQuaternion<float>>(0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
        float w  = 2;
        const float a[8][3]  = {{3, 1, -1}, {1, 1, 1}, {-1, 2, 3}, {-1, -1, 1}, {-2, -1, -1}, {2, -1, -3}, {-1, 1, -1}, {2, 1, 1}};
        const float b[10][3] = {{10, 10, 10}, {9, 11.1f, 10}, {10, 9, 10}, {10, 12, 9}, {9, 11, 10},
                                {12, 10, 9},  {9, 9, 12},    {10.5f, 9, 9}, {10.5f, 12, 12}, {11, 11, 10.9f}};
        // every node owns its own transform object (the reference shares one dg_se3 pointer between
        // all nodes, :51 — harmless there because updateTransformation replaces the pointer)
        for (auto& p : a) g1.push_back(std::make_shared<Node>(dfa::PointXYZ(p[0], p[1], p[2]), dq, w));
        for (auto& p : b) g2.push_back(std::make_shared<Node>(dfa::PointXYZ(p[0], p[1], p[2]), dq, w));
        all = g1;
        all.insert(all.end(), g2.begin(), g2.end());
    }
};

Cloud diag(std::initializer_list<float> xs) {
    Cloud c;
    for (float x : xs) c.push_back(dfa::PointXYZ(x, x, x));
    return c;
}
dfa::PointCloud<dfa::Normal> ones(size_t n) {
    dfa::PointCloud<dfa::Normal> c;
    for (size_t i = 0; i < n; ++i) c.push_back(dfa::Normal(1, 1, 1));
    return c;
}
std::shared_ptr<dynfu::Frame> frame(int id, const Cloud& v) { return std::make_shared<dynfu::Frame>(id, v, ones(v.size())); }

void solve(Fixture& f, std::shared_ptr<dynfu::Frame> canon, std::shared_ptr<dynfu::Frame> live) {
    CombinedSolver s(f.warpfield, f.params, f.tukeyOffset, f.psi_data, f.lambda, f.psi_reg);
    s.initializeProblemInstance(canon, live, f.affine);
    s.solveAll();
}
void expect_warp(Fixture& f, const Cloud& src, const Cloud& expected) {
    for (size_t j = 0; j < src.size(); ++j) {
        auto total  = f.warpfield.calcDQB(src[j]);
        auto result = total->transformVertex(src[j]);
        ASSERT_NEAR(result.x, expected[j].x, f.maxError);
        ASSERT_NEAR(result.y, expected[j].y, f.maxError);
        ASSERT_NEAR(result.z, expected[j].z, f.maxError);
    }
}
void one_step(Fixture& f, const Nodes& nodes, const Cloud& S, const Cloud& T) {
    f.warpfield.init(f.epsilon_dynfu, nodes);
    solve(f, frame(0, S), frame(1, T));
    expect_warp(f, S, T);
}
const Cloud S5 = diag({-3, -2, 0.01f, 2, 3}), T5 = diag({-2.99f, -1.99f, 0.02f, 2.01f, 3.01f});
const Cloud S5b = diag({-3, -2, 0.04f, 2, 3}), T5b1 = diag({-2.99f, -1.99f, 0.05f, 2.01f, 3.01f});
const Cloud T5b2 = diag({-2.98f, -1.98f, 0.06f, 2.02f, 3.02f}), T5b3 = diag({-2.96f, -1.96f, 0.09f, 2.04f, 3.05f});
}  // namespace

TEST(OptTest, SingleVertexOneGroupOfDeformationNodesTest) {  // :212-240
    Fixture f;
    Cloud S, T;
    S.push_back(dfa::PointXYZ(0, 0.04f, 0)), T.push_back(dfa::PointXYZ(0.01f, 0.03f, 0));
    one_step(f, f.g1, S, T);
}
TEST(OptTest, TwoVerticesOneNotMovingOneGroupOfDeformationNodesTest) {  // :243-277
    Fixture f;
    Cloud S, T;
    S.push_back(dfa::PointXYZ(0, 0.05f, 1)), S.push_back(dfa::PointXYZ(2, 2, 2));
    T.push_back(dfa::PointXYZ(0.01f, 0.04f, 1.01f)), T.push_back(dfa::PointXYZ(2, 2, 2));
    one_step(f, f.all, S, T);
}
TEST(OptTest, MultipleVerticesOneGroupOfDeformationNodesTest) {  // :280-326
    Fixture f;
    one_step(f, f.g1, S5, T5);
}
TEST(OptTest, OneGroupOfVerticesTwoGroupsOfDeformationNodes) {  // :329-375
    Fixture f;
    one_step(f, f.all, S5, T5);
}
TEST(OptTest, TwoGroupsOfVerticesTwoGroupsOfDeformationNodes) {  // :378-451
    Fixture f;
    Cloud S = S5, T = T5;
    for (float x : {12.f, 11.f, 10.f, 10.5f, 11.5f}) S.push_back(dfa::PointXYZ(x, x, x));
    for (float x : {11.99f, 10.99f, 9.99f, 10.51f, 11.49f}) T.push_back(dfa::PointXYZ(x, x, x));
    one_step(f, f.all, S, T);
}
TEST(OptTest, MultipleVerticesOneGroupOfDeformationNodesWarpTwiceTest) {  // :454-527
    Fixture f;
    f.warpfield.init(f.epsilon_dynfu, f.g1);
    auto canonical = frame(0, S5b);
    solve(f, canonical, frame(1, T5b1));
    expect_warp(f, S5b, T5b1);
    auto warped = f.warpfield.warpToLive(canonical);
    solve(f, warped, frame(1, T5b2));
    expect_warp(f, S5b, T5b2);
}
TEST(OptTest, MultipleVerticesOneGroupOfDeformationNodesWarpThriceTest) {  // :530-629
    Fixture f;
    f.warpfield.init(f.epsilon_dynfu, f.g1);
    auto canonical = frame(0, S5b);
    solve(f, canonical, frame(1, T5b1));
    expect_warp(f, S5b, T5b1);
    auto warped = f.warpfield.warpToLive(canonical);
    solve(f, warped, frame(1, T5b2));
    expect_warp(f, S5b, T5b2);
    auto warped2 = f.warpfield.warpToLive(warped);
    solve(f, warped2, frame(1, T5b3));
    expect_warp(f, warped->getVertices(), T5b3);
}
TEST(OptTest, MultipleVerticesOneGroupOfDeformationNodesWarpAndReverseTest) {  // :632-698
    Fixture f;
    f.warpfield.init(f.epsilon_dynfu, f.g1);
    solve(f, frame(0, S5b), frame(1, T5b1));
    expect_warp(f, S5b, T5b1);
    solve(f, frame(0, T5b1), frame(1, S5b));
    expect_warp(f, S5b, S5b);
}

int main(int argc, char** argv) { return mt::run_all(argc, argv); }
