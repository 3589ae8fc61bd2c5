// test_host_dynfusion.cpp — DynFusion's warp-field sequence through the host adaptor
// (reference: src/dynfu/dyn_fusion.cpp:147-242; the reference has no test of its own for it).
#include <algorithm>
#include <numeric>
#include <random>

#include <dynfu/dyn_fusion.hpp>

#include "minitest.hpp"

namespace {
typedef dfa::PointCloud<dfa::PointXYZ> Cloud;
typedef dfa::PointCloud<dfa::Normal> Normals;

// points on the camera-facing half of a sphere of radius 0.5 m around (0, 0, 1.5)
void half_sphere(int n, Cloud& v, Normals& nrm) {
    const double golden = 3.14159265358979323846 * (3.0 - std::sqrt(5.0));
    for (int i = 0; i < n; ++i) {
        const double z = -(i + 0.5) / n, r = std::sqrt(1.0 - z * z), a = golden * i;
        const float dx = (float)(r * std::cos(a)), dy = (float)(r * std::sin(a)), dz = (float)z;
        v.push_back(dfa::PointXYZ(0.5f * dx, 0.5f * dy, 1.5f + 0.5f * dz));
        nrm.push_back(dfa::Normal(dx, dy, dz));
    }
}
float dist2(const dfa::PointXYZ& a, const dfa::PointXYZ& b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return (dx * dx + dy * dy) + dz * dz;
}
}  // namespace

TEST(DynFusionTest, FindCorrespondingFrameIsTheNearestCanonicalVertex) {
    Cloud canon, live;
    Normals cn;
    half_sphere(3000, canon, cn);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-0.6f, 0.6f);
    for (int i = 0; i < 500; ++i) live.push_back(dfa::PointXYZ(u(rng), u(rng), 1.5f + u(rng)));
    live.push_back(canon[17]);  // an exact hit
    DynFusion df(DynFuParams::defaultParams());
    auto frame = df.findCorrespondingFrame(canon, cn, live);
    ASSERT_EQ(frame->getVertices().size(), live.size());
    ASSERT_EQ(frame->getNormals().size(), live.size());
    for (size_t i = 0; i < live.size(); ++i) {
        size_t best = 0;
        for (size_t j = 1; j < canon.size(); ++j)
            if (dist2(live[i], canon[j]) < dist2(live[i], canon[best])) best = j;
        ASSERT_EQ(frame->getVertices()[i].x, canon[best].x);
        ASSERT_EQ(frame->getVertices()[i].y, canon[best].y);
        ASSERT_EQ(frame->getVertices()[i].z, canon[best].z);
        ASSERT_EQ(frame->getNormals()[i].normal_z, cn[best].normal_z);
    }
}

TEST(DynFusionTest, InitSeedsEvery128thVertex) {
    Cloud canon;
    Normals cn;
    half_sphere(128 * 16 + 5, canon, cn);
    DynFuParams p = DynFuParams::defaultParams();
    DynFusion df(p);
    df.init(canon, cn);
    auto nodes = df.getWarpfield()->getNodes();
    ASSERT_EQ(nodes.size(), (size_t)17);
    for (size_t i = 0; i < nodes.size(); ++i) {
        ASSERT_EQ(nodes[i]->getPosition().x, canon[128 * i].x);
        ASSERT_NEAR(nodes[i]->getRadialBasisWeight(), 3 * p.epsilon, 1e-7);
    }
    ASSERT_EQ(df.getCanonicalWarpedToLive()->getVertices().size(), canon.size());
}

TEST(DynFusionTest, WarpCanonicalToLiveFollowsAShiftedShuffledLiveCloud) {
    Cloud canon, live;
    Normals cn, ln;
    const int n = 128 * 32;
    half_sphere(n, canon, cn);
    // live = canonical moved by 4 mm (a tenth of the vertex spacing), in a different order
    const float shift[3] = {0.004f, -0.002f, 0.003f};
    std::vector<int> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::shuffle(perm.begin(), perm.end(), std::mt19937(3));
    for (int i : perm) {
        live.push_back(dfa::PointXYZ(canon[i].x + shift[0], canon[i].y + shift[1], canon[i].z + shift[2]));
        ln.push_back(cn[i]);
    }
    DynFuParams p = DynFuParams::defaultParams();
    p.epsilon     = 0.05f;
    p.lambda      = 0.f;
    DynFusion df(p);
    df.solverParams.numIter = 3, df.solverParams.nonLinearIter = 2;
    df.init(canon, cn);
    df.addLiveFrame(1, live, ln);
    df.warpCanonicalToLiveOpt(dfa::Affine3f());
    // the frame stored by the call is the canonical cloud warped BEFORE the solve (dyn_fusion.cpp:196)
    auto before = df.getCanonicalWarpedToLive();
    ASSERT_EQ(before->getVertices()[100].x, canon[100].x);
    auto after = df.getWarpfield()->warpToLive(std::make_shared<dynfu::Frame>(0, canon, cn));
    double err = 0;
    for (int i = 0; i < n; ++i) {
        const auto& w = after->getVertices()[i];
        err += std::fabs(w.x - canon[i].x - shift[0]) + std::fabs(w.y - canon[i].y - shift[1]) +
               std::fabs(w.z - canon[i].z - shift[2]);
    }
    err /= n;
    const double moved = std::fabs(shift[0]) + std::fabs(shift[1]) + std::fabs(shift[2]);
    ASSERT_TRUE(err < 0.2 * moved);
}

int main(int argc, char** argv) { return mt::run_all(argc, argv); }
