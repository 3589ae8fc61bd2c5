// test_host_dynfusion.cpp — DynFusion's warp-field sequence through the host adaptor
// (reference: src/dynfu/dyn_fusion.cpp:147-242; the reference has no test of its own for it).
#include <algorithm>
#include <numeric>
#include <random>

#include <dynfu/dyn_fusion.hpp>

#include "../../oracle/oracle.h"
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

TEST(DynFusionTest, VoxelGridFilterMatchesTheOracleRestatementOfPcl) {
    Cloud pts;
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> u(-0.4f, 0.7f);
    for (int i = 0; i < 5000; ++i) pts.push_back(dfa::PointXYZ(u(rng), u(rng), 1.0f + 0.3f * u(rng)));
    pts.push_back(dfa::PointXYZ(NAN, 0.f, 0.f));  // non-finite points are skipped
    Cloud out = dfa::voxelGridFilter(pts, 0.05f);
    std::vector<float> flat(3 * pts.size()), ref(3 * pts.size());
    for (size_t i = 0; i < pts.size(); ++i) flat[3 * i] = pts[i].x, flat[3 * i + 1] = pts[i].y, flat[3 * i + 2] = pts[i].z;
    const int m = orc_voxel_grid(flat.data(), (int)pts.size(), 0.05f, ref.data());
    ASSERT_EQ((size_t)m, out.size());
    ASSERT_TRUE(m > 500 && m < 5000);
    for (int i = 0; i < m; ++i) {
        ASSERT_EQ(out[i].x, ref[3 * i]);
        ASSERT_EQ(out[i].y, ref[3 * i + 1]);
        ASSERT_EQ(out[i].z, ref[3 * i + 2]);
    }
    ASSERT_TRUE(dfa::voxelGridFilter(Cloud(), 0.05f).empty());
}

TEST(DynFusionTest, UpdateInsertsNodesWhereTheFieldDoesNotReach) {
    Cloud canon, extra;
    Normals cn, en;
    half_sphere(128 * 32, canon, cn);
    DynFuParams p = DynFuParams::defaultParams();  // epsilon 0.1: dg_w = 0.3 m at seeding, 0.2 m for inserted nodes
    DynFusion df(p);
    df.init(canon, cn);
    auto wf = df.getWarpfield();
    const size_t before = wf->getNodes().size();
    // every canonical vertex is supported by the seeded nodes
    ASSERT_TRUE(wf->getUnsupportedVertices(std::make_shared<dynfu::Frame>(0, canon, cn)).empty());
    // a patch 1 m away is not
    for (int i = 0; i < 400; ++i) extra.push_back(dfa::PointXYZ(1.5f + 0.001f * i, 0.2f + 0.0005f * (i % 37), 1.5f)), en.push_back(dfa::Normal());
    auto frame = std::make_shared<dynfu::Frame>(1, extra, en);
    ASSERT_EQ(wf->getUnsupportedVertices(frame).size(), (size_t)400);
    wf->update(frame);
    auto nodes = wf->getNodes();
    const Cloud seeds = dfa::voxelGridFilter(extra, 0.05f);
    ASSERT_EQ(nodes.size(), before + seeds.size());
    ASSERT_TRUE(seeds.size() >= 8 && seeds.size() <= 12);  // 0.4 m of points in 5 cm leaves
    for (size_t i = 0; i < seeds.size(); ++i) {
        ASSERT_EQ(nodes[before + i]->getPosition().x, seeds[i].x);
        ASSERT_NEAR(nodes[before + i]->getRadialBasisWeight(), 2 * p.epsilon, 1e-7);  // warp_field.cpp:79
        // identity field so far: calcDQB is the identity, real part normalised
        ASSERT_NEAR(nodes[before + i]->getTransformation()->getReal().a, 1.f, 1e-6);
    }
    // now the patch is supported
    ASSERT_TRUE(wf->getUnsupportedVertices(frame).empty());
}

// DynFusion::operator() (dyn_fusion.cpp:48-145) on two synthetic depth frames of a sphere in front of a wall
TEST(DynFusionTest, OperatorRunsTheWholeFrameSequence) {
    const int W = 160, H = 120;
    auto make = [&](float cz) {
        std::vector<unsigned short> d((size_t)W * H);
        const float f = 131.25f, cx = W / 2 - 0.5f, cy = H / 2 - 0.5f, R = 0.5f;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float dir[3] = {(x - cx) / f, (y - cy) / f, 1.f};
                const float n = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + 1.f);
                for (float& v : dir) v /= n;
                const float b = dir[2] * cz, disc = b * b - (cz * cz - R * R);
                float z = 2.5f;
                if (disc > 0) z = (b - std::sqrt(disc)) * dir[2];
                d[(size_t)y * W + x] = (x < 3 || y < 3 || x >= W - 3 || y >= H - 3) ? 0 : (unsigned short)std::lround(z * 1000.f);
            }
        return d;
    };
    DynFuParams p = DynFuParams::defaultParams();
    p.kinfuParams.cols = W, p.kinfuParams.rows = H;
    p.kinfuParams.intr = kfusion::Intr(131.25f, 131.25f, W / 2 - 0.5f, H / 2 - 0.5f);
    p.kinfuParams.volume_dims = kfusion::Vec3i::all(64);
    p.epsilon = 0.05f;
    DynFusion df(p);
    df.solverParams.numIter = 2, df.solverParams.nonLinearIter = 2, df.solverParams.linearIter = 64;
    kfusion::cuda::Depth d0, d1;
    d0.upload(make(1.5f), W);
    d1.upload(make(1.49f), W);  // the sphere moved 1 cm towards the camera
    ASSERT_TRUE(df(d0) == false);  // :97: nothing more to do with the first frame
    const size_t n_canon = df.getCanonicalWarpedToLive()->getVertices().size();
    const size_t n_nodes = df.getWarpfield()->getNodes().size();
    ASSERT_TRUE(n_canon > 3000 && n_canon % 3 == 0);
    ASSERT_EQ(n_nodes, (n_canon + 127) / 128);
    ASSERT_TRUE(df(d1) == true);
    ASSERT_EQ(df.frameCounter(), 2);
    ASSERT_TRUE(df.getLiveFrame()->getVertices().size() > 3000);
    ASSERT_TRUE(df.getWarpfield()->getNodes().size() >= n_nodes);
    // the solve moved the field: the sphere's nodes picked up a translation towards the camera (-z)
    double tz = 0;
    int moved = 0;
    for (auto& n : df.getWarpfield()->getNodes()) {
        if (n->getPosition().z < 1.6f) tz += n->getTransformation()->getTranslation()[2], ++moved;
    }
    ASSERT_TRUE(moved > 5 && tz / moved < -0.002);
}

// extension of the interface: with DynFuParams::mesh_normals the canonical frame carries the
// gradient of the TSDF at its vertices; off (the default) the normals are default-constructed as in the reference
TEST(DynFusionTest, MeshNormalsFromTheTsdfGradientFaceTheCamera) {
    const int W = 160, H = 120;
    std::vector<unsigned short> d((size_t)W * H);
    const float f = 131.25f, cx = W / 2 - 0.5f, cy = H / 2 - 0.5f, R = 0.5f, cz = 1.5f;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float dir[3] = {(x - cx) / f, (y - cy) / f, 1.f};
            const float n = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + 1.f);
            for (float& v : dir) v /= n;
            const float b = dir[2] * cz, disc = b * b - (cz * cz - R * R);
            float z = 0.f;  // no background: only the sphere is observed
            if (disc > 0) z = (b - std::sqrt(disc)) * dir[2];
            d[(size_t)y * W + x] = (unsigned short)std::lround(z * 1000.f);
        }
    for (int with = 0; with < 2; ++with) {
        DynFuParams p = DynFuParams::defaultParams();
        p.kinfuParams.cols = W, p.kinfuParams.rows = H;
        p.kinfuParams.intr = kfusion::Intr(f, f, cx, cy);
        p.kinfuParams.volume_dims = kfusion::Vec3i::all(64);
        p.mesh_normals = with != 0;
        DynFusion df(p);
        kfusion::cuda::Depth d0;
        d0.upload(d, W);
        ASSERT_TRUE(df(d0) == false);
        auto frame = df.getCanonicalWarpedToLive();
        const auto& verts = frame->getVertices();
        const auto& norms = frame->getNormals();
        ASSERT_TRUE(verts.size() > 1000 && norms.size() == verts.size());
        // the volume's frame: the sphere's centre sits at camera (0, 0, 1.5) = volume pose^-1 applied; the observed cap
        // faces the camera, so the outward normal has a negative z component there
        int facing = 0, finite = 0, nonzero = 0;
        for (size_t i = 0; i < verts.size(); ++i) {
            const dfa::Normal& n = norms[i];
            if (n.normal_x != 0.f || n.normal_y != 0.f || n.normal_z != 0.f) ++nonzero;
            if (std::isfinite(n.normal_x) && std::isfinite(n.normal_y) && std::isfinite(n.normal_z)) {
                ++finite;
                if (n.normal_z < 0.f) ++facing;
            }
        }
        if (!with) {
            ASSERT_EQ(nonzero, 0);  // the reference's behaviour
        } else {
            ASSERT_TRUE(finite > (int)(0.9 * verts.size()));
            ASSERT_TRUE(facing > (int)(0.9 * finite));
        }
    }
}

int main(int argc, char** argv) { return mt::run_all(argc, argv); }
