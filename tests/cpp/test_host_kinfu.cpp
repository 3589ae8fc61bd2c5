// test_host_kinfu.cpp — kfusion::KinFu::operator() (src/kfusion/kinfu.cpp:140-234) through the adaptor class: the rigid
// KinectFusion loop that DynFusion derives from in the reference.  The reference has no test of it; here a static
// synthetic scene is rendered from a camera moving along a known path and the tracked pose chain must follow it.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <kfusion/kinfu.hpp>

#include "minitest.hpp"

using namespace kfusion;

namespace {
// depth of two spheres in front of a tilted wall, seen from camera pose (R, t): X_world = R X_cam + t (millimetres)
std::vector<unsigned short> render(int W, int H, float f, const float R[9], const float t[3]) {
    std::vector<unsigned short> d((size_t)W * H, 0);
    const float cx = W / 2 - 0.5f, cy = H / 2 - 0.5f;
    const float C[2][3] = {{0.25f, -0.1f, 1.6f}, {-0.55f, 0.3f, 2.0f}}, rad[2] = {0.45f, 0.3f};
    const float pn[3] = {0.3f, 0.2f, -0.933f}, pd = -2.6f * 0.933f;  // wall: pn . X = pd
    for (int y = 3; y < H - 3; ++y)
        for (int x = 3; x < W - 3; ++x) {
            const float dc[3] = {(x - cx) / f, (y - cy) / f, 1.f};
            float dw[3], best = 1e9f;
            for (int i = 0; i < 3; ++i) dw[i] = R[3 * i] * dc[0] + R[3 * i + 1] * dc[1] + R[3 * i + 2] * dc[2];
            for (int k = 0; k < 2; ++k) {
                const float oc[3] = {t[0] - C[k][0], t[1] - C[k][1], t[2] - C[k][2]};
                const float a = dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2], b = 2 * (oc[0] * dw[0] + oc[1] * dw[1] + oc[2] * dw[2]),
                            c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - rad[k] * rad[k], disc = b * b - 4 * a * c;
                if (disc > 0) {
                    const float s = (-b - std::sqrt(disc)) / (2 * a);
                    if (s > 0 && s < best) best = s;
                }
            }
            const float den = pn[0] * dw[0] + pn[1] * dw[1] + pn[2] * dw[2];
            if (std::fabs(den) > 1e-6f) {
                const float s = (pd - (pn[0] * t[0] + pn[1] * t[1] + pn[2] * t[2])) / den;
                if (s > 0 && s < best) best = s;
            }
            if (best < 60.f) d[(size_t)y * W + x] = (unsigned short)std::lround(best * 1000.f);  // z_cam = s (dc.z = 1)
        }
    return d;
}

KinFuParams small_params(int W, int H) {
    KinFuParams p = KinFuParams::default_params();
    p.cols = W, p.rows = H;
    p.intr = Intr(525.f * W / 640.f, 525.f * H / 480.f, W / 2 - 0.5f, H / 2 - 0.5f);
    p.volume_dims = Vec3i::all(256);
    return p;
}

void pose_at(int i, float R[9], float t[3]) {  // 0.6 degrees about y and (6, -3, 4) mm per frame
    const float a = 0.0105f * i;
    const float r[9] = {std::cos(a), 0, std::sin(a), 0, 1, 0, -std::sin(a), 0, std::cos(a)};
    for (int j = 0; j < 9; ++j) R[j] = r[j];
    t[0] = 0.006f * i, t[1] = -0.003f * i, t[2] = 0.004f * i;
}
}  // namespace

TEST(KinFuTest, DefaultParamsAreTheReferences) {
    const KinFuParams p = KinFuParams::default_params();  // kinfu.cpp:10-44
    ASSERT_EQ(p.cols, 640);
    ASSERT_EQ(p.volume_dims[0], 512);
    ASSERT_NEAR(p.icp_dist_thres, 0.1f, 1e-9);
    ASSERT_NEAR(p.icp_angle_thres, 30.f * 0.017453293f, 1e-7);
    ASSERT_TRUE((p.icp_iter_num == std::vector<int>{10, 5, 4, 0}));
    ASSERT_NEAR(p.intr(2).fx, 525.f / 4, 1e-6);
    ASSERT_NEAR(p.intr(1).cy, 239.5f / 2, 1e-6);
    KinFuParams bad = p;
    bad.volume_dims = Vec3i(100, 128, 128);  // :47
    bool threw = false;
    try {
        KinFu k(bad);
    } catch (const dfa::Error&) {
        threw = true;
    }
    ASSERT_TRUE(threw);
}

TEST(KinFuTest, TracksAMovingCameraOverAStaticScene) {
    const int W = 320, H = 240, N = 8;
    KinFu kinfu(small_params(W, H));
    for (int i = 0; i < N; ++i) {
        float R[9], t[3];
        pose_at(i, R, t);
        cuda::Depth depth;
        depth.upload(render(W, H, kinfu.params().intr.fx, R, t), W);
        const bool has_image = kinfu(depth);
        ASSERT_EQ(has_image, i >= 2);  // :164-174, :230-232
        ASSERT_EQ(kinfu.frameCounter(), i + 1);
        const Affine3f pose = kinfu.getCameraPose();
        // Frame 1 is aligned to the MEASURED maps of frame 0 and lands within 3e-4; from frame 2 on the target is the
        // raycast of a model that is the previous frame alone (the volume is cleared every frame, kinfu.cpp:209-210),
        // and the pose drifts by a constant ~2.1e-3 rad and ~0.7 mm per frame: integrate() samples the depth image at
        // texel floor(projection) (tsdf_volume.cu:73), i.e. half a pixel off the ray convention of the measured maps
        // and of the raycaster — 0.5 px / 262 px = 1.9e-3 rad.  That is the reference's arithmetic, reproduced bit for
        // bit (a model averaged over many frames hides it; this fork's per-frame clear does not).  The bounds below are
        // that drift plus a margin.
        float er = 0, et = 0;
        for (int j = 0; j < 9; ++j) er = std::max(er, std::fabs(pose.R[j] - R[j]));
        for (int j = 0; j < 3; ++j) et = std::max(et, std::fabs(pose.t[j] - t[j]));
        if (std::getenv("DFA_TEST_VERBOSE")) std::printf("    frame %d: max |dR| %.5f  max |dt| %.5f m\n", i, er, et);
        for (int j = 0; j < 9; ++j) ASSERT_NEAR(pose.R[j], R[j], 5e-4 + 2.6e-3 * std::max(0, i - 1));
        for (int j = 0; j < 3; ++j) ASSERT_NEAR(pose.t[j], t[j], 5e-4 + 1.0e-3 * std::max(0, i - 1));
    }
    ASSERT_NEAR(kinfu.getCameraPose(0).t[0], 0.f, 1e-9);  // the chain starts at the identity
    const auto mesh = kinfu.extractMesh();  // the model of the last frame
    ASSERT_TRUE(mesh->polygons.size() > 10000 && mesh->cloud.size() == 3 * mesh->polygons.size());
    kinfu.reset();
    ASSERT_EQ(kinfu.frameCounter(), 0);
    ASSERT_NEAR(kinfu.getCameraPose().t[2], 0.f, 1e-9);
}

TEST(KinFuTest, ALostTrackResetsThePipeline) {
    const int W = 320, H = 240;
    KinFu kinfu(small_params(W, H));
    float R[9], t[3];
    pose_at(0, R, t);
    cuda::Depth depth;
    depth.upload(render(W, H, kinfu.params().intr.fx, R, t), W);
    ASSERT_TRUE(!kinfu(depth));
    std::vector<unsigned short> nothing((size_t)W * H, 0);  // an empty frame: the ICP system is singular (:190-193)
    depth.upload(nothing, W);
    ASSERT_TRUE(!kinfu(depth));
    ASSERT_EQ(kinfu.frameCounter(), 0);
}

int main(int argc, char** argv) { return mt::run_all(argc, argv); }
