// test_host_icp.cpp — kfusion::cuda::ProjectiveICP::estimateTransform through the host adaptor, the way
// KinFu::operator() drives it (src/kfusion/kinfu.cpp:150-200): bilateral -> pyramids -> masked depth + normals per
// level -> ICP.  The reference has no ICP test; here a synthetic scene is rendered from two camera poses and the
// estimate must recover the relative pose.
#include <cmath>

#include <kfusion/cuda/imgproc.hpp>
#include <kfusion/cuda/projective_icp.hpp>

#include "minitest.hpp"

using namespace kfusion;

namespace {
// depth of two spheres in front of a tilted wall (roll about the optical axis is observable), seen from camera
// pose (R, t): X_world = R X_cam + t
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
            if (best < 1e8f && best < 60.f) d[(size_t)y * W + x] = (unsigned short)std::lround(best * 1000.f);  // z_cam = s (dc.z = 1)
        }
    return d;
}

struct Pyramids {
    cuda::ProjectiveICP::DepthPyr depth;
    cuda::ProjectiveICP::NormalsPyr normals;
    cuda::ProjectiveICP::PointsPyr points;
    cuda::ProjectiveICP::NormalsPyr pnormals;
};
Pyramids build(const std::vector<unsigned short>& raw, int W, const Intr& intr, int levels) {
    Pyramids p;
    p.depth.resize(levels), p.normals.resize(levels), p.points.resize(levels), p.pnormals.resize(levels);
    cuda::Depth in;
    in.upload(raw, W);
    cuda::depthBilateralFilter(in, p.depth[0], 7, 4.5f, 0.04f);  // kinfu.cpp:150-153
    for (int i = 1; i < levels; ++i) cuda::depthBuildPyramid(p.depth[i - 1], p.depth[i], 0.04f);
    for (int i = 0; i < levels; ++i) {
        const int div = 1 << i;
        const Intr li(intr.fx / div, intr.fy / div, intr.cx / div, intr.cy / div);
        cuda::computePointNormals(li, p.depth[i], p.points[i], p.pnormals[i]);
        cuda::computeNormalsAndMaskDepth(li, p.depth[i], p.normals[i]);  // kinfu.cpp:161-167
    }
    return p;
}
}  // namespace

TEST(ProjectiveIcpTest, DefaultsAndLevels) {
    cuda::ProjectiveICP icp;
    ASSERT_NEAR(icp.getDistThreshold(), 0.1f, 1e-9);
    ASSERT_NEAR(icp.getAngleThreshold(), 20.f * 0.017453293f, 1e-9);
    ASSERT_EQ(icp.getUsedLevelsNum(), 3);  // {10, 5, 4, 0}
    icp.setIterationsNum({3});
    ASSERT_EQ(icp.getUsedLevelsNum(), 1);
    icp.setIterationsNum({1, 2, 3, 4, 5, 6});
    ASSERT_EQ(icp.getUsedLevelsNum(), 4);
}

TEST(ProjectiveIcpTest, EstimateTransformRecoversTheCameraMotion) {
    const int W = 320, H = 240;
    const Intr intr(262.5f, 262.5f, W / 2 - 0.5f, H / 2 - 0.5f);
    // previous camera at the origin; current camera rotated by 1.2 degrees about y and moved by (15, -8, 10) mm
    const float a = 0.021f, Rc[9] = {std::cos(a), 0, std::sin(a), 0, 1, 0, -std::sin(a), 0, std::cos(a)}, tc[3] = {0.015f, -0.008f, 0.010f};
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z[3] = {0, 0, 0};
    Pyramids prev = build(render(W, H, intr.fx, I, z), W, intr, 3), curr = build(render(W, H, intr.fx, Rc, tc), W, intr, 3);
    cuda::ProjectiveICP icp;
    icp.setDistThreshold(0.1f), icp.setAngleThreshold(30.f * 0.017453293f);  // kinfu.cpp:31-32
    for (int variant = 0; variant < 2; ++variant) {
        Affine3f est;
        const bool ok = variant == 0 ? icp.estimateTransform(est, intr, curr.depth, curr.normals, prev.depth, prev.normals)
                                     : icp.estimateTransform(est, intr, curr.points, curr.pnormals, prev.points, prev.pnormals);
        ASSERT_TRUE(ok);
        // the estimate maps current-camera coordinates into the previous camera's frame: X_prev = R_c X_cur + t_c
        for (int i = 0; i < 9; ++i) ASSERT_NEAR(est.R[i], Rc[i], 3e-3);
        for (int i = 0; i < 3; ++i) ASSERT_NEAR(est.t[i], tc[i], 5e-3);
    }
    // nothing to align against: the normal equations are singular and the reference returns false (:136-142)
    cuda::ProjectiveICP::DepthPyr empty(3);
    cuda::ProjectiveICP::NormalsPyr nempty(3);
    for (int i = 0; i < 3; ++i) {
        std::vector<unsigned short> zeros((size_t)(W >> i) * (H >> i), 0);
        empty[i].upload(zeros, W >> i);
        const int div = 1 << i;
        cuda::computeNormalsAndMaskDepth(Intr(intr.fx / div, intr.fy / div, intr.cx / div, intr.cy / div), empty[i], nempty[i]);
    }
    Affine3f est;
    ASSERT_TRUE(!icp.estimateTransform(est, intr, empty, nempty, prev.depth, prev.normals));
}

int main(int argc, char** argv) { return mt::run_all(argc, argv); }
