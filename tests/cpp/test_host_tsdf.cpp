// test_host_tsdf.cpp — kfusion::cuda::TsdfVolume adaptor (create / clear / integrate / raycast /
// setters with the trunc-dist clamp) against the CPU oracle, bit-exact.  The reference has no
// TSDF tests; the call sequence is KinFu's (src/kfusion/kinfu.cpp:47-60,206-225).
#include <cstring>

#include <kfusion/cuda/imgproc.hpp>
#include <kfusion/cuda/marching_cubes.hpp>
#include <kfusion/cuda/tsdf_volume.hpp>

#include "../../include/dynfu_amd.h"
#include "../../oracle/oracle.h"
#include "minitest.hpp"

using namespace kfusion;

namespace {
std::vector<unsigned short> make_depth(int W, int H) {
    std::vector<unsigned short> d((size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float u = (x - W / 2) / (float)W, v = (y - H / 2) / (float)H;
            const float r2 = u * u + v * v;
            d[(size_t)y * W + x] = r2 < 0.09f ? (unsigned short)(1200 + 900 * r2 * 10) : 2400;  // bump in front of a wall
            if (x < 3 || y < 3 || x >= W - 3 || y >= H - 3) d[(size_t)y * W + x] = 0;
        }
    return d;
}
}  // namespace

TEST(TsdfVolumeTest, SettersFollowTheReferenceDefaultsAndClamp) {
    cuda::TsdfVolume vol(Vec3i::all(64));
    ASSERT_EQ(vol.getMaxWeight(), 128);                            // tsdf_volume.cpp:21
    ASSERT_NEAR(vol.getTruncDist(), 2.1f * 3.f / 64, 1e-7);        // 0.03 clamped to 2.1 voxels (:57-61)
    vol.setTruncDist(0.2f);
    ASSERT_NEAR(vol.getTruncDist(), 0.2f, 1e-7);
    vol.setSize(Vec3f::all(1.f));
    ASSERT_NEAR(vol.getVoxelSize()[0], 1.f / 64, 1e-9);
    vol.setPose(Affine3f().translate(Vec3f(1, 2, 3)));
    vol.applyAffine(Affine3f().translate(Vec3f(1, 0, 0)));
    ASSERT_NEAR(vol.getPose().translation()[0], 2.f, 1e-7);
}

TEST(TsdfVolumeTest, IntegrateAndRaycastMatchTheOracleBitExactly) {
    const int W = 160, H = 120, DIM = 64;
    const Intr intr(131.25f, 131.25f, W / 2 - 0.5f, H / 2 - 0.5f);
    cuda::TsdfVolume vol(Vec3i::all(DIM));
    vol.setTruncDist(0.04f), vol.setMaxWeight(64), vol.setSize(Vec3f::all(3.f));
    vol.setPose(Affine3f().translate(Vec3f(-1.5f, -1.5f, 0.5f)));  // kinfu.cpp:22
    vol.setRaycastStepFactor(0.75f), vol.setGradientDeltaFactor(0.5f);

    std::vector<unsigned short> depth = make_depth(W, H);
    cuda::Depth d_depth;
    d_depth.upload(depth, W);
    cuda::Dists dists;
    cuda::computeDists(d_depth, dists, intr);
    const Affine3f camera;  // identity
    vol.clear();
    vol.integrate(dists, camera, intr);
    vol.integrate(dists, camera, intr);

    // oracle
    std::vector<unsigned short> o_dists((size_t)W * H);
    orc_compute_dists(depth.data(), W * 2, o_dists.data(), W * 2, W, H, intr.fx, intr.fy, intr.cx, intr.cy);
    std::vector<unsigned short> h_dists;
    int cols;
    dists.download(h_dists, cols);
    ASSERT_TRUE(h_dists == o_dists);
    std::vector<uint32_t> o_vol((size_t)DIM * DIM * DIM, 0u), h_vol(o_vol.size());
    const Vec3f vs = vol.getVoxelSize();
    float vol2cam[12];
    (camera.inv() * vol.getPose()).to12(vol2cam);
    for (int rep = 0; rep < 2; ++rep)
        orc_tsdf_integrate(o_dists.data(), W * 2, W, H, o_vol.data(), DIM, DIM, DIM, vs.v, vol.getTruncDist(), 64, vol2cam,
                           intr.fx, intr.fy, intr.cx, intr.cy, 4);
    vol.data().download(h_vol.data(), h_vol.size() * 4);
    ASSERT_TRUE(h_vol == o_vol);

    // fused clear + integrate == clear(); integrate()
    cuda::TsdfVolume vol2(Vec3i::all(DIM));
    vol2.setTruncDist(0.04f), vol2.setMaxWeight(64), vol2.setPose(vol.getPose());
    vol2.clearAndIntegrate(dists, camera, intr);
    vol.clear();
    vol.integrate(dists, camera, intr);
    std::vector<uint32_t> h2(h_vol.size());
    vol2.data().download(h2.data(), h2.size() * 4);
    vol.data().download(h_vol.data(), h_vol.size() * 4);
    ASSERT_TRUE(h2 == h_vol);

    // raycast (points) vs oracle, raw bits
    cuda::Cloud points(H, W);
    cuda::Normals normals(H, W);
    vol.raycast(camera, intr, points, normals);
    std::vector<Point> hp;
    std::vector<Normal> hn;
    points.download(hp, cols), normals.download(hn, cols);
    float cam2vol[12], rinv[9];
    Affine3f c2v = vol.getPose().inv() * camera;
    c2v.to12(cam2vol), c2v.inverse_rotation(rinv);
    std::vector<float> op((size_t)W * H * 4), on(op.size());
    orc_tsdf_raycast_points(h_vol.data(), DIM, DIM, DIM, vs.v, vol.getTruncDist(), cam2vol, rinv, intr.fx, intr.fy, intr.cx,
                            intr.cy, 0.75f, 0.5f, op.data(), W * 16, on.data(), W * 16, W, H, 4);
    ASSERT_TRUE(std::memcmp(hp.data(), op.data(), op.size() * 4) == 0);
    ASSERT_TRUE(std::memcmp(hn.data(), on.data(), on.size() * 4) == 0);
    size_t hits = 0;
    for (auto& p : hp) hits += p.x == p.x;
    ASSERT_TRUE(hits > (size_t)W * H / 2);

    // swap() hands the blob over (tsdf_volume.cpp:71)
    cuda::CudaData other;
    vol.swap(other);
    ASSERT_TRUE(vol.data().empty() && other.sizeBytes() == (size_t)DIM * DIM * DIM * 4);
}

// cuda::MarchingCubes::run as DynFusion::operator() calls it (dyn_fusion.cpp:73-75,119-121): mesh of an
// integrated depth frame, bit-exact against the oracle given the same case tables
// The occupancy map is trusted only while the volume object is the sole owner of its storage: a handle taken
// BEFORE a clear can write the voxels after it; a copy of the volume shares the map it may clear.
TEST(TsdfVolumeTest, OccupancyMapIsOnlyTrustedBySoleOwners) {
    cuda::TsdfVolume vol(Vec3i::all(32));
    ASSERT_TRUE(vol.occupancy() != nullptr);  // created and cleared by this object alone
    {
        cuda::CudaData h = vol.data();             // a writable handle ...
        ASSERT_TRUE(vol.occupancy() == nullptr);
        vol.clear();                          // ... that outlives a clear: the voxels can still change behind the map
        ASSERT_TRUE(vol.occupancy() == nullptr);
    }
    ASSERT_TRUE(vol.occupancy() == nullptr);  // (what the handle wrote meanwhile is unknown)
    vol.clear();
    ASSERT_TRUE(vol.occupancy() != nullptr);  // sole owner again, and swept
    {
        const cuda::TsdfVolume& cv = vol;
        const cuda::CudaData ro = cv.data();        // a const handle: untrusted while it lives, nothing lost afterwards
        ASSERT_TRUE(vol.occupancy() == nullptr);
    }
    ASSERT_TRUE(vol.occupancy() != nullptr);
    {
        cuda::TsdfVolume copy(vol);           // shares voxels AND map
        ASSERT_TRUE(vol.occupancy() == nullptr && copy.occupancy() == nullptr);
        copy.clear();
        ASSERT_TRUE(copy.occupancy() == nullptr);
    }
    ASSERT_TRUE(vol.occupancy() == nullptr);  // the copy may have written: unknown until this object sweeps again
    vol.clear();
    ASSERT_TRUE(vol.occupancy() != nullptr);
    cuda::CudaData other((size_t)32 * 32 * 32 * sizeof(int));
    vol.swap(other);
    ASSERT_TRUE(vol.occupancy() == nullptr);
}

TEST(MarchingCubesTest, RunMatchesTheOracleBitExactly) {
    const int W = 160, H = 120, DIM = 64;
    const Intr intr(131.25f, 131.25f, W / 2 - 0.5f, H / 2 - 0.5f);
    cuda::TsdfVolume vol(Vec3i::all(DIM));
    vol.setTruncDist(0.04f), vol.setMaxWeight(64), vol.setSize(Vec3f::all(3.f));
    vol.setPose(Affine3f().translate(Vec3f(-1.5f, -1.5f, 0.5f)));
    cuda::Depth d_depth;
    d_depth.upload(make_depth(W, H), W);
    cuda::Dists dists;
    cuda::computeDists(d_depth, dists, intr);
    vol.clearAndIntegrate(dists, Affine3f(), intr);

    cuda::MarchingCubes mc;
    dfa::DeviceArray<cuda::MarchingCubes::PointType> buffer;
    auto triangles = mc.run(vol, buffer);
    ASSERT_EQ(buffer.size(), (size_t)cuda::MarchingCubes::DEFAULT_TRIANGLES_BUFFER_SIZE);  // marching_cubes.cpp:23-25
    ASSERT_TRUE(triangles.size() > 3000 && triangles.size() % 3 == 0);
    ASSERT_EQ((int)triangles.size(), mc.totalVertices());

    std::vector<uint32_t> h_vol((size_t)DIM * DIM * DIM);
    vol.data().download(h_vol.data(), h_vol.size() * 4);
    std::vector<int32_t> tri(256 * 16), nv(256);
    ASSERT_EQ(dfa_mc_default_tables(tri.data(), nv.data()), 0);
    const Vec3f vs = vol.getVoxelSize();
    std::vector<float> ref(4 * triangles.size());
    long occupied = 0;
    const long total = orc_marching_cubes(h_vol.data(), DIM, DIM, DIM, vs.v, tri.data(), nv.data(), ref.data(),
                                          (long)triangles.size(), &occupied);
    ASSERT_EQ(total, (long)triangles.size());
    std::vector<cuda::MarchingCubes::PointType> got;
    triangles.download(got);
    ASSERT_TRUE(std::memcmp(got.data(), ref.data(), ref.size() * 4) == 0);

    // a buffer that is too small gets the first points only
    dfa::DeviceArray<cuda::MarchingCubes::PointType> small(300);
    auto part = mc.run(vol, small);
    ASSERT_EQ(part.size(), (size_t)300);
    ASSERT_EQ(mc.totalVertices(), (int)total);
    part.download(got);
    ASSERT_TRUE(std::memcmp(got.data(), ref.data(), 300 * 16) == 0);

    // an empty volume gives an empty array (marching_cubes.cpp:42-46)
    vol.clear();
    ASSERT_TRUE(mc.run(vol, buffer).empty());
}

// the depth pre-processing of DynFusion::operator() / KinFu::operator() (dyn_fusion.cpp:58-66, kinfu.cpp:150-175)
// through the kfusion::cuda functions, against the oracle, bit for bit
TEST(ImgprocTest, PreprocessingChainMatchesTheOracleBitExactly) {
    const int W = 160, H = 120;
    const Intr intr(131.25f, 131.25f, W / 2 - 0.5f, H / 2 - 0.5f);
    std::vector<unsigned short> depth = make_depth(W, H);
    for (int i = 0; i < W * H; i += 7) depth[i] = (unsigned short)(depth[i] + (i % 5));  // a little texture
    cuda::Depth d_in, d_f, d_half;
    d_in.upload(depth, W);
    cuda::depthBilateralFilter(d_in, d_f, 7, 4.5f, 0.04f);  // kinfu.cpp:26-28
    cuda::depthTruncation(d_f, 2.0f);
    cuda::depthBuildPyramid(d_f, d_half, 0.04f);
    cuda::Normals n_f;
    cuda::computeNormalsAndMaskDepth(intr, d_f, n_f);
    cuda::Depth d_q;
    cuda::Normals n_q;
    cuda::resizeDepthNormals(d_f, n_f, d_q, n_q);
    cuda::Cloud pts, pts_q;
    cuda::Normals nrm, nrm_q;
    cuda::computePointNormals(intr, d_f, pts, nrm);
    cuda::resizePointsNormals(pts, nrm, pts_q, nrm_q);

    std::vector<unsigned short> o_f(depth.size()), o_half((size_t)(W / 2) * (H / 2)), o_q(o_half.size());
    orc_bilateral(depth.data(), W * 2, o_f.data(), W * 2, W, H, 7, 4.5f, 0.04f);
    orc_truncate_depth(o_f.data(), W * 2, W, H, 2.0f);
    orc_depth_pyr(o_f.data(), W * 2, W, H, o_half.data(), (W / 2) * 2, 0.04f);
    std::vector<float> o_n((size_t)W * H * 4), o_nq(o_half.size() * 4);
    orc_normals_mask_depth(o_f.data(), W * 2, W, H, intr.fx, intr.fy, intr.cx, intr.cy, o_n.data(), W * 16);
    orc_resize_depth_normals(o_f.data(), W * 2, o_n.data(), W * 16, W, H, o_q.data(), (W / 2) * 2, o_nq.data(), (W / 2) * 16);
    std::vector<float> o_p((size_t)W * H * 4), o_pn(o_p.size()), o_pq(o_nq.size()), o_pnq(o_nq.size());
    orc6_points_normals(o_f.data(), W * 2, W, H, intr.fx, intr.fy, intr.cx, intr.cy, o_p.data(), W * 16, o_pn.data(), W * 16);
    orc_resize_points_normals(o_p.data(), W * 16, o_pn.data(), W * 16, W, H, o_pq.data(), (W / 2) * 16, o_pnq.data(), (W / 2) * 16);

    std::vector<unsigned short> h;
    int cols;
    d_f.download(h, cols);
    ASSERT_TRUE(h == o_f);
    d_half.download(h, cols);
    ASSERT_TRUE(cols == W / 2 && h == o_half);
    d_q.download(h, cols);
    ASSERT_TRUE(h == o_q);
    std::vector<Normal> hn;
    n_f.download(hn, cols);
    ASSERT_TRUE(std::memcmp(hn.data(), o_n.data(), o_n.size() * 4) == 0);
    n_q.download(hn, cols);
    ASSERT_TRUE(std::memcmp(hn.data(), o_nq.data(), o_nq.size() * 4) == 0);
    std::vector<Point> hp;
    pts_q.download(hp, cols);
    ASSERT_TRUE(std::memcmp(hp.data(), o_pq.data(), o_pq.size() * 4) == 0);
    nrm_q.download(hn, cols);
    ASSERT_TRUE(std::memcmp(hn.data(), o_pnq.data(), o_pnq.size() * 4) == 0);
}

int main(int argc, char** argv) { return mt::run_all(argc, argv); }
