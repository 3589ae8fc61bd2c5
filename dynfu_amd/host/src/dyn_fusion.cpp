// dyn_fusion.cpp — DynFusion's warp-field sequence on the dynfu_amd C ABI
// (reference: src/dynfu/dyn_fusion.cpp:6-31,147-242).
#include <dynfu/dyn_fusion.hpp>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>

#include <dfa_host/device.hpp>

#include "../../../include/dynfu_amd.h"

kfusion::KinFuParams kfusion::KinFuParams::default_params() {  // src/kfusion/kinfu.cpp:10-44
    KinFuParams p;
    p.cols = 640, p.rows = 480;
    p.intr        = Intr(525.f, 525.f, p.cols / 2 - 0.5f, p.rows / 2 - 0.5f);
    p.volume_dims = Vec3i::all(512);
    p.volume_size = Vec3f::all(3.f);
    p.volume_pose = Affine3f().translate(Vec3f(-p.volume_size[0] / 2, -p.volume_size[1] / 2, 0.5f));
    p.bilateral_sigma_depth = 0.04f, p.bilateral_sigma_spatial = 4.5f, p.bilateral_kernel_size = 7;
    p.icp_truncate_depth_dist = 0.f;
    p.icp_dist_thres = 0.1f, p.icp_angle_thres = 30.f * 0.017453293f, p.icp_iter_num = {10, 5, 4, 0};
    p.tsdf_min_camera_movement = 0.f;
    p.tsdf_trunc_dist = 0.04f, p.tsdf_max_weight = 64;
    p.raycast_step_factor = 0.75f, p.gradient_delta_factor = 0.5f;
    return p;
}

DynFuParams DynFuParams::defaultParams() {  // dyn_fusion.cpp:6-31
    DynFuParams p;
    p.kinfuParams             = kfusion::KinFuParams::default_params();
    p.kinfuParams.volume_dims = kfusion::Vec3i::all(128);  // :10
    p.intr        = p.kinfuParams.intr;
    p.tukeyOffset = 4.652f;
    p.lambda      = 200.f;
    p.psi_data    = 0.01f;
    p.psi_reg     = 1e-4f;
    p.L           = 4;
    p.beta        = 4;
    p.epsilon     = 0.1f;
    return p;
}

DynFusion::DynFusion(const DynFuParams& params) : dynfuParams(params) {
    solverParams.numIter       = 24;  // dyn_fusion.cpp:183-189
    solverParams.nonLinearIter = 16;
    solverParams.linearIter    = 256;
    solverParams.useOpt        = true;
    solverParams.useOptLM      = false;
    solverParams.earlyOut      = true;
}
DynFusion::~DynFusion() = default;

DynFuParams& DynFusion::params() { return dynfuParams; }

void DynFusion::init(dfa::PointCloud<dfa::PointXYZ>& canonicalVertices, dfa::PointCloud<dfa::Normal>& canonicalNormals) {
    initCanonicalFrame(canonicalVertices, canonicalNormals);
    std::vector<std::shared_ptr<Node>> seeds;
    const float dg_w = 3 * dynfuParams.epsilon;  // :158
    for (size_t i = 0; i < canonicalVertices.size(); i += (size_t)nodeStep)
        seeds.push_back(std::make_shared<Node>(
            canonicalVertices[i], std::make_shared<DualQuaternion<float>>(0.f, 0.f, 0.f, 0.f, 0.f, 0.f), dg_w));
    warpfield = std::make_shared<Warpfield>();
    warpfield->init(dynfuParams.epsilon, seeds);
}

void DynFusion::initCanonicalFrame(dfa::PointCloud<dfa::PointXYZ>& vertices, dfa::PointCloud<dfa::Normal>& normals) {
    canonicalFrame             = std::make_shared<dynfu::Frame>(0, vertices, normals);
    canonicalFrameWarpedToLive = std::make_shared<dynfu::Frame>(0, vertices, normals);
}

void DynFusion::addLiveFrame(int frameID, dfa::PointCloud<dfa::PointXYZ>& vertices,
                             dfa::PointCloud<dfa::Normal>& normals) {
    liveFrame = std::make_shared<dynfu::Frame>(frameID, vertices, normals);
}

namespace {
// DFA_HOST_PROFILE=1: wall time of the stages of a frame (device synchronised at every mark) on stderr
struct StageClock {
    bool on = std::getenv("DFA_HOST_PROFILE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[dfa host] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};
}  // namespace

void DynFusion::warpCanonicalToLiveOpt(dfa::Affine3f affine) {
    if (!warpfield || !canonicalFrame || !liveFrame)
        throw dfa::Error(DFA_ERR_INVALID, "warpCanonicalToLiveOpt before init / addLiveFrame");
    StageClock clk;
    CombinedSolver combinedSolver(*warpfield, solverParams, dynfuParams.tukeyOffset, dynfuParams.psi_data,
                                  dynfuParams.lambda, dynfuParams.psi_reg);
    clk.mark("  CombinedSolver ctor");
    canonicalFrameWarpedToLive = warpfield->warpToLive(canonicalFrame);  // :196
    clk.mark("  warpToLive");
    auto corresponding         = findCorrespondingFrame(canonicalFrameWarpedToLive->getVertices(),
                                                        canonicalFrameWarpedToLive->getNormals(), liveFrame->getVertices());
    clk.mark("  findCorrespondingFrame");
    combinedSolver.initializeProblemInstance(corresponding, liveFrame, affine);  // :206
    clk.mark("  initializeProblemInstance");
    combinedSolver.solveAll();                                                   // :207
    clk.mark("  solveAll");
}

std::shared_ptr<dynfu::Frame> DynFusion::findCorrespondingFrame(dfa::PointCloud<dfa::PointXYZ> canonicalVertices,
                                                                dfa::PointCloud<dfa::Normal> canonicalNormals,
                                                                dfa::PointCloud<dfa::PointXYZ> liveVertices) {
    const size_t nc = canonicalVertices.size(), nl = liveVertices.size();
    dfa::PointCloud<dfa::PointXYZ> outV;
    dfa::PointCloud<dfa::Normal> outN;
    if (nl == 0) return std::make_shared<dynfu::Frame>(0, outV, outN);
    if (nc == 0) throw dfa::Error(DFA_ERR_INVALID, "findCorrespondingFrame: empty canonical cloud");
    std::vector<float> cv(3 * nc), cn(3 * nc), lv(3 * nl);
    for (size_t i = 0; i < nc; ++i) {
        cv[3 * i] = canonicalVertices[i].x, cv[3 * i + 1] = canonicalVertices[i].y, cv[3 * i + 2] = canonicalVertices[i].z;
        const dfa::Normal n = i < canonicalNormals.size() ? canonicalNormals[i] : dfa::Normal();
        cn[3 * i] = n.normal_x, cn[3 * i + 1] = n.normal_y, cn[3 * i + 2] = n.normal_z;
    }
    for (size_t i = 0; i < nl; ++i)
        lv[3 * i] = liveVertices[i].x, lv[3 * i + 1] = liveVertices[i].y, lv[3 * i + 2] = liveVertices[i].z;
    dfa::DeviceArray<float> dcv, dcn, dlv, dov(3 * nl), don(3 * nl);
    dcv.upload(cv), dcn.upload(cn), dlv.upload(lv);
    dfa::check(dfa_correspond(dcv.ptr(), dcn.ptr(), (int)nc, dlv.ptr(), (int)nl, dov.ptr(), don.ptr(), nullptr, nullptr),
               "DynFusion::findCorrespondingFrame");
    std::vector<float> ov, on;
    dov.download(ov), don.download(on);
    for (size_t i = 0; i < nl; ++i) {
        outV.push_back(dfa::PointXYZ(ov[3 * i], ov[3 * i + 1], ov[3 * i + 2]));
        outN.push_back(dfa::Normal(on[3 * i], on[3 * i + 1], on[3 * i + 2]));
    }
    return std::make_shared<dynfu::Frame>(0, outV, outN);
}

std::shared_ptr<dynfu::Frame> DynFusion::getCanonicalWarpedToLive() { return canonicalFrameWarpedToLive; }

void DynFusion::fuse(const kfusion::cuda::Depth& depth, kfusion::cuda::TsdfVolume& volume,
                     const dfa::Affine3f& camera_pose) {
    kfusion::cuda::computeDists(depth, dists_, dynfuParams.intr);       // :58
    volume.clearAndIntegrate(dists_, camera_pose, dynfuParams.intr);   // :113-114
}

// ------------------------------------------------------------------------- the per-frame sequence

kfusion::cuda::TsdfVolume& DynFusion::tsdf() {
    if (!volume_) {  // KinFu::KinFu, src/kfusion/kinfu.cpp:47-58
        const kfusion::KinFuParams& p = dynfuParams.kinfuParams;
        volume_ = std::make_shared<kfusion::cuda::TsdfVolume>(p.volume_dims);
        volume_->setTruncDist(p.tsdf_trunc_dist);
        volume_->setMaxWeight(p.tsdf_max_weight);
        volume_->setSize(p.volume_size);
        volume_->setPose(p.volume_pose);
        volume_->setRaycastStepFactor(p.raycast_step_factor);
        volume_->setGradientDeltaFactor(p.gradient_delta_factor);
    }
    return *volume_;
}

void DynFusion::extractSurface(dfa::PointCloud<dfa::PointXYZ>& vertices, dfa::PointCloud<dfa::Normal>& normals) {
    if (!mc_) mc_ = std::make_shared<kfusion::cuda::MarchingCubes>();
    dfa::DeviceArray<kfusion::cuda::MarchingCubes::PointType> buffer;
    auto triangles = mc_->run(tsdf(), buffer);  // :73-75 / :119-121
    std::vector<kfusion::cuda::MarchingCubes::PointType> host;
    if (!triangles.empty()) triangles.download(host);
    vertices.points = host;                          // PointType is dfa::PointXYZ: {x, y, z, 1}
    mesh_triangles_ = std::move(host), mesh_.reset();  // convertToMesh (:76 / :122) on demand: getMesh()
    const size_t nv = vertices.size();
    if (dynfuParams.mesh_normals && nv) {  // extension: gradient of the TSDF at the vertices
        dfa::DeviceArray<dfa::Normal> dn;
        mc_->computeNormals(tsdf(), triangles, dn);
        std::vector<dfa::Normal> hn;
        dn.download(hn);
        normals.points = std::move(hn);
        return;
    }
    // pcl::copyPointCloud<PointXYZ, Normal> (:87-88 / :133-134) copies the fields the two types share — none:
    // the normals are default-constructed, one per vertex
    normals.points.assign(nv, dfa::Normal());
}

bool DynFusion::operator()(const kfusion::cuda::Depth& depth) {
    const kfusion::KinFuParams& p = dynfuParams.kinfuParams;
    kfusion::cuda::computeDists(depth, dists_, p.intr);                                            // :58
    kfusion::cuda::depthBilateralFilter(depth, depth_filtered_, p.bilateral_kernel_size, p.bilateral_sigma_spatial,
                                        p.bilateral_sigma_depth);                                  // :60-61
    if (p.icp_truncate_depth_dist > 0) kfusion::cuda::depthTruncation(depth_filtered_, p.icp_truncate_depth_dist);  // :64-66
    const dfa::Affine3f camera;  // poses_.back(): the rigid tracker is skipped, the camera stays at the origin (:100-105)
    dfa::PointCloud<dfa::PointXYZ> vertices;
    dfa::PointCloud<dfa::Normal> normals;
    if (frame_counter_ == 0) {
        tsdf().integrate(dists_, camera, p.intr);  // :71
        extractSurface(vertices, normals);
        init(vertices, normals);                   // :95
        return ++frame_counter_, false;
    }
    StageClock clk;
    tsdf().clearAndIntegrate(dists_, camera, p.intr);  // :113-114 as one sweep
    clk.mark("pre-process + fuse");
    extractSurface(vertices, normals);
    clk.mark("marching cubes -> host clouds");
    addLiveFrame(frame_counter_, vertices, normals);  // :137
    clk.mark("addLiveFrame");
    warpCanonicalToLiveOpt(camera);                   // :140
    clk.mark("warpCanonicalToLiveOpt");
    warpfield->update(getCanonicalWarpedToLive());    // :142
    clk.mark("warpfield->update");
    return ++frame_counter_, true;
}

SequenceReport runSequence(DynFusion& dynfu, const std::string& dir, int max_frames) {
    const dfa::io::SequenceFiles files = dfa::io::listSequence(dir);  // demo.cpp:39-55
    if (files.images.size() < files.depths.size())
        throw dfa::Error(1, "fewer colour images than depth frames in " + dir);  // the demo indexes images[i]
    const std::string out = dir + "/out";                                      // demo.cpp:58-66
    std::filesystem::create_directory(out);
    SequenceReport rep;
    kfusion::cuda::Depth depth_device;
    const size_t n = max_frames < 0 ? files.depths.size() : std::min(files.depths.size(), (size_t)max_frames);
    for (size_t i = 0; i < n; ++i) {
        const dfa::io::DepthImage depth = dfa::io::readDepthPng(files.depths[i]);  // demo.cpp:81
        depth_device.upload(depth.data.data(), (size_t)depth.cols * sizeof(uint16_t), depth.rows, depth.cols);  // :90
        const auto t0        = std::chrono::steady_clock::now();
        const bool has_image = dynfu(depth_device);  // :94
        rep.dynfu_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ++rep.frames;
        if (has_image) {  // :115-118
            dfa::io::savePCDFileASCII(out + "/pcl_canonical_to_live" + std::to_string(i) + ".pcd",
                                      dynfu.getCanonicalWarpedToLive()->getVertices());
            dfa::io::saveVTKFile(out + "/" + std::to_string(i) + "_tsdf_mesh.vtk", *dynfu.getMesh());
            ++rep.saved;
        }
    }
    return rep;
}
