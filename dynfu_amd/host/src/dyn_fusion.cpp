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

DynFusion::DynFusion(const DynFuParams& params) : kfusion::KinFu(params.kinfuParams), dynfuParams(params) {  // dyn_fusion.cpp:33
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
    seedNodes(canonicalVertices.points);
}

// :147-168 — a node at every nodeStep-th canonical vertex
void DynFusion::seedNodes(const std::vector<dfa::PointXYZ>& canonicalVertices) {
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

// frame 0 of operator(): the canonical frame is the device-resident marching-cubes cloud; only the seeding reads it
// on the host (once, through the const accessor: the device arrays stay the master)
void DynFusion::initFromFrame(std::shared_ptr<dynfu::Frame> frame) {
    dfa::DeviceArray<float> v3, n3;
    frame->deviceArrays(v3, n3);
    canonicalFrame             = frame;
    canonicalFrameWarpedToLive = dynfu::Frame::fromDevice(0, v3, n3, frame->size());  // a second Frame, as :171-175
    seedNodes(frame->vertices().points);
}

void DynFusion::addLiveFrame(int frameID, std::shared_ptr<dynfu::Frame> frame) {
    (void)frameID;
    liveFrame = frame;
}

void DynFusion::addLiveFrame(int frameID, dfa::PointCloud<dfa::PointXYZ>& vertices,
                             dfa::PointCloud<dfa::Normal>& normals) {
    liveFrame = std::make_shared<dynfu::Frame>(frameID, vertices, normals);
}

namespace {
// DFA_HOST_PROFILE=1: wall time of the stages of a frame (device synchronised at every mark) on stderr
struct StageClock {
    bool on = dfa::host_switch("DFA_HOST_PROFILE");
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
    auto corresponding         = findCorrespondingFrame(canonicalFrameWarpedToLive, liveFrame);
    clk.mark("  findCorrespondingFrame");
    combinedSolver.initializeProblemInstance(corresponding, liveFrame, affine);  // :206
    clk.mark("  initializeProblemInstance");
    combinedSolver.solveAll();                                                   // :207
    clk.mark("  solveAll");
}

std::shared_ptr<dynfu::Frame> DynFusion::findCorrespondingFrame(dfa::PointCloud<dfa::PointXYZ> canonicalVertices,
                                                                dfa::PointCloud<dfa::Normal> canonicalNormals,
                                                                dfa::PointCloud<dfa::PointXYZ> liveVertices) {
    return findCorrespondingFrame(std::make_shared<dynfu::Frame>(0, std::move(canonicalVertices), std::move(canonicalNormals)),
                                  std::make_shared<dynfu::Frame>(0, std::move(liveVertices), dfa::PointCloud<dfa::Normal>()));
}

// :212-242 on frames: both clouds are read where they are (HBM for the adaptor's own frames), the result stays there
std::shared_ptr<dynfu::Frame> DynFusion::findCorrespondingFrame(std::shared_ptr<dynfu::Frame> canonical,
                                                                std::shared_ptr<dynfu::Frame> live) {
    const size_t nc = canonical->size(), nl = live->size();
    if (nl == 0) return std::make_shared<dynfu::Frame>(0, dfa::PointCloud<dfa::PointXYZ>(), dfa::PointCloud<dfa::Normal>());
    if (nc == 0) throw dfa::Error(DFA_ERR_INVALID, "findCorrespondingFrame: empty canonical cloud");
    const dynfu::Frame::DeviceView c = canonical->device(), l = live->device();
    dfa::DeviceArray<float> dov(3 * nl), don(3 * nl);
    dfa::check(dfa_correspond(c.vertices, c.normals, (int)nc, l.vertices, (int)nl, dov.ptr(), don.ptr(), nullptr, nullptr),
               "DynFusion::findCorrespondingFrame");
    return dynfu::Frame::fromDevice(0, dov, don, nl);
}

std::shared_ptr<dynfu::Frame> DynFusion::getCanonicalWarpedToLive() { return canonicalFrameWarpedToLive; }

void DynFusion::fuse(const kfusion::cuda::Depth& depth, kfusion::cuda::TsdfVolume& volume,
                     const dfa::Affine3f& camera_pose) {
    kfusion::cuda::computeDists(depth, dists_, dynfuParams.intr);       // :58
    volume.clearAndIntegrate(dists_, camera_pose, dynfuParams.intr);   // :113-114
}

// ------------------------------------------------------------------------- the per-frame sequence

std::shared_ptr<dynfu::Frame> DynFusion::extractSurface(int frame_id, bool with_normals) {
    auto triangles = mc_->run(tsdf(), mc_buffer_);  // :73-75 / :119-121 (one host sync: the vertex count)
    // convertToMesh (:76 / :122) on demand, in getMesh(): the float4 triangle soup stays in mc_buffer_ until then
    mesh_source_ = triangles, mesh_.reset(), mesh_triangles_.clear(), mesh_downloaded_ = false;
    const size_t nv = triangles.size();
    if (nv == 0) return std::make_shared<dynfu::Frame>(frame_id, dfa::PointCloud<dfa::PointXYZ>(), dfa::PointCloud<dfa::Normal>());
    // pcl::fromPCLPointCloud2 / copyPointCloud (:80-88): the vertices as a cloud of their own — packed N x 3 in HBM
    dfa::DeviceArray<float> v3(3 * nv), n3(3 * nv);
    dfa::check(dfa_repack_points((const float*)triangles.ptr(), 4, v3.ptr(), 3, (int)nv, 0.f, nullptr), "DynFusion: vertices");
    if (with_normals) {  // extension: gradient of the TSDF at the vertices
        dfa::DeviceArray<dfa::Normal> dn;
        mc_->computeNormals(tsdf(), triangles, dn);
        dfa::check(dfa_repack_points((const float*)dn.ptr(), 4, n3.ptr(), 3, (int)nv, 0.f, nullptr), "DynFusion: normals");
    } else {
        // pcl::copyPointCloud<PointXYZ, Normal> (:87-88 / :133-134) copies the fields the two types share — none:
        // the normals are default-constructed, one per vertex
        if (hipMemsetAsync(n3.ptr(), 0, 3 * nv * sizeof(float), nullptr) != hipSuccess)
            throw dfa::Error(DFA_ERR_HIP, "DynFusion::extractSurface: hipMemsetAsync");
    }
    return dynfu::Frame::fromDevice(frame_id, v3, n3, nv);
}

std::shared_ptr<dfa::PolygonMesh> DynFusion::getMesh() {  // KinFu::getMesh (kinfu.cpp:262)
    if (!mesh_) {
        if (!mesh_downloaded_) {
            mesh_triangles_.clear();
            if (!mesh_source_.empty()) mesh_source_.download(mesh_triangles_);
            mesh_downloaded_ = true;
        }
        mesh_ = std::make_shared<dfa::PolygonMesh>(dfa::convertToMesh(mesh_triangles_));
    }
    return mesh_;
}

bool DynFusion::operator()(const kfusion::cuda::Depth& depth) {
    const kfusion::KinFuParams& p = params_;                                                       // :51
    kfusion::cuda::Depth& depth_filtered_ = curr_.depth_pyr[0];
    kfusion::cuda::computeDists(depth, dists_, p.intr);                                            // :58
    kfusion::cuda::depthBilateralFilter(depth, depth_filtered_, p.bilateral_kernel_size, p.bilateral_sigma_spatial,
                                        p.bilateral_sigma_depth);                                  // :60-61
    if (p.icp_truncate_depth_dist > 0) kfusion::cuda::depthTruncation(depth_filtered_, p.icp_truncate_depth_dist);  // :64-66
    if (dynfuParams.north_star) return northStarFrame(depth);
    const dfa::Affine3f camera;  // poses_.back(): the rigid tracker is skipped, the camera stays at the origin (:100-105)
    if (frame_counter_ == 0) {
        tsdf().integrate(dists_, camera, p.intr);  // :71
        initFromFrame(extractSurface(0, dynfuParams.mesh_normals));  // :73-95
        return ++frame_counter_, false;
    }
    StageClock clk;
    tsdf().clearAndIntegrate(dists_, camera, p.intr);  // :113-114 as one sweep
    clk.mark("pre-process + fuse");
    auto live = extractSurface(frame_counter_, dynfuParams.mesh_normals);
    clk.mark("marching cubes -> live frame");
    addLiveFrame(frame_counter_, live);  // :137
    warpCanonicalToLiveOpt(camera);      // :140
    clk.mark("warpCanonicalToLiveOpt");
    warpfield->update(getCanonicalWarpedToLive());    // :142
    clk.mark("warpfield->update");
    return ++frame_counter_, true;
}

// The same frame sequence with the north-star solve in the middle: the live side of the solve is the depth frame itself
// (vertex / normal maps of the filtered depth, as KinFu builds them for its tracker, kinfu.cpp:150-175), so neither the
// live marching-cubes cloud nor the nearest-neighbour correspondence is on the path; everything the solve sees is in
// the camera frame.
bool DynFusion::northStarFrame(const kfusion::cuda::Depth& depth) {
    const kfusion::KinFuParams& p = params_;
    kfusion::cuda::Depth& depth_filtered_ = curr_.depth_pyr[0];
    const dfa::Affine3f camera;  // the camera stays at the origin, as in the reference's operator() (:100-105)
    if (frame_counter_ == 0) {
        tsdf().integrate(dists_, camera, p.intr);
        auto vol_frame = extractSurface(0, true);  // volume frame, normals from the TSDF gradient
        const size_t n = vol_frame->size();
        if (n == 0) throw dfa::Error(DFA_ERR_INVALID, "DynFusion (north-star mode): the first frame has no surface");
        // volume frame -> camera frame: camera_pose^-1 * volume_pose, the inverse of what integrate() applies (tsdf_volume.cpp:83)
        float aff[12];
        (camera.inv() * tsdf().getPose()).to12(aff);
        const dynfu::Frame::DeviceView v = vol_frame->device();
        dfa::DeviceArray<float> cv(3 * n), cn(3 * n);
        dfa::check(dfa_transform_points(v.vertices, (int)n, aff, 1, cv.ptr(), nullptr), "DynFusion: canonical vertices to the camera frame");
        dfa::check(dfa_transform_points(v.normals, (int)n, aff, 0, cn.ptr(), nullptr), "DynFusion: canonical normals to the camera frame");
        initFromFrame(dynfu::Frame::fromDevice(0, cv, cn, n));
        return ++frame_counter_, false;
    }
    StageClock clk;
    tsdf().clearAndIntegrate(dists_, camera, p.intr);
    addLiveFrame(frame_counter_, extractSurface(frame_counter_, dynfuParams.mesh_normals));  // for getLiveFrame() / getMesh()
    clk.mark("pre-process + fuse + marching cubes");
    NorthStarSolver solver(*warpfield, dynfuParams.northStarParams, dynfuParams.tukeyOffset, dynfuParams.psi_data,
                           dynfuParams.lambda, dynfuParams.psi_reg);
    solver.initializeProblemInstance(canonicalFrame);
    clk.mark("  initializeProblemInstance");
    kfusion::cuda::computePointNormals(p.intr, depth_filtered_, live_points_, live_normals_);
    solver.solveAll(live_points_, live_normals_, p.intr);
    clk.mark("  solveAll");
    ns_initial_cost_ = solver.initialCost(), ns_final_cost_ = solver.finalCost(), ns_valid_rows_ = solver.validRows();
    canonicalFrameWarpedToLive = solver.warpCanonicalToLive();
    warpfield->update(canonicalFrameWarpedToLive);
    clk.mark("warp + warpfield->update");
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
        rep.frame_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        rep.dynfu_ms += rep.frame_ms.back();
        ++rep.frames;
        if (has_image) {  // :115-118
            dfa::io::savePCDFileASCII(out + "/pcl_canonical_to_live" + std::to_string(i) + ".pcd",
                                      dynfu.getCanonicalWarpedToLive()->vertices());
            dfa::io::saveVTKFile(out + "/" + std::to_string(i) + "_tsdf_mesh.vtk", *dynfu.getMesh());
            if (i + 1 < n) dynfu.dropMesh();  // the demo's mesh pointer goes out of scope here, outside the timed call
            ++rep.saved;
        }
    }
    return rep;
}
