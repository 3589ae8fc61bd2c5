// dyn_fusion.cpp — DynFusion's warp-field sequence on the dynfu_amd C ABI
// (reference: src/dynfu/dyn_fusion.cpp:6-31,147-242).
#include <dynfu/dyn_fusion.hpp>

#include <dfa_host/device.hpp>

#include "../../../include/dynfu_amd.h"

DynFuParams DynFuParams::defaultParams() {  // dyn_fusion.cpp:6-31, kinfu.cpp:16-18
    DynFuParams p;
    p.intr        = kfusion::Intr(525.f, 525.f, 319.5f, 239.5f);
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

void DynFusion::warpCanonicalToLiveOpt(dfa::Affine3f affine) {
    if (!warpfield || !canonicalFrame || !liveFrame)
        throw dfa::Error(DFA_ERR_INVALID, "warpCanonicalToLiveOpt before init / addLiveFrame");
    CombinedSolver combinedSolver(*warpfield, solverParams, dynfuParams.tukeyOffset, dynfuParams.psi_data,
                                  dynfuParams.lambda, dynfuParams.psi_reg);
    canonicalFrameWarpedToLive = warpfield->warpToLive(canonicalFrame);  // :196
    auto corresponding         = findCorrespondingFrame(canonicalFrameWarpedToLive->getVertices(),
                                                        canonicalFrameWarpedToLive->getNormals(), liveFrame->getVertices());
    combinedSolver.initializeProblemInstance(corresponding, liveFrame, affine);  // :206
    combinedSolver.solveAll();                                                   // :207
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
