// warp_field.cpp — Warpfield on the dynfu_amd C ABI (reference: src/dynfu/warp_field.cpp).
#include <dynfu/warp_field.hpp>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include <dfa_host/device.hpp>

#include "../../../include/dynfu_amd.h"

struct Warpfield::DeviceNodes {
    dfa::DeviceArray<float> pos, w, dq;  // D x 3, D, D x 8
    int D = 0;
    // neighbours of the cloud warpToLive saw last: every frame of a sequence warps the SAME canonical cloud
    // (dyn_fusion.cpp:196) and the node set only ever grows, so while the cloud's device array, its size and the node
    // count are the ones of the call before, the k-NN search of that call still holds (DFA_HOST_NO_GRAPH_REUSE=1: search
    // every time)
    dfa::DeviceArray<float> knn_verts;  // (kept alive: its address identifies the cloud)
    dfa::DeviceArray<int32_t> knn_idx;
    size_t knn_N = 0;
    int knn_D = 0, knn_k = 0;
};

Warpfield::Warpfield()  = default;
Warpfield::~Warpfield() = default;

void Warpfield::init(float epsilon_, std::vector<std::shared_ptr<Node>> nodes_) {  // warp_field.cpp:10-28
    epsilon = epsilon_;
    nodes   = std::make_shared<NodeList>(std::move(nodes_));
    dev     = std::make_shared<DeviceNodes>();
    syncPositions();
}

void Warpfield::addNode(std::shared_ptr<Node> newNode) {
    ownNodes().emplace_back(newNode);
    syncPositions();
}
std::vector<std::shared_ptr<Node>> Warpfield::getNodes() { return *nodes; }

// node positions and radial basis weights never change after construction: uploaded once
void Warpfield::syncPositions() {
    const NodeList& nodes = *this->nodes;
    std::vector<float> p(3 * nodes.size()), w(nodes.size());
    for (size_t i = 0; i < nodes.size(); ++i) {
        const dfa::PointXYZ g = nodes[i]->getPosition();
        p[3 * i] = g.x, p[3 * i + 1] = g.y, p[3 * i + 2] = g.z;
        w[i] = nodes[i]->getRadialBasisWeight();
    }
    if (!dev) dev = std::make_shared<DeviceNodes>();
    dev->D = (int)nodes.size();
    if (dev->D) {
        dev->pos.upload(p);
        dev->w.upload(w);
    }
}

Warpfield::DeviceNodeView Warpfield::deviceNodes(bool refresh_transforms) {
    if (!dev || dev->D == 0) return DeviceNodeView{nullptr, nullptr, nullptr, 0};
    if (refresh_transforms) syncTransforms();
    return DeviceNodeView{dev->pos.ptr(), dev->w.ptr(), dev->dq.ptr(), dev->D};
}

std::vector<size_t> Warpfield::findNeighborsIndex(int numNeighbor, dfa::PointXYZ vertex) {  // :111-122
    std::vector<size_t> out;
    if (!dev || dev->D == 0) return out;
    dfa::DeviceArray<float> q(3);
    dfa::DeviceArray<int32_t> idx(numNeighbor);
    const float host_q[3] = {vertex.x, vertex.y, vertex.z};
    q.DeviceMemory::upload(host_q, sizeof(host_q));
    dfa::check(dfa_knn(dev->pos.ptr(), dev->w.ptr(), dev->D, q.ptr(), 1, numNeighbor, idx.ptr(), nullptr, nullptr),
               "Warpfield::findNeighborsIndex");
    std::vector<int32_t> h;
    idx.download(h);
    for (int32_t i : h)
        if (i >= 0) out.push_back((size_t)i);  // n < k results when there are fewer than k nodes
    return out;
}

std::vector<std::shared_ptr<Node>> Warpfield::findNeighbors(int numNeighbor, dfa::PointXYZ vertex) {  // :99-109
    std::vector<std::shared_ptr<Node>> out;
    for (size_t i : findNeighborsIndex(numNeighbor, vertex)) out.push_back((*nodes)[i]);
    return out;
}

// :127-148 — ordered product of the neighbours' dual-scaled transforms, real part normalised
std::shared_ptr<DualQuaternion<float>> Warpfield::calcDQB(dfa::PointXYZ point) {
    DualQuaternion<float> sum(0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
    for (auto& node : findNeighbors(knn_, point)) {
        const float w = node->getTransformationWeight(point);
        sum *= (*node->getTransformation() * w);
    }
    return std::make_shared<DualQuaternion<float>>(sum.normalize());
}

static void pack_transforms(const std::vector<std::shared_ptr<Node>>& nodes, std::vector<float>& hdq) {
    hdq.resize(8 * nodes.size());
    for (size_t i = 0; i < nodes.size(); ++i) {
        const auto& dq = *nodes[i]->getTransformation();
        const auto r = dq.getReal(), d = dq.getDual();
        float* o = &hdq[8 * i];
        o[0] = r.a, o[1] = r.b, o[2] = r.c, o[3] = r.d, o[4] = d.a, o[5] = d.b, o[6] = d.c, o[7] = d.d;
    }
}

// the nodes' current transforms dg_se3 as D x 8 floats (real w,x,y,z ; dual w,x,y,z) -> dev->dq
void Warpfield::syncTransforms() {
    std::vector<float> hdq;
    pack_transforms(*nodes, hdq);
    if (!hdq.empty()) dev->dq.upload(hdq);
}

void Warpfield::hostArrays(std::vector<float>& pos, std::vector<float>& w, std::vector<float>& dq) {
    const NodeList& nodes = *this->nodes;
    pos.resize(3 * nodes.size()), w.resize(nodes.size());
    for (size_t i = 0; i < nodes.size(); ++i) {
        const dfa::PointXYZ g = nodes[i]->getPosition();
        pos[3 * i] = g.x, pos[3 * i + 1] = g.y, pos[3 * i + 2] = g.z;
        w[i] = nodes[i]->getRadialBasisWeight();
    }
    pack_transforms(nodes, dq);
}

// :150-171 — bulk warp on the GPU; the warped frame stays in HBM until somebody asks for its clouds
std::shared_ptr<dynfu::Frame> Warpfield::warpToLive(std::shared_ptr<dynfu::Frame> canonicalFrame) {
    const size_t N = canonicalFrame->size();
    if (N == 0 || !dev || dev->D == 0)
        return std::make_shared<dynfu::Frame>(0, dfa::PointCloud<dfa::PointXYZ>(), dfa::PointCloud<dfa::Normal>());
    syncTransforms();
    dfa::DeviceArray<float> ov(3 * N), on(3 * N);
    if (canonicalFrame->deviceResident() && !dfa::host_switch("DFA_HOST_NO_GRAPH_REUSE")) {
        // a device-resident frame never changes its arrays: their address names the cloud
        dfa::DeviceArray<float> v3, n3;
        canonicalFrame->deviceArrays(v3, n3);
        DeviceNodes& d = *dev;
        if (!(d.knn_verts.ptr() == v3.ptr() && d.knn_N == N && d.knn_D == d.D && d.knn_k == knn_)) {
            d.knn_idx = dfa::DeviceArray<int32_t>(N * (size_t)knn_);
            dfa::check(dfa_knn(d.pos.ptr(), d.w.ptr(), d.D, v3.ptr(), (int)N, knn_, d.knn_idx.ptr(), nullptr, nullptr),
                       "Warpfield::warpToLive (k-NN)");
            d.knn_verts = v3, d.knn_N = N, d.knn_D = d.D, d.knn_k = knn_;
        }
        dfa::check(dfa_warp_to_live_graph(d.pos.ptr(), d.dq.ptr(), d.w.ptr(), d.D, knn_, d.knn_idx.ptr(), v3.ptr(), n3.ptr(),
                                          (int)N, ov.ptr(), on.ptr(), nullptr),
                   "Warpfield::warpToLive");
        return dynfu::Frame::fromDevice(0, ov, on, N);
    }
    const dynfu::Frame::DeviceView in = canonicalFrame->device();
    dfa::check(dfa_warp_to_live(dev->pos.ptr(), dev->dq.ptr(), dev->w.ptr(), dev->D, knn_, in.vertices, in.normals, (int)N,
                                ov.ptr(), on.ptr(), nullptr),
               "Warpfield::warpToLive");
    return dynfu::Frame::fromDevice(0, ov, on, N);
}

// ------------------------------------------------------------------------------- node insertion

dfa::PointCloud<dfa::PointXYZ> dfa::voxelGridFilter(const dfa::PointCloud<dfa::PointXYZ>& cloud, float leaf) {
    dfa::PointCloud<dfa::PointXYZ> out;
    const float inv = 1.0f / leaf;
    float lo[3] = {HUGE_VALF, HUGE_VALF, HUGE_VALF}, hi[3] = {-HUGE_VALF, -HUGE_VALF, -HUGE_VALF};
    std::vector<int> keep;
    for (size_t i = 0; i < cloud.size(); ++i) {
        const float p[3] = {cloud[i].x, cloud[i].y, cloud[i].z};
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        keep.push_back((int)i);
        for (int c = 0; c < 3; ++c) lo[c] = std::min(lo[c], p[c]), hi[c] = std::max(hi[c], p[c]);
    }
    if (keep.empty()) return out;
    long min_b[3], div_b[3];
    for (int c = 0; c < 3; ++c) {
        min_b[c] = (long)std::floor(lo[c] * inv);
        div_b[c] = (long)std::floor(hi[c] * inv) - min_b[c] + 1;
    }
    std::vector<std::pair<long, int>> cells;  // (leaf index, point)
    cells.reserve(keep.size());
    for (int i : keep) {
        const long i0 = (long)std::floor(cloud[i].x * inv) - min_b[0], i1 = (long)std::floor(cloud[i].y * inv) - min_b[1],
                   i2 = (long)std::floor(cloud[i].z * inv) - min_b[2];
        cells.emplace_back(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1], i);
    }
    std::sort(cells.begin(), cells.end());  // by leaf, then by input order
    for (size_t i = 0; i < cells.size();) {
        size_t j = i;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (; j < cells.size() && cells[j].first == cells[i].first; ++j)
            sx += cloud[cells[j].second].x, sy += cloud[cells[j].second].y, sz += cloud[cells[j].second].z;
        const float n = (float)(j - i);
        out.push_back(dfa::PointXYZ(sx / n, sy / n, sz / n));
        i = j;
    }
    return out;
}

dfa::PointCloud<dfa::PointXYZ> Warpfield::getUnsupportedVertices(std::shared_ptr<dynfu::Frame> frame) {  // :34-62
    dfa::PointCloud<dfa::PointXYZ> out;
    const size_t N = frame->size();
    if (N == 0) return out;
    const dynfu::Frame::DeviceView in = frame->device();
    dfa::DeviceArray<unsigned char> flags(N);
    dfa::DeviceArray<float> picked(3 * N);
    dfa::DeviceArray<int32_t> count(1);
    const int D = dev ? dev->D : 0;
    dfa::check(dfa_unsupported_vertices(D ? dev->pos.ptr() : nullptr, D ? dev->w.ptr() : nullptr, D, knn_, in.vertices,
                                        (int)N, flags.ptr(), nullptr),
               "Warpfield::getUnsupportedVertices");
    // the reference's push_back loop (:42-59) as an order-preserving compaction on the device: only the unsupported
    // vertices (usually a handful) cross PCIe
    dfa::check(dfa_compact_points(in.vertices, flags.ptr(), (int)N, picked.ptr(), nullptr, count.ptr(), nullptr),
               "Warpfield::getUnsupportedVertices (compaction)");
    std::vector<int32_t> n_host;
    count.download(n_host);
    const size_t m = (size_t)n_host[0];
    if (m == 0) return out;
    std::vector<float> h(3 * m);
    picked.DeviceMemory::download(h.data(), h.size() * sizeof(float));
    out.points.resize(m);
    for (size_t i = 0; i < m; ++i) out.points[i] = dfa::PointXYZ(h[3 * i], h[3 * i + 1], h[3 * i + 2]);
    return out;
}

void Warpfield::update(std::shared_ptr<dynfu::Frame> frame) {  // :64-95
    const dfa::PointCloud<dfa::PointXYZ> unsupported = getUnsupportedVertices(frame);
    const dfa::PointCloud<dfa::PointXYZ> seeds       = dfa::voxelGridFilter(unsupported, 0.05f);  // :68-72
    const int n = (int)seeds.size();
    if (n == 0) return;
    // dg_se3 of a new node = calcDQB(dg_v) over the nodes that existed BEFORE this update (the reference's KD-tree
    // is rebuilt only after the loop, :85-94); with no node yet calcDQB is the identity
    std::vector<float> hq(8 * (size_t)n, 0.f);
    for (int i = 0; i < n; ++i) hq[8 * (size_t)i] = 1.f;
    if (dev && dev->D > 0) {
        std::vector<float> hp(3 * (size_t)n);
        for (int i = 0; i < n; ++i) hp[3 * i] = seeds[i].x, hp[3 * i + 1] = seeds[i].y, hp[3 * i + 2] = seeds[i].z;
        dfa::DeviceArray<float> dp, dq_out(8 * (size_t)n);
        dp.upload(hp);
        syncTransforms();
        dfa::check(dfa_calc_dqb(dev->pos.ptr(), dev->dq.ptr(), dev->w.ptr(), dev->D, knn_, dp.ptr(), n, dq_out.ptr(), nullptr),
                   "Warpfield::update (calcDQB)");
        dq_out.download(hq);
    }
    NodeList& list = ownNodes();  // (a copy of this warp field held elsewhere keeps the list it was copied with)
    for (int i = 0; i < n; ++i) {
        const float* q = &hq[8 * (size_t)i];
        auto dq = std::make_shared<DualQuaternion<float>>(dfa::quaternion<float>(q[0], q[1], q[2], q[3]),
                                                          dfa::quaternion<float>(q[4], q[5], q[6], q[7]));
        list.emplace_back(std::make_shared<Node>(seeds[i], dq, 2 * epsilon));  // :79-82
    }
    syncPositions();  // :85-94
}
