// warp_field.cpp — Warpfield on the dynfu_amd C ABI (reference: src/dynfu/warp_field.cpp).
#include <dynfu/warp_field.hpp>

#include <dfa_host/device.hpp>

#include "../../../include/dynfu_amd.h"

struct Warpfield::DeviceNodes {
    dfa::DeviceArray<float> pos, w, dq;  // D x 3, D, D x 8
    int D = 0;
};

Warpfield::Warpfield()  = default;
Warpfield::~Warpfield() = default;

void Warpfield::init(float epsilon_, std::vector<std::shared_ptr<Node>> nodes_) {  // warp_field.cpp:10-28
    epsilon = epsilon_;
    nodes   = nodes_;
    dev     = std::make_shared<DeviceNodes>();
    syncPositions();
}

void Warpfield::addNode(std::shared_ptr<Node> newNode) {
    nodes.emplace_back(newNode);
    syncPositions();
}
std::vector<std::shared_ptr<Node>> Warpfield::getNodes() { return nodes; }

// node positions and radial basis weights never change after construction: uploaded once
void Warpfield::syncPositions() {
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
    for (size_t i : findNeighborsIndex(numNeighbor, vertex)) out.push_back(nodes[i]);
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

// :150-171 — bulk warp on the GPU
std::shared_ptr<dynfu::Frame> Warpfield::warpToLive(std::shared_ptr<dynfu::Frame> canonicalFrame) {
    auto& verts   = canonicalFrame->getVertices();
    auto& normals = canonicalFrame->getNormals();
    const int N   = (int)verts.size();
    dfa::PointCloud<dfa::PointXYZ> wv;
    dfa::PointCloud<dfa::Normal> wn;
    if (N == 0 || !dev || dev->D == 0) return std::make_shared<dynfu::Frame>(0, wv, wn);
    std::vector<float> hv(3 * (size_t)N), hn(3 * (size_t)N), hdq(8 * nodes.size());
    for (int i = 0; i < N; ++i) {
        hv[3 * i] = verts[i].x, hv[3 * i + 1] = verts[i].y, hv[3 * i + 2] = verts[i].z;
        const dfa::Normal n = i < (int)normals.size() ? normals[i] : dfa::Normal();
        hn[3 * i] = n.normal_x, hn[3 * i + 1] = n.normal_y, hn[3 * i + 2] = n.normal_z;
    }
    for (size_t i = 0; i < nodes.size(); ++i) {
        const auto& dq = *nodes[i]->getTransformation();
        const auto r = dq.getReal(), d = dq.getDual();
        float* o = &hdq[8 * i];
        o[0] = r.a, o[1] = r.b, o[2] = r.c, o[3] = r.d, o[4] = d.a, o[5] = d.b, o[6] = d.c, o[7] = d.d;
    }
    dfa::DeviceArray<float> dv, dn, ov(3 * (size_t)N), on(3 * (size_t)N);
    dv.upload(hv), dn.upload(hn), dev->dq.upload(hdq);
    dfa::check(dfa_warp_to_live(dev->pos.ptr(), dev->dq.ptr(), dev->w.ptr(), dev->D, knn_, dv.ptr(), dn.ptr(), N,
                                ov.ptr(), on.ptr(), nullptr),
               "Warpfield::warpToLive");
    ov.download(hv), on.download(hn);
    for (int i = 0; i < N; ++i) {
        wv.push_back(dfa::PointXYZ(hv[3 * i], hv[3 * i + 1], hv[3 * i + 2]));
        wn.push_back(dfa::Normal(hn[3 * i], hn[3 * i + 1], hn[3 * i + 2]));
    }
    return std::make_shared<dynfu::Frame>(0, wv, wn);
}
