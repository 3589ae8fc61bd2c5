// northstar_solver.cpp — NorthStarSolver on the dfa_solver6 plan (see dynfu/utils/northstar_solver.hpp).
#include <cstdio>
#include <dynfu/utils/northstar_solver.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include <dfa_host/plan_cache.hpp>
#include <kfusion/cuda/imgproc.hpp>

#include "../../../include/dynfu_amd.h"

extern void dfa_host_copy_from_device(void*, const void*, size_t);

namespace {
// What the graphs of a plan were built from.  A DynFusion solves frame after frame over the SAME canonical cloud and —
// until the field grows — the same nodes: k-NN, transposition and pair lists of the frame before are then still right and
// only the starting transforms change (dfa_solver6_set_node_transforms).  The arrays are kept alive here: the plan
// borrows them.
struct Built {
    dfa::DeviceArray<float> node_pos, node_w, canon, canon_n;
    int D = 0, N = 0, k = 0;
    uint64_t nodes_hash = 0;
};
std::mutex g_built_mu;
std::map<dfa_solver6*, Built> g_built;

void destroy_plan6(dfa_solver6* p) {
    {
        std::lock_guard<std::mutex> lock(g_built_mu);
        g_built.erase(p);
    }
    dfa_solver6_destroy(p);
}
dfa::PlanCache<dfa_solver6, destroy_plan6>& plan_cache() {
    static dfa::PlanCache<dfa_solver6, destroy_plan6> c;
    return c;
}
uint64_t fnv1a(const std::vector<float>& a, uint64_t h) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(a.data());
    for (size_t i = 0; i < a.size() * sizeof(float); ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}
}  // namespace

struct NorthStarSolver::Impl {
    dfa_solver6* plan = nullptr;
    int plan_D = 0, plan_N = 0, plan_k = 0;
    dfa::DeviceArray<float> node_pos, node_dq, node_w, canon, canon_n;  // borrowed by the plan until the next set_problem
    kfusion::cuda::Cloud vmap;
    kfusion::cuda::Normals nmap;
    int D = 0, N = 0;
    ~Impl() { plan_cache().park(plan, plan_D, plan_N, plan_k); }
};

NorthStarSolver::NorthStarSolver(Warpfield warpfield, NorthStarParameters params, float tukeyOffset_, float psi_data_,
                                 float lambda_, float psi_reg_)
    : m_warpfield(std::move(warpfield)), m_params(params), tukeyOffset(tukeyOffset_), psi_data(psi_data_), lambda(lambda_),
      psi_reg(psi_reg_), impl(std::make_shared<Impl>()) {}
NorthStarSolver::~NorthStarSolver() = default;

void NorthStarSolver::initializeProblemInstance(const std::shared_ptr<dynfu::Frame> canonicalFrame) {
    const int D = (int)m_warpfield.nodesRef().size(), N = (int)canonicalFrame->size();
    if (D == 0) throw dfa::Error(DFA_ERR_INVALID, "NorthStarSolver: the warp field has no nodes");
    std::vector<float> pos, w, dq;
    m_warpfield.hostArrays(pos, w, dq);
    Impl& I = *impl;
    I.D = D, I.N = N;
    canonicalFrame->deviceArrays(I.canon, I.canon_n);
    const int k = std::min(m_warpfield.getKnn(), 8);  // a dfa_solver6 plan blends at most 8 nodes (the reference's KNN)
    if (I.plan && !(I.plan_k == k && I.plan_D >= D && I.plan_N >= N)) {
        plan_cache().park(I.plan, I.plan_D, I.plan_N, I.plan_k);
        I.plan = nullptr;
    }
    if (!I.plan) {
        I.plan   = plan_cache().take(D, N, k, I.plan_D, I.plan_N);
        I.plan_k = k;
    }
    if (!I.plan) {
        I.plan_D = D + D / 4 + 16, I.plan_N = N + N / 4 + 1024;
        dfa::check(dfa_solver6_create(I.plan_D, I.plan_N, k, &I.plan), "NorthStarSolver: dfa_solver6_create");
    }
    I.node_dq.upload(dq);
    const uint64_t nodes_hash = fnv1a(w, fnv1a(pos, 1469598103934665603ull));
    {
        std::lock_guard<std::mutex> lock(g_built_mu);
        auto it = g_built.find(I.plan);
        if (it != g_built.end() && !dfa::host_switch("DFA_HOST_NO_GRAPH_REUSE")) {
            const Built& b = it->second;
            if (b.D == D && b.N == N && b.k == k && b.nodes_hash == nodes_hash && b.canon.ptr() == I.canon.ptr() &&
                b.canon_n.ptr() == I.canon_n.ptr()) {
                I.node_pos = b.node_pos, I.node_w = b.node_w;  // shared: the plan reads them
                dfa::check(dfa_solver6_set_node_transforms(I.plan, I.node_dq.ptr()), "NorthStarSolver: set_node_transforms");
                return;
            }
        }
    }
    I.node_pos = dfa::DeviceArray<float>(), I.node_w = dfa::DeviceArray<float>();  // fresh arrays: older ones may be shared
    I.node_pos.upload(pos), I.node_w.upload(w);
    dfa::check(dfa_solver6_set_problem(I.plan, I.node_pos.ptr(), I.node_dq.ptr(), I.node_w.ptr(), D, I.canon.ptr(),
                                       I.canon_n.ptr(), N, nullptr),
               "NorthStarSolver::initializeProblemInstance");
    std::lock_guard<std::mutex> lock(g_built_mu);
    Built& b = g_built[I.plan];
    b.node_pos = I.node_pos, b.node_w = I.node_w, b.canon = I.canon, b.canon_n = I.canon_n;
    b.D = D, b.N = N, b.k = k, b.nodes_hash = nodes_hash;
}

void NorthStarSolver::solveAll(const kfusion::cuda::Depth& liveDepth, const kfusion::Intr& intr) {
    Impl& I = *impl;
    kfusion::cuda::computePointNormals(intr, liveDepth, I.vmap, I.nmap);  // imgproc.cu:187-226
    solveAll(I.vmap, I.nmap, intr);
}

void NorthStarSolver::solveAll(const kfusion::cuda::Cloud& vmap, const kfusion::cuda::Normals& nmap, const kfusion::Intr& intr) {
    Impl& I = *impl;
    if (!I.plan) throw dfa::Error(DFA_ERR_INVALID, "solveAll before initializeProblemInstance");
    if (vmap.rows() != nmap.rows() || vmap.cols() != nmap.cols() || vmap.empty())
        throw dfa::Error(DFA_ERR_INVALID, "NorthStarSolver: vertex / normal maps differ in size or are empty");
    dfa_solve6_params p;
    p.num_iter     = m_params.numIter;
    p.gn_iter      = m_params.gnIter;
    p.linear_iter  = m_params.linearIter;
    p.tukey_offset = tukeyOffset;
    p.psi_data     = psi_data;
    p.lambda       = lambda;
    p.psi_reg      = psi_reg;
    p.dist_thresh  = m_params.distThresh;
    p.cos_thresh   = m_params.cosThresh;
    p.damping      = m_params.damping;
    p.pcg_tol      = m_params.pcgTol;
    p.pcg_tol_first = m_params.pcgTolFirst;
    p.pcg_tol_decay = m_params.pcgTolDecay;
    p.pcg_tol_adapt = m_params.pcgTolAdapt;
    p.adaptive_launch = m_params.adaptiveLaunch ? 1 : 0;
    p.gn_tol          = m_params.gnTol;
    dfa::check(dfa_solver6_solve(I.plan, (const float*)vmap.ptr(), (int)vmap.step(), (const float*)nmap.ptr(), (int)nmap.step(),
                                 vmap.cols(), vmap.rows(), intr.fx, intr.fy, intr.cx, intr.cy, &p, nullptr),
               "NorthStarSolver::solveAll");
    dfa_solve6_stats st;
    dfa::check(dfa_solver6_get_stats(I.plan, &st, nullptr), "NorthStarSolver::solveAll (stats)");  // synchronises
    initial_cost_ = st.initial_cost, final_cost_ = st.final_cost, valid_rows_ = st.valid_last, pcg_iters_ = st.pcg_iters;
    pcg_short_ = st.pcg_short, gn_solves_ = st.gn_solves, gn_rejected_ = st.gn_rejected;
    if (st.pcg_short > 0) {  // a PCG stopped where its launch budget ended (still a descent step); the plan doubles that budget
        static bool told = false;
        if (!told)
            std::fprintf(stderr, "NorthStarSolver: %d PCG(s) of this solve were cut short by the adaptive launch budget "
                                 "(pcgsCutShort(); NorthStarParameters::adaptiveLaunch = false enqueues the full budget)\n",
                         st.pcg_short);
        told = true;
    }
    // the solved transforms replace dg_se3 of the shared Nodes (the reference's solver composes a translation onto it,
    // opt_solver.cpp:270-285; here the unknown IS the transform)
    std::vector<float> dq(8 * (size_t)I.D);
    dfa_host_copy_from_device(dq.data(), dfa_solver6_node_dq(I.plan), dq.size() * sizeof(float));
    const auto& nodes = m_warpfield.nodesRef();
    for (int i = 0; i < I.D; ++i) {
        const float* q = &dq[8 * (size_t)i];
        nodes[i]->setTransformation(std::make_shared<DualQuaternion<float>>(dfa::quaternion<float>(q[0], q[1], q[2], q[3]),
                                                                            dfa::quaternion<float>(q[4], q[5], q[6], q[7])));
    }
}

std::shared_ptr<dynfu::Frame> NorthStarSolver::warpCanonicalToLive() {
    Impl& I = *impl;
    if (!I.plan) throw dfa::Error(DFA_ERR_INVALID, "warpCanonicalToLive before initializeProblemInstance");
    dfa::DeviceArray<float> ov(3 * (size_t)I.N), on(3 * (size_t)I.N);
    if (I.N) dfa::check(dfa_solver6_warp(I.plan, ov.ptr(), on.ptr(), nullptr), "NorthStarSolver::warpCanonicalToLive");
    return dynfu::Frame::fromDevice(0, ov, on, (size_t)I.N);
}
