// opt_solver.cpp — CombinedSolver on the dynfu_amd solver plan
// (reference: src/dynfu/utils/opt_solver.cpp; the Opt GN/PCG it drives is replaced by
// dfa_solver_solve).
#include <dynfu/utils/opt_solver.hpp>

#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

#include <dfa_host/device.hpp>
#include <dfa_host/plan_cache.hpp>

#include "../../../include/dynfu_amd.h"

namespace {
dfa::PlanCache<dfa_solver, dfa_solver_destroy>& plan_cache() {
    static dfa::PlanCache<dfa_solver, dfa_solver_destroy> c;
    return c;
}
}  // namespace

struct CombinedSolver::Impl {
    dfa_solver* plan = nullptr;
    int plan_D = 0, plan_N = 0, plan_k = 0;  // capacity of `plan`
    dfa::DeviceArray<float> node_pos, node_dq, node_w, canon, live, canon_n, live_n;
    int D = 0, N = 0;
    ~Impl() { plan_cache().park(plan, plan_D, plan_N, plan_k); }
};

CombinedSolver::CombinedSolver(Warpfield warpfield, CombinedSolverParameters params, float tukeyOffset_,
                               float psi_data_, float lambda_, float psi_reg_)
    : m_warpfield(std::move(warpfield)), m_params(params), tukeyOffset(tukeyOffset_), psi_data(psi_data_), lambda(lambda_),
      psi_reg(psi_reg_), impl(std::make_shared<Impl>()) {}
CombinedSolver::~CombinedSolver() = default;

// opt_solver.cpp:15-54 (+ resetGPUMemory :149-202): stage AoS -> SoA, upload, build the graphs
void CombinedSolver::initializeProblemInstance(const std::shared_ptr<dynfu::Frame> canonicalFrame,
                                               const std::shared_ptr<dynfu::Frame> liveFrame, dfa::Affine3f /*affine*/) {
    const int D = (int)m_warpfield.nodesRef().size(), N = (int)canonicalFrame->size();
    if ((int)liveFrame->size() != N) throw dfa::Error(DFA_ERR_INVALID, "canonical / live vertex counts differ");
    std::vector<float> pos, w, dq;
    m_warpfield.hostArrays(pos, w, dq);
    Impl& I = *impl;
    I.D = D, I.N = N;
    I.node_pos.upload(pos), I.node_w.upload(w), I.node_dq.upload(dq);
    // the clouds: the frames' own packed device arrays, shared for the life of this solver (frames produced by the
    // adaptor's stages are already in HBM; frames built from host clouds upload here, as resetGPUMemory does)
    canonicalFrame->deviceArrays(I.canon, I.canon_n);
    liveFrame->deviceArrays(I.live, I.live_n);
    const int k = m_warpfield.getKnn();
    if (I.plan && !(I.plan_k == k && I.plan_D >= D && I.plan_N >= N)) {
        plan_cache().park(I.plan, I.plan_D, I.plan_N, I.plan_k);
        I.plan = nullptr;
    }
    if (!I.plan) {
        I.plan   = plan_cache().take(D, N, k, I.plan_D, I.plan_N);
        I.plan_k = k;
    }
    if (!I.plan) {  // headroom: the surface of the next frames has about as many vertices, rarely exactly as many
        I.plan_D = std::max(D, std::min(D + D / 4 + 16, 32768)), I.plan_N = N + N / 4 + 1024;  // (32768: a plan's node limit)
        dfa::check(dfa_solver_create(I.plan_D, I.plan_N, k, &I.plan), "CombinedSolver: dfa_solver_create");
    }
    dfa::check(dfa_solver_set_problem(I.plan, I.node_pos.ptr(), I.node_dq.ptr(), I.node_w.ptr(), D, I.canon.ptr(),
                                      I.canon_n.ptr(), I.live.ptr(), I.live_n.ptr(), N, nullptr),
               "CombinedSolver::initializeProblemInstance");
}

// CombinedSolverBase::solveAll with the hooks of opt_solver.cpp:107-147; results are written back
// to the shared Nodes ONCE (copyResultToCPUFromFloat3 :270-285 + Node::updateTransformation)
void CombinedSolver::solveAll() {
    Impl& I = *impl;
    if (!I.plan) throw dfa::Error(DFA_ERR_INVALID, "solveAll before initializeProblemInstance");
    dfa_solve_params p;
    // Opt's CombinedSolverBase::singleSolve (un-vendored; upstream examples/shared/CombinedSolverBase.h) leaves its
    // outer loop after the first pass when earlyOut is set — as DynFusion and every OptTest set it
    // (dyn_fusion.cpp:189, opt_optimisation_test.cpp:43): one robust re-weighting, then <= nonLinearIter Gauss-Newton
    // steps.  DFA_HOST_ALL_OUTER=1 runs all numIter passes (the behaviour of round 1) for comparison.
    const bool one_pass = m_params.earlyOut && !dfa::host_switch("DFA_HOST_ALL_OUTER");
    p.num_iter       = one_pass ? std::min(1, m_params.numIter) : m_params.numIter;
    p.nonlinear_iter = m_params.nonLinearIter;
    p.linear_iter    = m_params.linearIter;
    p.tukey_offset   = tukeyOffset;
    p.psi_data       = psi_data;
    p.lambda         = lambda;
    p.psi_reg        = psi_reg;
    p.pcg_tol        = 0.f;
    p.gn_tol         = 0.f;
    dfa::check(dfa_solver_solve(I.plan, &p, nullptr), "CombinedSolver::solveAll");
    dfa_solve_stats st;
    dfa::check(dfa_solver_get_stats(I.plan, &st, nullptr), "CombinedSolver::solveAll (stats)");
    initial_cost_ = st.initial_cost, final_cost_ = st.final_cost;
    std::vector<float> t(3 * (size_t)I.D);
    dfa::DeviceMemory tmp;  // borrowed view: copy out of the plan
    if (I.D) {
        dfa::device_synchronize();
        // plan-owned device array -> host
        extern void dfa_host_copy_from_device(void*, const void*, size_t);
        dfa_host_copy_from_device(t.data(), dfa_solver_translations(I.plan), t.size() * sizeof(float));
    }
    const auto& nodes = m_warpfield.nodesRef();  // (the Nodes are shared with the caller's warp field: opt_solver.cpp:5)
    for (int i = 0; i < I.D; ++i) {
        nodes[i]->updateTransformation(DualQuaternion<float>(0.f, 0.f, 0.f, t[3 * i], t[3 * i + 1], t[3 * i + 2]));
    }
}
