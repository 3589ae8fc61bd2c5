// dynfu/utils/northstar_solver.hpp — the 6-DoF solve of BASELINE.json's north star behind an interface shaped like
// CombinedSolver's (include/dynfu/utils/opt_solver.hpp:19-110).  NOT in the reference: its energy.t moves node
// TRANSLATIONS towards matched live VERTICES; this solver moves full node transforms (a twist per node) towards the live
// DEPTH FRAME — dual-quaternion blend of the k nearest nodes, projective association into the live vertex / normal maps,
// point-to-plane data term (Tukey), ARAP regulariser (Huber), Gauss-Newton with a block-Jacobi PCG (DESIGN.md §4.5,
// dfa_solver6_* in include/dynfu_amd.h).  Same life cycle as CombinedSolver:
//     NorthStarSolver solver(*warpfield, params, tukeyOffset, psi_data, lambda, psi_reg);
//     solver.initializeProblemInstance(canonicalFrame);      // graphs of the frame (canonical cloud WITH normals)
//     solver.solveAll(liveDepth, intr);                      // vertex / normal maps of the depth frame, solve, write-back
// and the shared Nodes come out with their dg_se3 REPLACED by the solved transforms.  The canonical cloud, the nodes and
// the depth frame must be in ONE frame: the camera's.
#pragma once
#include <memory>

#include <dynfu/utils/frame.hpp>
#include <dynfu/warp_field.hpp>
#include <kfusion/types.hpp>

struct NorthStarParameters {
    int numIter      = 2;   // outer iterations: robust weights re-evaluated
    int gnIter       = 3;   // Gauss-Newton iterations per outer iteration
    int linearIter   = 64;  // PCG iterations per Gauss-Newton iteration, at most
    float distThresh = 0.1f;   // association gate |p - l| (metres)         (KinFuParams::icp_dist_thres)
    float cosThresh  = 0.5f;   // association gate n_warped . n_live
    float damping    = 1e-4f;  // added to the diagonal of the normal matrix
    // PCG stop test: relative residual max(pcgTol, pcgTolFirst * pcgTolDecay^i) at Gauss-Newton iteration i of an outer
    // iteration (inexact Newton: early linearisations are solved loosely); pcgTolFirst <= 0: constant pcgTol
    float pcgTol      = 1e-3f;
    float pcgTolFirst = 0.1f;
    float pcgTolDecay = 0.5f;
    // > 0: the Eisenstat-Walker forcing term instead of the geometric schedule (dfa_solve6_params.pcg_tol_adapt)
    float pcgTolAdapt = 0.9f;
    // enqueue only as many PCG launches per Gauss-Newton iteration as the previous frames needed (+ a quarter): see
    // dfa_solve6_params.adaptive_launch in dynfu_amd.h
    bool adaptiveLaunch = true;
    // Gauss-Newton stopping rule + step acceptance (dfa_solve6_params.gn_tol): gnIter is a cap, as nonLinearIter is for the
    // reference, which runs Opt with earlyOut = true (src/dynfu/dyn_fusion.cpp:183-189).  0: every iteration runs.
    float gnTol = 1e-3f;
};

class NorthStarSolver {
public:
    NorthStarSolver(Warpfield warpfield, NorthStarParameters params, float tukeyOffset, float psi_data, float lambda,
                    float psi_reg);
    ~NorthStarSolver();

    void initializeProblemInstance(const std::shared_ptr<dynfu::Frame> canonicalFrame);
    void solveAll(const kfusion::cuda::Depth& liveDepth, const kfusion::Intr& intr);
    // ... or against maps the caller already has (kfusion::cuda::computePointNormals, TsdfVolume::raycast)
    void solveAll(const kfusion::cuda::Cloud& liveVertexMap, const kfusion::cuda::Normals& liveNormalMap,
                  const kfusion::Intr& intr);

    // the canonical frame of the problem warped by the solved transforms (dual-quaternion blend, as the solve models
    // it); stays in HBM
    std::shared_ptr<dynfu::Frame> warpCanonicalToLive();

    double initialCost() const { return initial_cost_; }
    double finalCost() const { return final_cost_; }
    long long validRows() const { return valid_rows_; }  // data rows with an association at the last linearisation
    int pcgIterations() const { return pcg_iters_; }
    /* PCGs of the last solveAll that stopped at the end of their adaptive launch budget instead of at their tolerance */
    int pcgsCutShort() const { return pcg_short_; }
    /* normal equations solved / steps undone by the acceptance test in the last solveAll (gnTol > 0) */
    int gnSolves() const { return gn_solves_; }
    int gnRejected() const { return gn_rejected_; }

private:
    Warpfield m_warpfield;  // copied by value, Nodes shared (as CombinedSolver, opt_solver.cpp:5)
    NorthStarParameters m_params;
    float tukeyOffset, psi_data, lambda, psi_reg;
    struct Impl;
    std::shared_ptr<Impl> impl;
    double initial_cost_ = 0.0, final_cost_ = 0.0;
    long long valid_rows_ = 0;
    int pcg_iters_        = 0;
    int pcg_short_        = 0;
    int gn_solves_ = 0, gn_rejected_ = 0;
};
