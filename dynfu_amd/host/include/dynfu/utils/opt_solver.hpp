// dynfu/utils/opt_solver.hpp — class CombinedSolver with the reference's interface
// (include/dynfu/utils/opt_solver.hpp:19-110, src/dynfu/utils/opt_solver.cpp) on the dynfu_amd
// solver plan: initializeProblemInstance == dfa_solver_set_problem, solveAll == dfa_solver_solve
// + the single write-back composition onto the (shared) Nodes.
#pragma once
#include <memory>

#include <dynfu/utils/frame.hpp>
#include <dynfu/warp_field.hpp>

// Opt's CombinedSolverParameters: the fields the reference sets (dyn_fusion.cpp:183-189,
// opt_optimisation_test.cpp:38-44)
struct CombinedSolverParameters {
    int numIter             = 1;
    int nonLinearIter       = 3;
    int linearIter          = 200;
    bool useOpt             = true;
    bool useOptLM           = false;
    bool earlyOut           = false;
    bool optDoublePrecision = false;  // the HIP solve is fp32 with double-accumulated costs
};

class CombinedSolver {
public:
    CombinedSolver(Warpfield warpfield, CombinedSolverParameters params, float tukeyOffset, float psi_data, float lambda,
                   float psi_reg);
    ~CombinedSolver();

    void initializeProblemInstance(const std::shared_ptr<dynfu::Frame> canonicalFrame,
                                   const std::shared_ptr<dynfu::Frame> liveFrame, dfa::Affine3f affine);
    void solveAll();

    // costs of the last solveAll (Opt: reportFinalCosts, opt_solver.cpp:144-147)
    double initialCost() const { return initial_cost_; }
    double finalCost() const { return final_cost_; }

private:
    Warpfield m_warpfield;  // copied by value, Nodes shared (opt_solver.cpp:5)
    CombinedSolverParameters m_params;
    float tukeyOffset, psi_data, lambda, psi_reg;
    struct Impl;
    std::shared_ptr<Impl> impl;
    double initial_cost_ = 0.0, final_cost_ = 0.0;
};
