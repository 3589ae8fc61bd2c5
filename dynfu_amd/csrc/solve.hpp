// solve.hpp — device-side view of a solver plan and the launchers of solve.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfa {

// Device-resident scalars of one solve (read back by dfa_solver_get_stats).
struct SolveState {
    double initial_cost;
    double cost;        // cost at the last linearisation
    double final_cost;
    double grad_first;  // g.M^-1.g of the first linearisation: scale of the convergence floor
    int have_initial;
    int done;           // Gauss-Newton early-out flag of the current outer iteration
    int gn_iters;
    int pcg_iters;
    int max_row_nnz;
    int overflow;       // a row of the normal matrix did not fit the plan's ELL capacity
    int pcg_fallback;   // set by the register-resident PCG when a row pair exceeds its slots
    long long prof[8];  // DFA_PCG_PROFILE builds: shader cycles per PCG phase (thread 0)
};

// All pointers are device pointers owned by the plan unless marked (borrowed).
struct SolveView {
    int N, D, k, Dpad, ell_cap;
    // problem (borrowed from the caller)
    const float* node_pos;  // D x 3
    const float* node_dq;   // D x 8
    const float* node_w;    // D
    const float* canon;     // N x 3
    const float* live;      // N x 3
    // residual rows: R = N + D*k rows of k slots
    int32_t* ridx;  // R x k   node per slot, -1 = empty
    float* rw;      // R x k   slot weight
    float* rtau;    // R       robust weight (Tukey) / w_reg^2
    float* rb;      // R x 3   target  (live - canonical | 0)
    float* re;      // R x (2k+4) packed row records: k node ids, k weights, e = b - sum w t, tau
    int32_t* reg_idx;  // D x k
    // transpose graph
    int32_t* blk_hist;    // TG_BLOCKS x D  workgroup-private histograms / bases of the counting sort
    int32_t* node_ptr;    // D + 1
    uint32_t* node_list;  // R x k   flat (row*k + slot) indices grouped by node
    // normal equations, ELL slot-major: entry q of row a at [q*D + a]
    int32_t* ell_cols;
    float* ell_vals;
    int32_t* ell_cnt;  // D
    float* diag;       // D
    float* g;          // D x 3   -J^T r
    // workspace of the streaming PCG: rows sorted by length, rank-major repacked matrix
    int32_t* pk_perm;   // D
    float* pk_vals;     // ell_cap x D
    uint16_t* pk_cols;  // ell_cap x D
    // unknown and outputs
    float* t;            // D x 3
    float* huber;        // D
    float* node_dq_out;  // D x 8
};

constexpr int SOLVE_TG_BLOCKS = 64;
hipError_t solve_build_graph(const SolveView& s, hipStream_t st);
// counting-sort transposition of an (rows x k) node-index array: blk_hist = SOLVE_TG_BLOCKS x D scratch,
// node_ptr = D + 1, node_list = flat (row * k + slot) indices grouped by node (entries < 0 skipped)
hipError_t solve_transpose_graph(const int32_t* ridx, size_t total, int D, int32_t* blk_hist, int32_t* node_ptr,
                                 uint32_t* node_list, hipStream_t st);
int solve_residual_blocks(const SolveView& s);
// robust weights (optional) + residuals + cost + Gauss-Newton control, one launch
hipError_t solve_linearise(const SolveView& s, SolveState* state, double* cost_partials, unsigned int* ticket,
                           int update_weights, int mode, float gn_tol, float tukey_offset, float psi_data,
                           float w_reg_sq, hipStream_t st);
hipError_t solve_huber(const SolveView& s, float psi_reg, hipStream_t st);
hipError_t solve_reset(const SolveView& s, SolveState* state, unsigned int* ticket, int nticket, hipStream_t st);
hipError_t solve_assemble(const SolveView& s, SolveState* state, hipStream_t st);
int solve_pcg_max_nodes();
hipError_t solve_pcg(const SolveView& s, SolveState* state, int max_iter, float pcg_tol, hipStream_t st);
hipError_t solve_writeback(const SolveView& s, hipStream_t st);

}  // namespace dfa
