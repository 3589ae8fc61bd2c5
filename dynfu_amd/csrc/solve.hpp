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
    int pcg_fallback;   // (unused since the streaming path runs inside the register-resident launch; keeps the layout)
    int split_iters;    // per-coordinate PCG: most iterations any coordinate took in the launch in flight
    unsigned int split_ticket;  // ... and how many of its three workgroups have finished
    // multi-workgroup PCG: flags and scalars carried from one launch to the next
    int mb_done;
    int converged;  // 1: sticky for the rest of the solve — the gradient of a linearisation with freshly evaluated robust
                    // weights was at the floor.  t can no longer change (re-weighting at the same t gives the same
                    // system), so every later Gauss-Newton iteration is a no-op and its kernels return at entry.
                    // 2: the same with STALE weights (an inner iteration of nonlinear_iter > 1): the energy is linear
                    // least squares while the weights are frozen, so the remaining inner iterations of this outer
                    // iteration are no-ops; the next re-weighting linearisation clears it.
                    // directly behind mb_done: the many-workgroup path reads both back with one copy
    int mb_skip, mb_iters;
    int weights_fresh;  // the last linearisation re-evaluated the robust weights (at the t it linearised about)
    int gn_noop;    // Gauss-Newton iterations that returned at entry this way (counted in gn_iters too)
    int cost_stale;  // t has moved since the last linearisation evaluated the cost (inner iterations restarted by
                     // solve_regradient): the closing evaluation must run even when the solve has converged
    float mb_rz0, mb_gamma_prev[2], mb_alpha_prev[2];
    float amax;     // largest addend tau w_a w_b of the normal matrix under the current robust weights (the re-weighting
                    // linearisation's last workgroup): the scale of the assembly's fixed-point sums
    long long prof[8];  // DFA_PCG_PROFILE builds: shader cycles per PCG phase (thread 0)
};

// All pointers are device pointers owned by the plan unless marked (borrowed).
// Words of a packed row record.  k a multiple of 8: the ids are 16-bit (a plan has at most 32 768 nodes), which makes the
// k = 8 record exactly one 64-byte cache line (80 bytes with 32-bit ids straddle two: every row is gathered by the
// workgroups of its k nodes).  Other k keep 32-bit ids (k = 4: 48 bytes; 16-bit ids would give 40, off the 16-byte grid).
__host__ __device__ inline bool solve_rec_ids16(int k) { return k % 8 == 0; }
__host__ __device__ inline int solve_rec_words(int k) { return solve_rec_ids16(k) ? k + k / 2 + 4 : 2 * k + 4; }
__host__ __device__ inline int solve_rec_tail(int k) { return solve_rec_words(k) - 4; }  // word offset of (e, tau)

// a PCG found its gradient at the round-off floor (see SolveState::converged)
__device__ __forceinline__ void solve_mark_at_floor(SolveState* st) {
    if (st->weights_fresh) st->converged = 1;
    else if (!st->converged) st->converged = 2;
}

// control block of the team PCG (pcg_team_kernel): device memory of the plan, zeroed once
struct TeamCtl {
    unsigned int count[3];                  // workgroups of team c that have arrived in the launch in flight (zeroed by the guard launch)
    unsigned int abort[3];                  // team c gave up in the launch in flight: the guard launch solves its coordinate
    unsigned int handled[3];                // team c has dealt with its coordinate in the launch in flight (the guard launch takes what nobody has)
};

struct SolveView {
    int N, D, k, Dpad, ell_cap;
    // Order-stable variant of the solve (dfa_solver_set_deterministic / DFA_ASSEMBLE_DETERMINISTIC=1): node lists sorted,
    // matrix assembled without cross-wave float atomics into rows sorted by column, rows of equal length in index order
    // in the PCG kernels — the same bits from the same inputs, for ~1.3x the assembly time.
    int deterministic;
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
    float* re;      // R x solve_rec_words(k) packed row records: k node ids, k weights, e = b - sum w t, tau
    int32_t* reg_idx;  // D x k
    // transpose graph
    int32_t* blk_hist;    // (TG_BLOCKS + 1) x D  workgroup-private histograms / bases of the counting sort, node totals
    int32_t* node_ptr;    // D + 1
    uint32_t* node_list;  // R x k   flat (row*k + slot) indices grouped by node
    // normal equations, ELL slot-major: entry q of row a at [q*D + a] as ONE 8-byte word (value, column as int bits) —
    // the register-resident PCG loads its rows in length-sorted order, i.e. scattered: one request per entry, not two
    float2* ell;
    int32_t* ell_cnt;  // D
    float* diag;       // D
    float* g;          // D x 3   -J^T r
    float* g_base;     // D x 3   g at the last full linearisation ...
    float* t_base;     // D x 3   ... and the t it was taken at (inner iterations: g = g_base - A (t - t_base))
    // workspace of the streaming PCG: rows sorted by length, rank-major repacked matrix
    int32_t* pk_perm;   // D
    int32_t* pk_perm2;  // D   (deterministic variant: the permutation before its equal-length runs are put in index order)
    float* pk_vals;     // ell_cap x D
    uint16_t* pk_cols;  // ell_cap x D
    // unknown and outputs
    // multi-workgroup PCG (more than 8192 nodes): vectors as float4 per node, ping-pong where other rows read them
    float4 *mb_x, *mb_r, *mb_p, *mb_s, *mb_w, *mb_u[2], *mb_m[2], *mb_t[2];
    float *mb_gpart[2], *mb_dpart[2];  // per-workgroup partial (r, u), (w, u)
    // team PCG: control block, and the (m, t) pairs the row owners publish: [barrier round of the launch][coordinate][team_stride]
    TeamCtl* team_ctl;
    float2* team_mt;
    unsigned long long* team_words;  // flag words {round, partial sum}: [coordinate][barrier round][copy][kind][member]
    int team_stride;
    float* t;            // D x 3
    float* huber;        // D
    float* node_dq_out;  // D x 8
};

constexpr int SOLVE_TG_BLOCKS = 256;  // one workgroup of the counting sort per CU (64 until round 5: a quarter of the chip)
// rows (regularisation rows, right-hand sides, packed record heads), reset of the unknowns / state / tickets, and the
// node -> rows transposition of the problem in `s`
hipError_t solve_build_graph(const SolveView& s, SolveState* state, unsigned int* ticket, int nticket, hipStream_t st);
// counting-sort transposition of an (rows x k) node-index array: blk_hist = (SOLVE_TG_BLOCKS + 1) x D scratch,
// node_ptr = D + 1, node_list = flat (row * k + slot) indices grouped by node (entries < 0 skipped)
hipError_t solve_transpose_graph(const int32_t* ridx, size_t total, int D, int32_t* blk_hist, int32_t* node_ptr,
                                 uint32_t* node_list, hipStream_t st);
int solve_residual_blocks(const SolveView& s);
// robust weights (optional) + residuals + cost + Gauss-Newton control, one launch
hipError_t solve_linearise(const SolveView& s, SolveState* state, double* cost_partials, unsigned int* ticket,
                           int update_weights, int mode, float gn_tol, float tukey_offset, float psi_data,
                           float w_reg_sq, float huber_psi /* > 0: the nodes' Huber weights too */,
                           long long* iters_total /* mode 2: += the solve's PCG iterations (optional, device) */, hipStream_t st);
hipError_t solve_huber(const SolveView& s, float psi_reg, hipStream_t st);
hipError_t solve_reset(const SolveView& s, SolveState* state, unsigned int* ticket, int nticket, hipStream_t st);
// w_reg_sq: the weight of the regularisation rows — with 1 (the data rows) the bound of an addend's magnitude, which sets the
// scale of the assembly's fixed-point sums
hipError_t solve_assemble(const SolveView& s, SolveState* state, int save_base /* g, t -> g_base, t_base */, float w_reg_sq,
                          hipStream_t st);
// inner Gauss-Newton iteration of an outer iteration whose robust weights are frozen: the energy is linear least squares,
// so the matrix of the outer iteration's first linearisation still holds and the new right-hand side is
// g = g_base - A (t - t_base) — one sparse matrix-vector product instead of a linearisation and an assembly
hipError_t solve_regradient(const SolveView& s, SolveState* state, hipStream_t st);
int solve_pcg_max_nodes();
__host__ __device__ inline int solve_mb_rows_per_block() { return 16; }  // multi-workgroup PCG: 16 lanes per row, 256 threads
__host__ __device__ inline int solve_mb_blocks(int D) { return (D + 15) / 16; }
// host_flag: pinned host word (may be null); with it, plans above 2048 nodes synchronise with the stream once per
// chunk of PCG launches to stop launching after convergence
// The launches of a chunk of many-workgroup PCG iterations, replayed as HIP graphs (one per range of iterations: the
// iteration number is a kernel argument).  Behind a stream synchronisation the host cannot enqueue two 5 us kernels
// per iteration fast enough — the GPU idled 4-24 us between launches (profile of C3) — a graph is one host call.
struct MbGraphCache {
    struct Entry {
        int it0 = 0, it1 = 0;
        float tol = 0.f;
        SolveView view;
        const SolveState* state = nullptr;
        hipGraphExec_t exec = nullptr;
    };
    Entry e[24];
    int used = 0;
    // iterations the PCG of Gauss-Newton iteration i took in the solve before (0 = unknown): the first chunk of launches
    // is sized by it — a sequence changes little from frame to frame — instead of growing 16, 32, 64, ... past the end
    int pred[64] = {};
    int call     = 0;  // index of the next PCG inside the current solve (reset by the caller)
    hipStream_t capture = nullptr;  // capture is not allowed on the legacy default stream
    bool disabled = false;
    void release();
};

// host side of the team PCG of one plan (the device side is in the view)
struct TeamPcg {
    TeamCtl* ctl    = nullptr;  // == SolveView::team_ctl (null: no team PCG for this plan)
    int* host_abort = nullptr;  // pinned: teams that have given up so far, written by the device
    unsigned epoch  = 1;        // first barrier round of the next launch (rounds only ever grow: no flag is ever reset)
    long launches   = 0;
    bool disabled   = false;    // a team has given up before: the launched form from now on
};
bool solve_team_pcg_fits(int D);
int solve_team_pcg_rounds();
size_t solve_team_pcg_words();  // 64-bit flag words a plan holds  // exchange areas a plan holds (barrier rounds of one launch)

bool solve_pcg_is_async(const SolveView& s, const TeamPcg* team, int max_iter);
// `main_done` (optional) is recorded behind the solving kernel(s), before the fallback launch that usually returns at once
hipError_t solve_pcg(const SolveView& s, SolveState* state, int max_iter, float pcg_tol, int* host_flag /* pinned int[4] or null */,
                     MbGraphCache* graphs /* or null */, TeamPcg* team /* or null */, hipEvent_t main_done,
                     hipStream_t st);
// books n Gauss-Newton iterations that the host did not launch because the plan had converged (SolveState::converged)
hipError_t solve_count_noop(SolveState* state, int n, hipStream_t st);

}  // namespace dfa
