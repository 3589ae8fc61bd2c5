// solve6.hpp — device-side view of a north-star (6-DoF) solver plan and the launchers of
// solve6.hip (internal).  Formulas: DESIGN.md §4.5; CPU statement: oracle/solve6_oracle.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfa {

struct Solve6Params {
    int num_iter, gn_iter, linear_iter;
    float tukey_offset, psi_data, lambda, psi_reg, dist_thresh, cos_thresh, damping, pcg_tol;
    // inexact-Newton forcing: Gauss-Newton iteration i of an outer iteration stops its PCG at
    // max(pcg_tol, pcg_tol_first * pcg_tol_decay^i); pcg_tol_first <= 0: constant pcg_tol
    float pcg_tol_first, pcg_tol_decay;
    // pcg_tol_adapt > 0: Eisenstat-Walker forcing (choice 2, alpha = 2) instead of the geometric schedule: iteration i > 0
    // of an outer iteration stops at clamp(pcg_tol_adapt (r.z)_0,i / (r.z)_0,i-1, pcg_tol, pcg_tol_first)
    float pcg_tol_adapt;
    // > 0: Gauss-Newton stopping rule + step acceptance (include/dynfu_amd.h: dfa_solve6_params.gn_tol), decided on the
    // device by the last workgroup of the linearisation; <= 0: every iteration runs (the launches and the bookkeeping are
    // those of the rounds before the rule existed)
    float gn_tol;
};

constexpr int S6_HIST = 32;  // = DFA_SOLVE6_HIST of include/dynfu_amd.h
constexpr int S6_LIN_SHARDS = 32;  // arrival counters of the linearisation (one word would serialise ~2 000 atomics at ~11 ns)
constexpr int S6_MIRROR_SKIPPED = -(1 << 30);  // launch-budget mirror: the Gauss-Newton iteration ran no PCG (its outer iteration had ended)

// Device-resident scalars of one solve
struct Solve6State {
    double cost;         // accumulator of the linearisation in flight
    double initial_cost;
    double final_cost;
    unsigned long long valid, valid_first, valid_last;
    int have_first;
    int gn_iters, pcg_iters;
    int overflow;        // a block row did not fit the plan's capacity / hash table
    int max_row_blocks;
    int pcg_done;        // sticky flag of the PCG in flight
    float rz0;           // (r, u) of its first iteration
    float gamma_prev[2], alpha_prev[2];  // scalars of the previous iteration (ping-pong)
    float tol2;          // stop test of the PCG in flight: (r, u) <= tol2 (r, u)_0   (set by the assembly launch; with the
                         // adaptive forcing term: by the PCG's first step, from the two gradients)
    float ew_gamma, ew_min2, ew_max2;  // adaptive forcing of the PCG in flight: factor (0: off), bounds of tol2
    int ew_slot;         // parity of the Gauss-Newton iteration inside its outer iteration
    float rz0_gn[2];     // (r, u)_0 of this and of the previous Gauss-Newton iteration (by parity)
    float pcg_tol_hist[S6_HIST];  // the relative residual every PCG was asked for
    int pcg_last_it;     // iterations the PCG in flight has completed
    int pcg_short;       // PCGs of this solve that used every launch enqueued for them — fewer than linear_iter, by the plan's
                         // prediction — without reaching their tolerance
    // per Gauss-Newton iteration (the first S6_HIST): energy at the linearisation, PCG iterations, relative residual
    // sqrt((r, u) / (r, u)_0) the PCG stopped at
    double cost_hist[S6_HIST];
    float pcg_rel_hist[S6_HIST];
    int pcg_it_hist[S6_HIST];
    // ---- Gauss-Newton control (gn_tol > 0; all zero / unused otherwise)
    int cur;             // history slot of the Gauss-Newton iteration in flight (gn_tol <= 0: gn_iters - 1)
    int hist_n;          // history slots written
    int gn_stop;         // 0: the outer iteration is running; 1: it has ended (converged); 2: it has ended by a rejected step —
                         // every s6_update until the next outer iteration puts the transforms before that step back
    int gn_solves, gn_rejected, gn_converged;
    double cost_ref;     // energy at the last accepted linearisation of the outer iteration
    unsigned long long valid_ref;
    unsigned int valid_hist[S6_HIST];
    int stop_hist[S6_HIST];  // 0 solved, 1 converged here, 2 rejected here, 3 skipped
    unsigned int lin_ticket[S6_LIN_SHARDS + 1];  // arrivals of the linearisation's workgroups (gn_tol > 0: the last one decides)
};

struct Solve6Image {  // live vertex / normal maps (borrowed): float4 pixels, NaN where undefined
    const float* vmap;
    const float* nmap;
    int vstep, nstep;  // bytes per row
    int cols, rows;
    float fx, fy, cx, cy;
};

struct Solve6View {
    int N, D, k, cap;  // cap = 6x6 blocks per block row (slot 0 = diagonal)
    const int32_t* xcd_perm;  // development builds, DFA_XCD_MAP=2: nodes along a Morton curve of their positions (else null)
    // problem (borrowed)
    const float* node_pos;  // D x 3
    const float* node_w;    // D
    const float* canon;     // N x 3          canonical vertices / normals in the solver's order (own copies, see below)
    const float* canon_n;   // N x 3 or null
    // The solver works on its own copy of the vertices, SORTED BY NEAREST NODE (s6_build_graph): lanes of a wave then
    // share their nodes' transforms and neighbouring pixels, the rows a node's workgroup gathers lie in a few runs
    // instead of all over the cloud.  vperm maps the solver's order back to the caller's (s6_warp scatters through it).
    int32_t* idx_nat;   // N x k  nearest nodes, caller's vertex order (k-NN output)
    int32_t* near;      // N      nearest node of every vertex
    int32_t* vptr;      // D + 1  first sorted position of every node's vertices
    uint32_t* vlist;    // N      vertices grouped by nearest node (scratch of the sort)
    uint32_t* vperm;    // N      sorted position -> caller's vertex
    float* canon_own;   // N x 3  (= canon below)
    float* canon_n_own; // N x 3  (= canon_n below, when the caller gave normals)
    // graphs (sorted order from here on)
    int32_t* idx;       // N x k  nearest nodes
    float* wn;          // N x k  normalised radial basis weights
    int32_t* reg_idx;   // D x k  nearest OTHER nodes
    int32_t* blk_hist;  // transposition scratch
    int32_t* node_ptr;  // D + 1
    uint32_t* node_list;   // (vertex * k + slot) grouped by node
    int32_t* rnode_ptr;    // D + 1
    uint32_t* rnode_list;  // (node * k + slot) of the regularisation edges ARRIVING at a node
    // state of the iteration
    float* dq;    // D x 8  current node transforms
    float* dq_prev;  // D x 8  the transforms before the last step (what a rejected step is undone with)
    float* ghat;  // D x 3  current node positions T_i(g_i)
    // linearisation
    // rows of the data term, one record per VERTEX: the row's 6-vector for neighbour j is f_j M_j l with the per-vertex
    // functional l = (lW, lD).  rec[v] = { l[8], f[K] (f_j = w~_j s_j / |a|^2), robust weight (0 = no association),
    // weight * residual, 0, 0 }, K = 4 or 8 (the kernels' template): 2 + K / 4 + 1 chunks of 16 bytes, the unit the
    // assembly's LDS-DMA fetches them in
    float* rec;    // N x (12 + K)
    double* cost_part;        // per workgroup of s6_linearise: its share of the energy (summed by the assembly's first workgroup)
    unsigned int* valid_part; // ... and of the valid rows
    float* mnode;  // D x 6 x 8  M_n: twist components of node n as (W, Wd) increments
    float* rho;   // N           Tukey weight (frozen between re-weightings)
    float* rres;  // D x k x 3   regularisation residuals
    float* rvec;  // D x k x 18  d e / d xi_n  (3 rows x 6)
    float* rhub;  // D x k       Huber weights (frozen)
    // normal equations, block ELL
    int32_t* bcols;  // D x cap
    int32_t* bcnt;   // D
    int32_t* bfu;    // D  first "upper" slot of the row (column > row; slots 1 .. bfu-1 are mirrored from their columns' rows)
    uint8_t* rslot;  // D x cap  slot of the row's node in the row of each of its columns
    // The 256 work units of a node's assembly workgroup, one per lane: unit u walks records phase, phase + n, ... of the pair
    // list of block `slot` (slot 0 or an upper slot) — longer lists get more units, so that every unit walks about the same
    // number of records.  utab[256 a + u] = slot | phase << 6 | n << 16; slot 63: idle.
    uint32_t* utab;  // D x 256
    uint8_t* eslot;  // (N k) x k scratch of s6_pattern (nodes whose slot bytes do not fit its LDS buffer)
    // the same relation by slot: for node a and slot q >= 1, pair_list[pair_ptr[a (cap+1) + q] .. pair_ptr[.. q+1]) are
    // the (row of a's list << 4 | neighbour) pairs that land in slot q, ascending; slot 0 = every row with its own neighbour
    uint32_t* pair_list;  // (N k) x k
    int32_t* pair_ptr;    // D x (cap + 1)
    float* bvals;    // D x cap x 36
    float* minv;     // D x 36  inverse of the (damped) diagonal block
    float* g;        // D x 6   -J^T W r
    // PCG
    float *x, *r, *p, *s, *w;   // D x 6, touched by the owning wave only
    float *u[2], *t[2], *m[2];  // D x 6, ping-pong: read by every wave while the owner writes the other copy
    float *g_part[2], *d_part[2];  // per matvec workgroup: partial (r, u) and (w, u)
};

constexpr int S6_NODES_PER_BLOCK = 8;  // matvec: one wave per node, 512 threads
__host__ __device__ inline int s6_matvec_blocks(int D) { return (D + S6_NODES_PER_BLOCK - 1) / S6_NODES_PER_BLOCK; }
__host__ __device__ inline int s6_update_blocks(int D) { return (6 * D + 255) / 256; }

// canon_user / canon_n_user: the caller's vertices (and normals or null) in the caller's order; raw_w and s.idx_nat hold
// the k-NN pass's output for them
hipError_t s6_build_graph(const Solve6View& s, Solve6State* state, const float* canon_user, const float* canon_n_user,
                          const float* raw_w /* N x k */, const int32_t* raw_reg /* D x (k+1) */, int kreg, hipStream_t st);
// slots: history slots the solve will enqueue when gn_tol > 0 (they start as "skipped"), 0 otherwise
hipError_t s6_begin(const Solve6View& s, Solve6State* state, const float* node_dq, int slots, hipStream_t st);
// gn_tol > 0: the launch also sums the energy and applies the Gauss-Newton stopping rule for history slot gi (gn_in_outer:
// index inside the outer iteration; closing: the check of the solve's last step) — by its last workgroup, which every other
// workgroup's partial sums reach through write-through stores
hipError_t s6_linearise(const Solve6View& s, Solve6State* state, const Solve6Image& img, const Solve6Params& p,
                        int update_weights, int gi, int gn_in_outer, int closing, hipStream_t st);
// gn_in_outer: index of the Gauss-Newton iteration inside its outer iteration (selects the PCG tolerance of the forcing schedule)
hipError_t s6_assemble(const Solve6View& s, Solve6State* state, const Solve6Params& p, int gn_in_outer, hipStream_t st);
struct S6Forcing {  // what the assembly launch (gn_tol > 0: the linearisation) leaves in the state block for the PCG that follows
    float tol2, ew_gamma, ew_min2, ew_max2;
    int ew_slot;
    int decided;  // the linearisation has done the bookkeeping itself (gn_tol > 0): the assembly launch only assembles
};
S6Forcing s6_forcing(const Solve6Params& p, int gn_in_outer);
// gn_tol > 0: what the linearisation's last workgroup needs to decide — the history slot gi = outer * gn_iter + gn_in_outer,
// `closing`: the check of the solve's last step (no iteration follows), the PCG parameters of the iteration
struct S6Decide {
    int on, gi, gn, closing;
    float gn_tol;
    S6Forcing f;
};
hipError_t s6_pcg(const Solve6View& s, Solve6State* state, const Solve6Params& p, hipStream_t st);
// launched: step launches enqueued for this PCG (<= linear_iter); mirror: pinned host int[S6_HIST] or null — iterations of
// every PCG of the solve as the device finishes them, negative when the PCG used every launch without converging
// gi: history slot (the mirror's too); apply = 0: the closing check's launch — a rejected step is undone, nothing is applied
hipError_t s6_update(const Solve6View& s, Solve6State* state, int launched, int linear_iter, int* mirror, int gi, int apply,
                     hipStream_t st);
hipError_t s6_pcg_n(const Solve6View& s, Solve6State* state, int launches, hipStream_t st);
hipError_t s6_warp(const Solve6View& s, const float* dq, float* out_v, float* out_n, hipStream_t st);
hipError_t launch_points_normals(const uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                                 float cy, float* points, int points_step, float* normals, int normals_step,
                                 hipStream_t st);

}  // namespace dfa
