// capi.cpp — the C ABI declared in include/dynfu_amd.h: argument checking, error strings,
// the solver plan's device memory, and the per-frame launch sequences.  No arithmetic lives
// here; the kernels are in tsdf.hip / warp.hip / solve.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "../../include/dynfu_amd.h"
#include "kernels.hpp"
#include "dev_switch.hpp"
#include "solve.hpp"
#include "solve6.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(e == hipErrorNoDevice ? DFA_ERR_NO_GPU : DFA_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define REQUIRE(cond, msg) \
    if (!(cond)) return fail(DFA_ERR_INVALID, "%s: %s", __func__, msg)
#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

inline hipStream_t S(dfa_stream_t s) { return (hipStream_t)s; }

bool volume_args_ok(const void* vol, int X, int Y, int Z) { return vol && X > 0 && Y > 0 && Z > 0; }

// Scratch of the entry points that have no plan to keep it in (dfa_knn, dfa_warp_to_live, dfa_correspond,
// dfa_marching_cubes, dfa_icp_sums ...): one instance per (device, stream), created on first use and kept.  Work on one
// stream is ordered, so a call never overwrites the scratch of a call still running — whichever host threads and
// however many streams the caller uses (round 1 kept these per host THREAD: two streams driven by one thread shared
// them).  Growing frees the old block with hipFree, which waits for the device.
template <class T>
T& stream_scratch(hipStream_t s) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, T> table;  // (device, stream): the null stream exists on every device
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    return table[std::make_pair(dev, s)];  // std::map nodes never move
}

// device scratch of one node grid (warp.hip); grows on demand, never shrinks
struct GridScratch {
    dfa::KnnGridView v{};
    int cap_nodes = 0;
    void release() {
        (void)hipFree(v.desc), (void)hipFree(v.cell_count), (void)hipFree(v.cell_start);
        (void)hipFree(v.node_cell), (void)hipFree(v.sorted);
        v         = dfa::KnnGridView{};
        cap_nodes = 0;
    }
    hipError_t reserve(int D_needed) {
        if (D_needed <= cap_nodes) return hipSuccess;
        // a warp field grows by a few nodes per frame: a quarter of headroom, so that growing (hipFree waits for the
        // device, and a fresh hipMalloc costs milliseconds) happens a handful of times in a sequence, not every frame
        const int D = D_needed + D_needed / 4 + 64;
        release();
        hipError_t e;
        if ((e = hipMalloc((void**)&v.desc, sizeof(dfa::KnnGridDesc))) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&v.cell_count, sizeof(int32_t) * dfa::KNN_GRID_MAX_CELLS)) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&v.cell_start, sizeof(int32_t) * (dfa::KNN_GRID_MAX_CELLS + 1))) != hipSuccess)
            return e;
        if ((e = hipMalloc((void**)&v.node_cell, sizeof(int32_t) * (size_t)D)) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&v.sorted, sizeof(float4) * (size_t)D)) != hipSuccess) return e;
        cap_nodes = D;
        return hipSuccess;
    }
};

// scratch of the 128^3 point grid of dfa_correspond; grows on demand, never shrinks
struct PointGridScratch {
    dfa::PointGridView v{};
    int cap_points = 0;
    void release() {
        (void)hipFree(v.g.desc), (void)hipFree(v.g.cell_count), (void)hipFree(v.g.cell_start);
        (void)hipFree(v.g.node_cell), (void)hipFree(v.g.sorted), (void)hipFree(v.chunk_sums);
        (void)hipFree(v.bbox_partials);
        v          = dfa::PointGridView{};
        cap_points = 0;
    }
    hipError_t reserve(int n_needed) {
        if (n_needed <= cap_points) return hipSuccess;
        const int n = n_needed + n_needed / 4 + 1024;  // clouds of consecutive frames differ by a few per cent (see GridScratch)
        release();
        hipError_t e;
        const size_t cells = dfa::PGRID_MAX_CELLS;
        if ((e = hipMalloc((void**)&v.g.desc, sizeof(dfa::KnnGridDesc))) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&v.g.cell_count, sizeof(int32_t) * cells)) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&v.g.cell_start, sizeof(int32_t) * (cells + 1))) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&v.g.node_cell, sizeof(int32_t) * (size_t)n)) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&v.g.sorted, sizeof(float4) * (size_t)n)) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&v.chunk_sums, sizeof(int32_t) * (cells / dfa::PGRID_CHUNK))) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&v.bbox_partials, sizeof(float) * 6 * dfa::PGRID_BBOX_BLOCKS)) != hipSuccess) return e;
        cap_points = n;
        return hipSuccess;
    }
};

// scratch of dfa_marching_cubes (segment offsets + scan partials); grows on demand
struct McScratch {
    int32_t* seg_off    = nullptr;
    int32_t* chunk_sums = nullptr;
    long cap_segs       = 0;
    hipError_t reserve(long nsegs) {
        if (nsegs <= cap_segs) return hipSuccess;
        (void)hipFree(seg_off), (void)hipFree(chunk_sums);
        seg_off = chunk_sums = nullptr, cap_segs = 0;
        hipError_t e;
        if ((e = hipMalloc((void**)&seg_off, sizeof(int32_t) * (size_t)(nsegs + 1))) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&chunk_sums, sizeof(int32_t) * (size_t)dfa::mc_scan_chunks(nsegs))) != hipSuccess)
            return e;
        cap_segs = nsegs;
        return hipSuccess;
    }
};


// exhaustive scan below this many distance evaluations (grid build = 4 small launches)
// the uniform-grid k-NN (three small launches to build, ~40 us) against the exhaustive scan: worth it for many queries, and for
// ANY number of queries over a large node set — a dozen new nodes against 8.5 k existing ones is one workgroup scanning them
// all, 0.86 ms in the adaptor's 512^3 sequence
bool want_grid(int D, long n_query) { return D >= 64 && ((long)D * n_query >= (1L << 22) || D >= 1024); }

}  // namespace

struct dfa_solver {
    int max_D, max_N, k;
    size_t max_R;
    int ell_cap;
    dfa::SolveView v;          // pointers into `blocks`
    dfa::SolveState* state;    // device
    double* cost_partials;     // device
    unsigned int* ticket;      // device: arrival counter of the linearise kernel (self re-arming)
    std::vector<void*> blocks;  // every hipMalloc of this plan
    GridScratch grid;           // node grid of the current problem
    bool has_problem;
    bool timing;
    std::vector<hipEvent_t> events;  // pool of timing events: they accumulate from enable_timing(1) on
    std::vector<int> ev_pcg, ev_asm; // indices of the begin events of each bracketed launch
    size_t ev_used;
    int timed_solves = 0;
    dfa::MbGraphCache mb_graphs;  // HIP graphs of the many-workgroup PCG's launch chunks
    dfa::TeamPcg team;            // host side of the team PCG (plans of 2 049 .. ~9 300 nodes)
    bool deterministic = false;  // order-stable variant (dfa_solver_set_deterministic), applied by the next set_problem
    bool just_reset = false;  // the unknowns and the state block were zeroed by set_problem and not touched since
    long long* iters_total = nullptr;  // device: PCG iterations of all solves since enable_timing(1)
    dfa_overlap_fn overlap_fn = nullptr;  // called behind the first assembly launch of every solve
    void* overlap_user        = nullptr;
    int* host_flag = nullptr;        // pinned int[4]: stop flag of the many-workgroup PCG, the plan's converged flag, (skip), iterations;
                                     // read back between launch chunks
};

struct dfa_solver6 {
    int max_D, max_N, k;
    dfa::Solve6View v;
    dfa::Solve6State* state;
    float* raw_w;       // N x k un-normalised weights of the k-NN pass
    int32_t* raw_reg;   // D x (k + 1)
    const float* node_dq;  // borrowed: transforms at set_problem time
    int32_t* xcd_perm = nullptr;  // (development builds, DFA_XCD_MAP=2: Morton order of the nodes — an experiment's scratch)
    std::vector<void*> blocks;
    GridScratch grid;
    bool has_problem;
    // the PCG of one Gauss-Newton iteration (linear_iter + 2 dependent launches) captured as a HIP graph:
    // re-captured when the problem size or the iteration parameters change
    std::map<int, hipGraphExec_t> pcg_graphs;  // by the number of step launches
    int last_launches = 0;  // step launches enqueued by the last solve
    // Adaptive launch budget (dfa_solve6_params.adaptive_launch).  The device writes the PCG iterations of every
    // Gauss-Newton iteration of solve n into slot n % S6_RING of a pinned mirror; the budget of solve n is a function of
    // the solves up to n - 2 ONLY, folded into the history in order behind their completion events — never of how far
    // the device happens to have got: the same sequence of solves gets the same budgets in every run.
    static constexpr int S6_RING = 4;
    int* mirror = nullptr;                        // pinned int[S6_RING][S6_HIST]
    hipEvent_t done_ev[S6_RING] = {};             // end of solve n, n % S6_RING
    int slot_gn[S6_RING] = {};                    // Gauss-Newton iterations solve n enqueued
    unsigned long long solve_seq = 0, folded = 0;  // solves started; solves whose counts are in the history
    int pred[dfa::S6_HIST] = {};                  // iterations per Gauss-Newton iteration: raised at once, lowered by one per solve
    struct BudgetKey { int D, N, num_iter, gn_iter, linear_iter; float tol, tol_first, tol_decay, tol_adapt, gn_tol; } budget_key = {};
    bool graph_disabled = false;
    hipStream_t capture_stream = nullptr;  // capture is not allowed on the legacy default stream
    bool timing = false;  // hipEvent brackets around linearise / assemble / PCG of every Gauss-Newton iteration
    std::vector<hipEvent_t> events;
    size_t ev_used = 0;
    int pcg_key_D = -1;  // node count the captured PCG graphs were recorded for
};

namespace {
template <class T>
int plan6_alloc(dfa_solver6* s, T** out, size_t count) {
    void* p      = nullptr;
    hipError_t e = hipMalloc(&p, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (solver6 plan)");
    s->blocks.push_back(p);
    *out = (T*)p;
    return DFA_OK;
}
}  // namespace

namespace {
// returns the index of a fresh event pair's begin event, recorded on st
int timing_begin(dfa_solver* s, hipStream_t st) {
    if (s->ev_used + 2 > s->events.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            s->events.push_back(e);
        }
    }
    const int idx = (int)s->ev_used;
    s->ev_used += 2;
    (void)hipEventRecord(s->events[idx], st);
    return idx;
}
void timing_end(dfa_solver* s, int idx, hipStream_t st) {
    if (idx >= 0) (void)hipEventRecord(s->events[idx + 1], st);
}

template <class T>
int plan_alloc(dfa_solver* s, T** out, size_t count) {
    void* p       = nullptr;
    hipError_t e  = hipMalloc(&p, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (solver plan)");
    s->blocks.push_back(p);
    *out = (T*)p;
    return DFA_OK;
}
}  // namespace

extern "C" {

const char* dfa_last_error(void) { return g_err; }

const char* dfa_version(void) { return "dynfu_amd 0.1 (gfx950, HIP)"; }

int dfa_abi_version(void) { return DFA_ABI_VERSION; }

size_t dfa_abi_struct_size(int id) {
    switch (id) {
        case DFA_STRUCT_SOLVE_PARAMS: return sizeof(dfa_solve_params);
        case DFA_STRUCT_SOLVE_STATS: return sizeof(dfa_solve_stats);
        case DFA_STRUCT_SOLVE_TIMING: return sizeof(dfa_solve_timing);
        case DFA_STRUCT_SOLVE6_PARAMS: return sizeof(dfa_solve6_params);
        case DFA_STRUCT_SOLVE6_STATS: return sizeof(dfa_solve6_stats);
        case DFA_STRUCT_SOLVE6_TIMING: return sizeof(dfa_solve6_timing);
        default: return 0;
    }
}

// ---------------------------------------------------------------------------------- TSDF seam

int dfa_compute_dists(const uint16_t* depth, int depth_step, uint16_t* dists, int dists_step, int cols, int rows,
                      float fx, float fy, float cx, float cy, dfa_stream_t stream) {
    REQUIRE(depth && dists, "null image");
    REQUIRE(cols > 0 && rows > 0, "empty image");
    REQUIRE(depth_step >= cols * 2 && dists_step >= cols * 2, "row step smaller than a row");
    REQUIRE(fx != 0.f && fy != 0.f, "zero focal length");
    HIP_TRY(dfa::launch_compute_dists(depth, depth_step, dists, dists_step, cols, rows, fx, fy, cx, cy, S(stream)));
    return DFA_OK;
}

int dfa_tsdf_clear(uint32_t* volume, int X, int Y, int Z, dfa_stream_t stream) {
    REQUIRE(volume_args_ok(volume, X, Y, Z), "bad volume");
    HIP_TRY(dfa::launch_tsdf_clear(volume, X, Y, Z, S(stream)));
    return DFA_OK;
}

size_t dfa_tsdf_occupancy_bytes(int X, int Y, int Z) {
    return X > 0 && Y > 0 && Z > 0 ? dfa::occ_dims(X, Y, Z).bytes() : 0;
}

int dfa_tsdf_clear_occ(uint32_t* volume, int X, int Y, int Z, uint8_t* occupancy, dfa_stream_t stream) {
    REQUIRE(volume_args_ok(volume, X, Y, Z), "bad volume");
    REQUIRE(occupancy, "null occupancy map");
    HIP_TRY(dfa::launch_tsdf_clear(volume, X, Y, Z, S(stream)));
    HIP_TRY(hipMemsetAsync(occupancy, 0, dfa::occ_dims(X, Y, Z).bytes(), S(stream)));
    return DFA_OK;
}

static int integrate_common(bool fused, const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume,
                            int X, int Y, int Z, const float voxel_size[3], float trunc_dist, int max_weight,
                            const float vol2cam[12], float fx, float fy, float cx, float cy, uint8_t* occupancy,
                            dfa_stream_t stream, bool occupancy_known = false) {
    REQUIRE(volume_args_ok(volume, X, Y, Z), "bad volume");
    REQUIRE(dists && cols > 0 && rows > 0 && dists_step >= cols * 2, "bad dists image");
    REQUIRE(voxel_size && vol2cam, "null parameter block");
    REQUIRE(trunc_dist > 0.f, "trunc_dist must be positive");
    REQUIRE(max_weight >= 0 && max_weight <= 65535, "max_weight must fit the 16-bit weight");
    HIP_TRY(dfa::launch_tsdf_integrate(fused, dists, dists_step, cols, rows, volume, X, Y, Z, voxel_size, trunc_dist,
                                       max_weight, vol2cam, fx, fy, cx, cy, occupancy, occupancy_known, S(stream)));
    return DFA_OK;
}

int dfa_tsdf_integrate_occ(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X, int Y, int Z,
                           const float voxel_size[3], float trunc_dist, int max_weight, const float vol2cam[12], float fx,
                           float fy, float cx, float cy, uint8_t* occupancy, dfa_stream_t stream) {
    REQUIRE(occupancy, "null occupancy map");
    return integrate_common(false, dists, dists_step, cols, rows, volume, X, Y, Z, voxel_size, trunc_dist, max_weight, vol2cam,
                            fx, fy, cx, cy, occupancy, stream);
}

int dfa_tsdf_clear_integrate_occ(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X, int Y,
                                 int Z, const float voxel_size[3], float trunc_dist, int max_weight,
                                 const float vol2cam[12], float fx, float fy, float cx, float cy, uint8_t* occupancy,
                                 dfa_stream_t stream) {
    REQUIRE(occupancy, "null occupancy map");
    return integrate_common(true, dists, dists_step, cols, rows, volume, X, Y, Z, voxel_size, trunc_dist, max_weight, vol2cam,
                            fx, fy, cx, cy, occupancy, stream);
}

int dfa_tsdf_clear_integrate_known_occ(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X,
                                       int Y, int Z, const float voxel_size[3], float trunc_dist, int max_weight,
                                       const float vol2cam[12], float fx, float fy, float cx, float cy, uint8_t* occupancy,
                                       dfa_stream_t stream) {
    REQUIRE(occupancy, "null occupancy map");
    return integrate_common(true, dists, dists_step, cols, rows, volume, X, Y, Z, voxel_size, trunc_dist, max_weight, vol2cam,
                            fx, fy, cx, cy, occupancy, stream, true);
}

int dfa_tsdf_integrate(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X, int Y, int Z,
                       const float voxel_size[3], float trunc_dist, int max_weight, const float vol2cam[12], float fx,
                       float fy, float cx, float cy, dfa_stream_t stream) {
    return integrate_common(false, dists, dists_step, cols, rows, volume, X, Y, Z, voxel_size, trunc_dist, max_weight,
                            vol2cam, fx, fy, cx, cy, nullptr, stream);
}

int dfa_tsdf_clear_integrate(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X, int Y,
                             int Z, const float voxel_size[3], float trunc_dist, int max_weight,
                             const float vol2cam[12], float fx, float fy, float cx, float cy, dfa_stream_t stream) {
    return integrate_common(true, dists, dists_step, cols, rows, volume, X, Y, Z, voxel_size, trunc_dist, max_weight,
                            vol2cam, fx, fy, cx, cy, nullptr, stream);
}

int dfa_tsdf_vertex_normals(const uint32_t* volume, int X, int Y, int Z, const float voxel_size[3],
                            float gradient_delta_factor, const float* points, int n, float* normals,
                            dfa_stream_t stream) {
    REQUIRE(volume && voxel_size, "null volume / voxel_size");
    REQUIRE(X > 1 && Y > 1 && Z > 1, "the volume needs two voxels per axis");
    REQUIRE(n >= 0 && (n == 0 || (points && normals)), "null points / normals");
    REQUIRE(voxel_size[0] > 0.f && voxel_size[1] > 0.f && voxel_size[2] > 0.f && gradient_delta_factor > 0.f,
            "voxel size and gradient delta must be positive");
    REQUIRE((((uintptr_t)points | (uintptr_t)normals) & 15) == 0, "points / normals must be 16-byte aligned");
    HIP_TRY(dfa::launch_vertex_normals(volume, X, Y, Z, voxel_size, gradient_delta_factor, points, n, normals, S(stream)));
    return DFA_OK;
}

int dfa_tsdf_raycast_points(const uint32_t* volume, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                            const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                            float step_factor, float delta_factor, float* points, int points_step, float* normals,
                            int normals_step, int cols, int rows, dfa_stream_t stream) {
    REQUIRE(volume_args_ok(volume, X, Y, Z), "bad volume");
    REQUIRE(X >= 2 && Y >= 2 && Z >= 2, "volume needs at least 2 voxels per axis");
    REQUIRE(points && normals && cols > 0 && rows > 0, "bad output images");
    REQUIRE(points_step >= cols * 16 && normals_step >= cols * 16, "row step smaller than a float4 row");
    REQUIRE(voxel_size && cam2vol && Rinv, "null parameter block");
    REQUIRE(trunc_dist > 0.f && step_factor > 0.f, "non-positive ray step");
    HIP_TRY(dfa::launch_raycast_points(volume, X, Y, Z, voxel_size, trunc_dist, cam2vol, Rinv, fx, fy, cx, cy,
                                       step_factor, delta_factor, points, points_step, normals, normals_step, cols,
                                       rows, S(stream)));
    return DFA_OK;
}

int dfa_tsdf_raycast_depth(const uint32_t* volume, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                           const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                           float step_factor, float delta_factor, uint16_t* depth, int depth_step, float* normals,
                           int normals_step, int cols, int rows, dfa_stream_t stream) {
    REQUIRE(volume_args_ok(volume, X, Y, Z), "bad volume");
    REQUIRE(X >= 2 && Y >= 2 && Z >= 2, "volume needs at least 2 voxels per axis");
    REQUIRE(depth && normals && cols > 0 && rows > 0, "bad output images");
    REQUIRE(depth_step >= cols * 2 && normals_step >= cols * 16, "row step smaller than a row");
    REQUIRE(voxel_size && cam2vol && Rinv, "null parameter block");
    REQUIRE(trunc_dist > 0.f && step_factor > 0.f, "non-positive ray step");
    HIP_TRY(dfa::launch_raycast_depth(volume, X, Y, Z, voxel_size, trunc_dist, cam2vol, Rinv, fx, fy, cx, cy,
                                      step_factor, delta_factor, depth, depth_step, normals, normals_step, cols, rows,
                                      S(stream)));
    return DFA_OK;
}

int dfa_tsdf_raycast_tally(const uint32_t* volume, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                           const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                           float step_factor, float delta_factor, int cols, int rows, uint64_t* counts,
                           uint32_t* touched_bits, dfa_stream_t stream) {
    REQUIRE(volume_args_ok(volume, X, Y, Z), "bad volume");
    REQUIRE(X >= 2 && Y >= 2 && Z >= 2, "volume needs at least 2 voxels per axis");
    REQUIRE(counts && cols > 0 && rows > 0, "bad counters / image size");
    REQUIRE(voxel_size && cam2vol && Rinv, "null parameter block");
    REQUIRE(trunc_dist > 0.f && step_factor > 0.f, "non-positive ray step");
    HIP_TRY(hipMemsetAsync(counts, 0, 4 * sizeof(uint64_t), S(stream)));
    HIP_TRY(dfa::launch_raycast_tally(volume, X, Y, Z, voxel_size, trunc_dist, cam2vol, Rinv, fx, fy, cx, cy, step_factor,
                                      delta_factor, cols, rows, (unsigned long long*)counts, touched_bits, S(stream)));
    return DFA_OK;
}

// -------------------------------------------------------------------- depth pre-processing seam

int dfa_depth_bilateral_filter(const uint16_t* src, int src_step, uint16_t* dst, int dst_step, int cols, int rows,
                               int kernel_size, float sigma_spatial, float sigma_depth, dfa_stream_t stream) {
    REQUIRE(src && dst && cols > 0 && rows > 0, "bad images");
    REQUIRE(src != dst, "the bilateral filter cannot run in place");
    REQUIRE(src_step >= cols * 2 && dst_step >= cols * 2, "row step smaller than a row");
    REQUIRE(kernel_size >= 1 && sigma_spatial > 0.f && sigma_depth > 0.f, "bad filter parameters");
    HIP_TRY(dfa::launch_bilateral(src, src_step, dst, dst_step, cols, rows, kernel_size, sigma_spatial, sigma_depth,
                                  S(stream)));
    return DFA_OK;
}

int dfa_depth_truncate(uint16_t* depth, int depth_step, int cols, int rows, float max_dist, dfa_stream_t stream) {
    REQUIRE(depth && cols > 0 && rows > 0 && depth_step >= cols * 2, "bad image");
    REQUIRE(max_dist >= 0.f && max_dist < 65.536f, "max_dist out of the 16-bit millimetre range");
    HIP_TRY(dfa::launch_truncate_depth(depth, depth_step, cols, rows, max_dist, S(stream)));
    return DFA_OK;
}

int dfa_depth_build_pyramid(const uint16_t* src, int src_step, int cols, int rows, uint16_t* dst, int dst_step,
                            float sigma_depth, dfa_stream_t stream) {
    REQUIRE(src && cols > 0 && rows > 0, "bad images");
    if (cols / 2 == 0 || rows / 2 == 0) return DFA_OK;  // empty output
    REQUIRE(dst, "null output");
    REQUIRE(src_step >= cols * 2 && dst_step >= (cols / 2) * 2, "row step smaller than a row");
    HIP_TRY(dfa::launch_depth_pyr(src, src_step, cols, rows, dst, dst_step, sigma_depth, S(stream)));
    return DFA_OK;
}

int dfa_compute_normals_mask_depth(uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                                   float cy, float* normals, int normals_step, dfa_stream_t stream) {
    REQUIRE(depth && normals && cols > 0 && rows > 0, "bad images");
    REQUIRE(depth_step >= cols * 2 && normals_step >= cols * 16, "row step smaller than a row");
    REQUIRE(fx != 0.f && fy != 0.f, "zero focal length");
    HIP_TRY(dfa::launch_normals_mask_depth(depth, depth_step, cols, rows, fx, fy, cx, cy, normals, normals_step, S(stream)));
    return DFA_OK;
}

int dfa_resize_depth_normals(const uint16_t* depth, int depth_step, const float* normals, int normals_step, int cols,
                             int rows, uint16_t* depth_out, int depth_out_step, float* normals_out, int normals_out_step,
                             dfa_stream_t stream) {
    REQUIRE(depth && normals && cols > 0 && rows > 0, "bad images");
    if (cols / 2 == 0 || rows / 2 == 0) return DFA_OK;  // empty output
    REQUIRE(depth_out && normals_out, "null output");
    REQUIRE(depth_step >= cols * 2 && normals_step >= cols * 16, "row step smaller than a row");
    REQUIRE(depth_out_step >= (cols / 2) * 2 && normals_out_step >= (cols / 2) * 16, "output row step smaller than a row");
    HIP_TRY(dfa::launch_resize_depth_normals(depth, depth_step, normals, normals_step, cols, rows, depth_out, depth_out_step,
                                             normals_out, normals_out_step, S(stream)));
    return DFA_OK;
}

int dfa_resize_points_normals(const float* points, int points_step, const float* normals, int normals_step, int cols,
                              int rows, float* points_out, int points_out_step, float* normals_out, int normals_out_step,
                              dfa_stream_t stream) {
    REQUIRE(points && normals && cols > 0 && rows > 0, "bad images");
    if (cols / 2 == 0 || rows / 2 == 0) return DFA_OK;  // empty output
    REQUIRE(points_out && normals_out, "null output");
    REQUIRE(points_step >= cols * 16 && normals_step >= cols * 16, "row step smaller than a row");
    REQUIRE(points_out_step >= (cols / 2) * 16 && normals_out_step >= (cols / 2) * 16, "output row step smaller than a row");
    HIP_TRY(dfa::launch_resize_points_normals(points, points_step, normals, normals_step, cols, rows, points_out,
                                              points_out_step, normals_out, normals_out_step, S(stream)));
    return DFA_OK;
}

// ------------------------------------------------------------------------------ rigid-ICP seam

namespace {
struct IcpScratch {
    float* partial = nullptr;
    size_t cap     = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        (void)hipFree(partial);
        partial = nullptr, cap = 0;
        hipError_t e = hipMalloc((void**)&partial, sizeof(float) * n);
        if (e == hipSuccess) cap = n;
        return e;
    }
};
}  // namespace

int dfa_icp_sums(int depth_variant, const void* curr, int curr_step, const float* ncurr, int ncurr_step, const void* prev,
                 int prev_step, const float* nprev, int nprev_step, int cols, int rows, const float aff[12], float fx,
                 float fy, float cx, float cy, float dist_thres, float angle_thres, float* sums27, unsigned int* matched,
                 dfa_stream_t stream) {
    REQUIRE(curr && ncurr && prev && nprev && sums27 && aff, "null argument");
    REQUIRE(cols > 0 && rows > 0, "bad image size");
    const int px = depth_variant ? 2 : 16;
    REQUIRE(curr_step >= cols * px && prev_step >= cols * px && ncurr_step >= cols * 16 && nprev_step >= cols * 16,
            "row step smaller than a row");
    REQUIRE(fx != 0.f && fy != 0.f && dist_thres >= 0.f, "bad intrinsics / threshold");
    IcpScratch& scratch = stream_scratch<IcpScratch>(S(stream));
    HIP_TRY(scratch.reserve(dfa::icp_partial_floats(cols, rows)));
    HIP_TRY(dfa::launch_icp_sums(depth_variant != 0, curr, curr_step, ncurr, ncurr_step, prev, prev_step, nprev, nprev_step,
                                 cols, rows, aff, fx, fy, cx, cy, dist_thres, angle_thres, scratch.partial, sums27,
                                 matched, S(stream)));
    return DFA_OK;
}

// ------------------------------------------------------------------------ marching-cubes seam

static int marching_cubes_common(const uint32_t* volume, int X, int Y, int Z, const float cell_size[3],
                                 const int32_t* tri_table, const int32_t* num_verts_table, float* out_points,
                                 int max_vertices, int32_t* total_vertices, const uint8_t* occupancy, dfa_stream_t stream) {
    REQUIRE(volume_args_ok(volume, X, Y, Z), "bad volume");
    REQUIRE(cell_size && tri_table && num_verts_table, "null parameter block / case tables");
    REQUIRE(max_vertices >= 0 && (max_vertices == 0 || out_points), "bad output buffer");
    REQUIRE((long)X * Y * Z / 64 < (1L << 31), "volume too large");
    const bool vec4  = (X % 4 == 0) && (((uintptr_t)volume & 15) == 0);
    const long nsegs = dfa::mc_segments(X, Y, Z, vec4);
    McScratch& scratch = stream_scratch<McScratch>(S(stream));
    HIP_TRY(scratch.reserve(nsegs));
    HIP_TRY(dfa::launch_marching_cubes(volume, X, Y, Z, cell_size, tri_table, num_verts_table, out_points,
                                       max_vertices, total_vertices, scratch.seg_off, scratch.chunk_sums, occupancy,
                                       S(stream)));
    return DFA_OK;
}

int dfa_marching_cubes(const uint32_t* volume, int X, int Y, int Z, const float cell_size[3],
                       const int32_t* tri_table, const int32_t* num_verts_table, float* out_points, int max_vertices,
                       int32_t* total_vertices, dfa_stream_t stream) {
    return marching_cubes_common(volume, X, Y, Z, cell_size, tri_table, num_verts_table, out_points, max_vertices,
                                 total_vertices, nullptr, stream);
}

int dfa_marching_cubes_occ(const uint32_t* volume, const uint8_t* occupancy, int X, int Y, int Z, const float cell_size[3],
                           const int32_t* tri_table, const int32_t* num_verts_table, float* out_points, int max_vertices,
                           int32_t* total_vertices, dfa_stream_t stream) {
    REQUIRE(occupancy, "null occupancy map");
    return marching_cubes_common(volume, X, Y, Z, cell_size, tri_table, num_verts_table, out_points, max_vertices,
                                 total_vertices, occupancy, stream);
}

int dfa_mc_default_tables(int32_t* tri_table, int32_t* num_verts_table) {
    REQUIRE(tri_table && num_verts_table, "null table");
    dfa::mc_default_tables(tri_table, num_verts_table);
    return DFA_OK;
}

// ---------------------------------------------------------------------------- warp-field seam

int dfa_knn(const float* node_pos, const float* node_w, int D, const float* query, int n_query, int k, int32_t* idx,
            float* weights, dfa_stream_t stream) {
    REQUIRE(node_pos && D > 0, "no nodes");
    REQUIRE(n_query >= 0 && (n_query == 0 || (query && idx)), "bad query / output");
    REQUIRE(k >= 1 && k <= DFA_MAX_KNN, "k out of range 1..16");
    REQUIRE(!weights || node_w, "weights requested without node_w");
    const dfa::KnnGridView* grid = nullptr;
    if (want_grid(D, n_query)) {
        GridScratch& gs = stream_scratch<GridScratch>(S(stream));

        HIP_TRY(gs.reserve(D));

        HIP_TRY(dfa::knn_grid_build(gs.v, node_pos, D, S(stream)));

        grid = &gs.v;
    }
    HIP_TRY(dfa::launch_knn(node_pos, node_w, D, query, n_query, k, idx, weights, grid, S(stream)));
    return DFA_OK;
}

int dfa_warp_to_live(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                     const float* vertices, const float* normals, int N, float* out_vertices, float* out_normals,
                     dfa_stream_t stream) {
    REQUIRE(node_pos && node_dq && node_w && D > 0, "no nodes");
    REQUIRE(N >= 0 && (N == 0 || (vertices && out_vertices)), "bad vertices / output");
    REQUIRE(k >= 1 && k <= DFA_MAX_KNN, "k out of range 1..16");
    const dfa::KnnGridView* grid = nullptr;
    if (want_grid(D, N)) {
        GridScratch& gs = stream_scratch<GridScratch>(S(stream));

        HIP_TRY(gs.reserve(D));

        HIP_TRY(dfa::knn_grid_build(gs.v, node_pos, D, S(stream)));

        grid = &gs.v;
    }
    HIP_TRY(dfa::launch_warp_to_live(node_pos, node_dq, node_w, D, k, vertices, normals, N, out_vertices,
                                     out_normals, grid, S(stream)));
    return DFA_OK;
}

int dfa_warp_to_live_graph(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                           const int32_t* idx, const float* vertices, const float* normals, int N, float* out_vertices,
                           float* out_normals, dfa_stream_t stream) {
    REQUIRE(node_pos && node_dq && node_w && D > 0, "no nodes");
    REQUIRE(N >= 0 && (N == 0 || (vertices && out_vertices && idx)), "bad vertices / graph / output");
    REQUIRE(k >= 1 && k <= DFA_MAX_KNN, "k out of range 1..16");
    if (N > 0)
        HIP_TRY(dfa::launch_warp_graph(node_pos, node_dq, node_w, k, idx, vertices, normals, N, out_vertices, out_normals,
                                       S(stream)));
    return DFA_OK;
}

int dfa_calc_dqb(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, const float* points,
                 int n, float* out_dq, dfa_stream_t stream) {
    REQUIRE(node_pos && node_dq && node_w && D > 0, "no nodes");
    REQUIRE(n >= 0 && (n == 0 || (points && out_dq)), "bad points / output");
    REQUIRE(k >= 1 && k <= DFA_MAX_KNN, "k out of range 1..16");
    const dfa::KnnGridView* grid = nullptr;
    if (want_grid(D, n)) {
        GridScratch& gs = stream_scratch<GridScratch>(S(stream));

        HIP_TRY(gs.reserve(D));

        HIP_TRY(dfa::knn_grid_build(gs.v, node_pos, D, S(stream)));

        grid = &gs.v;
    }
    HIP_TRY(dfa::launch_dqb_support(node_pos, node_dq, node_w, D, k, points, n, out_dq, nullptr, grid, S(stream)));
    return DFA_OK;
}

int dfa_unsupported_vertices(const float* node_pos, const float* node_w, int D, int k, const float* vertices, int N,
                             uint8_t* flags, dfa_stream_t stream) {
    REQUIRE(N >= 0 && (N == 0 || (vertices && flags)), "bad vertices / output");
    REQUIRE(k >= 1 && k <= DFA_MAX_KNN, "k out of range 1..16");
    if (D == 0) {  // no node supports anything (min stays HUGE_VALF, warp_field.cpp:40,53)
        if (N > 0) HIP_TRY(hipMemsetAsync(flags, 1, (size_t)N, S(stream)));
        return DFA_OK;
    }
    REQUIRE(node_pos && node_w && D > 0, "bad nodes");
    const dfa::KnnGridView* grid = nullptr;
    if (want_grid(D, N)) {
        GridScratch& gs = stream_scratch<GridScratch>(S(stream));

        HIP_TRY(gs.reserve(D));

        HIP_TRY(dfa::knn_grid_build(gs.v, node_pos, D, S(stream)));

        grid = &gs.v;
    }
    HIP_TRY(dfa::launch_dqb_support(node_pos, nullptr, node_w, D, k, vertices, N, nullptr, flags, grid, S(stream)));
    return DFA_OK;
}

int dfa_repack_points(const float* src, int src_stride, float* dst, int dst_stride, int n, float pad,
                      dfa_stream_t stream) {
    REQUIRE(n >= 0 && (n == 0 || (src && dst)), "bad points");
    REQUIRE(src_stride >= 3 && dst_stride >= 3, "strides are floats per point, >= 3");
    REQUIRE(src_stride != 4 || ((uintptr_t)src & 15) == 0, "float4 source must be 16-byte aligned");
    REQUIRE(dst_stride != 4 || ((uintptr_t)dst & 15) == 0, "float4 destination must be 16-byte aligned");
    HIP_TRY(dfa::launch_repack_points(src, src_stride, dst, dst_stride, n, pad, S(stream)));
    return DFA_OK;
}

int dfa_transform_points(const float* points, int n, const float aff[12], int with_translation, float* out,
                         dfa_stream_t stream) {
    REQUIRE(n >= 0 && (n == 0 || (points && out)), "bad points");
    REQUIRE(aff, "null transform");
    HIP_TRY(dfa::launch_transform_points(points, n, aff, with_translation != 0, out, S(stream)));
    return DFA_OK;
}

namespace {
struct CompactScratch {
    int32_t* chunks = nullptr;
    int cap         = 0;
    hipError_t reserve(int n_needed) {
        if (n_needed <= cap) return hipSuccess;
        const int n = n_needed + n_needed / 4 + 16;
        (void)hipFree(chunks);
        chunks = nullptr, cap = 0;
        hipError_t e = hipMalloc((void**)&chunks, sizeof(int32_t) * (size_t)n);
        if (e == hipSuccess) cap = n;
        return e;
    }
};
}  // namespace

int dfa_compact_points(const float* points, const uint8_t* flags, int N, float* out_points, int32_t* out_index,
                       int32_t* count, dfa_stream_t stream) {
    REQUIRE(count, "count is required");
    REQUIRE(N >= 0 && (N == 0 || flags), "bad flags");
    REQUIRE(!out_points || points || N == 0, "out_points requested without points");
    CompactScratch& cs = stream_scratch<CompactScratch>(S(stream));
    HIP_TRY(cs.reserve(dfa::compact_chunks(N > 0 ? N : 1)));
    HIP_TRY(dfa::launch_compact_points(points, flags, N, out_points, out_index, count, cs.chunks, S(stream)));
    return DFA_OK;
}

int dfa_correspond(const float* canon_vertices, const float* canon_normals, int n_canon, const float* live_vertices,
                   int n_live, float* out_vertices, float* out_normals, int32_t* out_index, dfa_stream_t stream) {
    REQUIRE(canon_vertices && n_canon > 0, "no canonical vertices");
    REQUIRE(n_live >= 0 && (n_live == 0 || live_vertices), "bad live vertices");
    REQUIRE(!out_normals || canon_normals, "normals requested without canonical normals");
    const dfa::KnnGridView* grid = nullptr;
    if (n_canon >= 16384 && want_grid(n_canon, n_live)) {  // large cloud: 128^3 point grid
        PointGridScratch& pg = stream_scratch<PointGridScratch>(S(stream));
        HIP_TRY(pg.reserve(n_canon));
        HIP_TRY(dfa::point_grid_build(pg.v, canon_vertices, n_canon, S(stream)));
        grid = &pg.v.g;
    } else if (want_grid(n_canon, n_live)) {
        GridScratch& gs = stream_scratch<GridScratch>(S(stream));

        HIP_TRY(gs.reserve(n_canon));

        HIP_TRY(dfa::knn_grid_build(gs.v, canon_vertices, n_canon, S(stream)));

        grid = &gs.v;
    }
    HIP_TRY(dfa::launch_correspond(canon_vertices, canon_normals, n_canon, live_vertices, n_live, out_vertices,
                                   out_normals, out_index, grid, S(stream)));
    return DFA_OK;
}

int dfa_correspond_projective(const float* vertices, const float* normals, int n, const float* vmap, int vmap_step,
                              const float* nmap, int nmap_step, int cols, int rows, float fx, float fy, float cx, float cy,
                              float dist_thresh, float min_cosine, float* out_vertices, float* out_normals,
                              int32_t* out_pixel, dfa_stream_t stream) {
    REQUIRE(n >= 0 && (n == 0 || vertices), "bad vertices");
    REQUIRE(vmap && cols > 0 && rows > 0 && vmap_step >= 16 * cols, "bad vertex map");
    REQUIRE(!nmap || nmap_step >= 16 * cols, "bad normal map pitch");
    REQUIRE(!out_normals || nmap, "normals requested without a normal map");
    REQUIRE(fx > 0.f && fy > 0.f && dist_thresh >= 0.f, "bad intrinsics / threshold");
    REQUIRE(((uintptr_t)vmap & 15) == 0 && (vmap_step & 15) == 0 && ((uintptr_t)nmap & 15) == 0 && (!nmap || (nmap_step & 15) == 0),
            "maps must be 16-byte aligned float4 images");
    HIP_TRY(dfa::launch_correspond_projective(vertices, normals, n, vmap, vmap_step, nmap, nmap_step, cols, rows, fx, fy, cx,
                                              cy, dist_thresh, min_cosine, out_vertices, out_normals, out_pixel,
                                              S(stream)));
    return DFA_OK;
}

// ------------------------------------------------------------------------------- solver seam

int dfa_solver_create(int max_D, int max_N, int k, dfa_solver** out) {
    REQUIRE(out, "null out");
    *out = nullptr;
    REQUIRE(max_D > 0 && max_N >= 0, "bad sizes");
    REQUIRE(k >= 1 && k <= DFA_MAX_KNN, "k out of range 1..16");
    REQUIRE(max_D <= dfa::solve_pcg_max_nodes(), "more nodes than a plan supports (32768)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DFA_ERR_NO_GPU, "no HIP device");
    dfa_solver* s = new (std::nothrow) dfa_solver();
    REQUIRE(s, "out of host memory");
    s->max_D = max_D, s->max_N = max_N, s->k = k;
    s->max_R       = (size_t)max_N + (size_t)max_D * k;
    s->ell_cap     = 256;
    s->has_problem = false;
    s->timing      = false;
    {
        // (the one environment variable of the product library, read once per plan: the order-stable assembly for a caller
        // that cannot reach dfa_solver_set_deterministic — include/dynfu_amd.h)
        const char* e    = getenv("DFA_ASSEMBLE_DETERMINISTIC");
        s->deterministic = e && atoi(e) != 0;
    }
    s->ev_used     = 0;
    std::memset(&s->v, 0, sizeof(s->v));
    const size_t R = s->max_R, D = (size_t)max_D;
    int rc = DFA_OK;
#define A(field, count) \
    if (rc == DFA_OK) rc = plan_alloc(s, &s->v.field, (count))
    A(ridx, R * k);
    A(rw, R * k);
    A(rtau, R);
    A(rb, R * 3);
    A(re, R * (size_t)dfa::solve_rec_words(k));
    A(reg_idx, D * k);
    A(blk_hist, D * (dfa::SOLVE_TG_BLOCKS + 1));
    A(node_ptr, D + 1);
    A(node_list, R * k);
    A(ell, D * s->ell_cap);
    A(ell_cnt, D);
    A(diag, D);
    A(g, D * 3);
    A(g_base, D * 3);
    A(t_base, D * 3);
    A(pk_perm, D);
    A(pk_perm2, D);
    A(pk_vals, D * s->ell_cap);
    A(pk_cols, D * s->ell_cap);
    A(mb_x, D);
    A(mb_r, D);
    A(mb_p, D);
    A(mb_s, D);
    A(mb_w, D);
    A(mb_u[0], D);
    A(mb_u[1], D);
    A(mb_m[0], D);
    A(mb_m[1], D);
    A(mb_t[0], D);
    A(mb_t[1], D);
    A(mb_gpart[0], (size_t)dfa::solve_mb_blocks(max_D));
    A(mb_gpart[1], (size_t)dfa::solve_mb_blocks(max_D));
    A(mb_dpart[0], (size_t)dfa::solve_mb_blocks(max_D));
    A(mb_dpart[1], (size_t)dfa::solve_mb_blocks(max_D));
    A(t, D * 3);
    A(huber, D);
    A(node_dq_out, D * 8);
#undef A
    if (rc == DFA_OK) {
        hipError_t e = s->grid.reserve(max_D);
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc (node grid)");
    }
    // (plans of up to 2 048 nodes solve in the register-resident kernels: no team buffers for them outside development builds)
    if (rc == DFA_OK && dfa::solve_team_pcg_fits(max_D) && (max_D > 2048 || dfa::kDevAB)) {
        // team PCG: control block (zeroed once: barrier rounds only ever grow), exchange buffer, the pinned abort count
        s->v.team_stride = (max_D + 3) & ~3;
        rc = plan_alloc(s, &s->v.team_ctl, 1);
        const size_t areas = (size_t)3 * dfa::solve_team_pcg_rounds();  // an area per barrier round and coordinate (50 MB at 8 k nodes)
        if (rc == DFA_OK) rc = plan_alloc(s, &s->v.team_mt, areas * s->v.team_stride);
        if (rc == DFA_OK) rc = plan_alloc(s, &s->v.team_words, dfa::solve_team_pcg_words());
        if (rc == DFA_OK && (hipMemset(s->v.team_ctl, 0, sizeof(dfa::TeamCtl)) != hipSuccess ||
                             hipMemset(s->v.team_words, 0, sizeof(unsigned long long) * dfa::solve_team_pcg_words()) != hipSuccess ||
                             hipMemset(s->v.team_mt, 0, sizeof(float2) * areas * (size_t)s->v.team_stride) != hipSuccess))
            rc = fail(DFA_ERR_HIP, "hipMemset (team PCG)");
        if (rc == DFA_OK && hipHostMalloc((void**)&s->team.host_abort, sizeof(int), hipHostMallocDefault) == hipSuccess) {
            *s->team.host_abort = 0;
            s->team.ctl = s->v.team_ctl;
        }
    }
    if (rc == DFA_OK) rc = plan_alloc(s, &s->state, 1);
    if (rc == DFA_OK) rc = plan_alloc(s, &s->iters_total, 1);
    if (rc == DFA_OK && hipMemset(s->iters_total, 0, sizeof(long long)) != hipSuccess) rc = DFA_ERR_HIP;
    if (rc == DFA_OK && hipHostMalloc((void**)&s->host_flag, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) s->host_flag = nullptr;
    // (per linearise workgroup, at most 1024 of them: its share of the energy; behind those its largest matrix addend)
    if (rc == DFA_OK) rc = plan_alloc(s, &s->cost_partials, std::max<size_t>((R + 255) / 256 + 1, 1024) + 1024);
    if (rc == DFA_OK) rc = plan_alloc(s, &s->ticket, 64);
    if (rc == DFA_OK && hipMemset(s->ticket, 0, 64 * sizeof(unsigned int)) != hipSuccess)
        rc = fail(DFA_ERR_HIP, "hipMemset (ticket)");
    if (rc != DFA_OK) {
        dfa_solver_destroy(s);
        return rc;
    }
    *out = s;
    return DFA_OK;
}

void dfa_solver_destroy(dfa_solver* s) {
    if (!s) return;
    for (void* p : s->blocks) (void)hipFree(p);
    s->grid.release();
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    if (s->host_flag) (void)hipHostFree(s->host_flag);
    if (s->team.host_abort) (void)hipHostFree(s->team.host_abort);
    s->mb_graphs.release();
    delete s;
}

int dfa_solver_set_problem(dfa_solver* s, const float* node_pos, const float* node_dq, const float* node_w, int D,
                           const float* canon_vertices, const float* canon_normals, const float* live_vertices,
                           const float* live_normals, int N, dfa_stream_t stream) {
    (void)canon_normals;
    (void)live_normals;  // declared by energy.t:28-31, never read by an Energy term
    REQUIRE(s, "null plan");
    REQUIRE(node_pos && node_dq && node_w && D > 0, "no nodes");
    REQUIRE(N >= 0 && (N == 0 || (canon_vertices && live_vertices)), "bad vertex arrays");
    if (D > s->max_D || N > s->max_N)
        return fail(DFA_ERR_CAPACITY, "dfa_solver_set_problem: D=%d N=%d exceed the plan (max_D=%d max_N=%d)", D, N,
                    s->max_D, s->max_N);
    dfa::SolveView& v = s->v;
    v.N = N, v.D = D, v.k = s->k, v.Dpad = (D + 3) & ~3, v.ell_cap = s->ell_cap;
    v.deterministic = s->deterministic ? 1 : 0;
    v.node_pos = node_pos, v.node_dq = node_dq, v.node_w = node_w;
    v.canon = canon_vertices, v.live = live_vertices;
    hipStream_t st = S(stream);
    // initializeDataGraph (opt_solver.cpp:56-72): rows [0, N) = k-NN of the canonical vertices + RBF weights
    const dfa::KnnGridView* grid = nullptr;
    if (want_grid(D, (long)N + D)) {
        HIP_TRY(dfa::knn_grid_build(s->grid.v, node_pos, D, st));
        grid = &s->grid.v;
    }
    if (N > 0) HIP_TRY(dfa::launch_knn(node_pos, node_w, D, canon_vertices, N, s->k, v.ridx, v.rw, grid, st));
    // initializeRegGraph (:74-105): k-NN of every node among the nodes (itself included at distance 0)
    HIP_TRY(dfa::launch_knn(node_pos, node_w, D, node_pos, D, s->k, v.reg_idx, nullptr, grid, st));
    // rows + transposition; resetGPUMemory (:149-202): unknowns start at zero (same launch as the row set-up)
    HIP_TRY(dfa::solve_build_graph(v, s->state, s->ticket, 64, st));
    s->has_problem = true;
    s->just_reset  = true;
    return DFA_OK;
}

int dfa_solver_set_deterministic(dfa_solver* s, int on) {
    REQUIRE(s, "null plan");
    s->deterministic = on != 0;
    return DFA_OK;
}

int dfa_solver_solve(dfa_solver* s, const dfa_solve_params* p, dfa_stream_t stream) {
    REQUIRE(s && p, "null plan / params");
    REQUIRE(s->has_problem, "set_problem has not been called");
    REQUIRE(p->num_iter >= 0 && p->nonlinear_iter >= 0 && p->linear_iter >= 0, "negative iteration count");
    REQUIRE(p->tukey_offset > 0.f && p->psi_data > 0.f, "tukey_offset / psi_data must be positive");
    REQUIRE(p->lambda >= 0.f && std::isfinite(p->lambda), "lambda must be non-negative and finite");
    const dfa::SolveView& v = s->v;
    hipStream_t st          = S(stream);
    if (!s->just_reset) HIP_TRY(dfa::solve_reset(v, s->state, s->ticket, 64, st));  // (set_problem has just done it)
    s->just_reset = false;
    // w_reg = sqrt(lambda / (D * KNN))  (opt_solver.cpp:30); rows carry tau = w_reg^2
    const double w_reg    = std::sqrt((double)p->lambda / ((double)v.D * (double)v.k));
    const float w_reg_f   = (float)w_reg;
    const float w_reg_sq  = w_reg_f * w_reg_f;
    if (s->timing) s->timed_solves += 1;
    if (s->host_flag) s->host_flag[0] = s->host_flag[1] = 0;
    s->mb_graphs.call = 0;
    int not_launched = 0;  // iterations after the host has seen the converged flag (many-workgroup path only)
    int gn_launched = 0;  // Gauss-Newton iterations whose assembly has been enqueued
    bool huber_done = false;
    const bool big_budget = (long)p->num_iter * p->nonlinear_iter > 8;
    const bool pcg_async  = dfa::solve_pcg_is_async(v, &s->team, p->linear_iter);  // (else the PCG itself reads the flags back)
    const bool no_regradient = dfa::dev_env("DFA_NO_REGRADIENT") != nullptr;  // (development builds: the tests compare both ways)
    for (int outer = 0; outer < p->num_iter; ++outer) {
        // preNonlinearSolve (opt_solver.cpp:135-140): the Huber weights are only observable after
        // the solve, so they are evaluated for the last outer iteration alone
        // (folded into that iteration's first linearisation; a launch of its own only if that one is never enqueued)
        // converged == 2 (gradient at the floor under stale weights) ended the inner iterations of the outer iteration
        // before only: this one re-weights, and its first linearisation clears the flag on the device
        if (s->host_flag && s->host_flag[1] == 2) s->host_flag[1] = 0;
        int gn_in_outer = 0;  // iterations of this outer iteration enqueued (the device's flag is current behind them)
        for (int gn = 0; gn < p->nonlinear_iter; ++gn) {
            // Iteration budgets like the reference's (24 x 16, dyn_fusion.cpp:183-189) are ~380 iterations of which a
            // handful do anything: behind a converged one the kernels return at entry, but 4 launches x 5 us x 380 is
            // still 8 ms of stream time.  Small budgets (<= 8 iterations: bench.py's 5) stay free of any host
            // synchronisation; larger ones read the `converged` flag back every 4th iteration (plans whose PCG does not
            // synchronise by itself: the register-resident kernels and the team form).
            if (big_budget && gn_in_outer >= 4 && gn_in_outer % 4 == 0 && s->host_flag && !s->host_flag[1] && pcg_async) {
                HIP_TRY(hipMemcpyAsync(&s->host_flag[1], &s->state->converged, sizeof(int), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
            if (s->host_flag && s->host_flag[1]) {  // t can no longer change (in this outer iteration, when the flag is
                ++not_launched;                     // 2): every further iteration is a no-op
                continue;
            }
            ++gn_in_outer;
            // An inner iteration (frozen robust weights: linear least squares) restarts the PCG on the true residual of
            // the SAME matrix, g = g_base - A (t - t_base): one sparse product instead of a linearisation and an assembly
            // (the reference's budget is 16 inner iterations per re-weighting).  With a cost-decrease tolerance the cost
            // of every iteration is wanted: the long way.  Without the regulariser (lambda = 0: the reference's OptTests,
            // fewer vertices than nodes) the matrix is singular, and the cancellation in g_base - A (t - t_base) leaves
            // round-off in its null space, which CG then amplifies (a re-linearised gradient is J^T of something and has
            // none; measured: two of the eight OptTest scenes leave their 1e-3 tolerance): the long way as well.
            // DFA_NO_REGRADIENT=1: the long way always (A/B).
            if (gn > 0 && p->gn_tol == 0.f && p->lambda > 0.f && !no_regradient) {
                int evg = s->timing ? timing_begin(s, st) : -1;  // booked with the assemblies: it stands in for one
                HIP_TRY(dfa::solve_regradient(v, s->state, st));
                timing_end(s, evg, st);
                if (evg >= 0) s->ev_asm.push_back(evg);
                if (s->overlap_fn) s->overlap_fn(s->overlap_user, stream, gn_launched);
                ++gn_launched;
                int evr = s->timing ? timing_begin(s, st) : -1;
                HIP_TRY(dfa::solve_pcg(v, s->state, p->linear_iter, p->pcg_tol, s->host_flag, &s->mb_graphs, &s->team,
                                       evr >= 0 ? s->events[evr + 1] : nullptr, st));
                if (evr >= 0) s->ev_pcg.push_back(evr);
                continue;
            }
            const bool with_huber = gn == 0 && outer == p->num_iter - 1;
            HIP_TRY(dfa::solve_linearise(v, s->state, s->cost_partials, s->ticket, gn == 0, gn == 0 ? 0 : 1,
                                         p->gn_tol, p->tukey_offset, p->psi_data, w_reg_sq, with_huber ? p->psi_reg : 0.f, nullptr, st));
            huber_done |= with_huber;
            int ev = s->timing ? timing_begin(s, st) : -1;
            HIP_TRY(dfa::solve_assemble(v, s->state, gn == 0 && p->nonlinear_iter > 1, w_reg_sq, st));
            timing_end(s, ev, st);
            if (ev >= 0) s->ev_asm.push_back(ev);
            // this iteration's PCG starts here: the caller's chip-wide work may run in its shadow
            if (s->overlap_fn) s->overlap_fn(s->overlap_user, stream, gn_launched);
            ++gn_launched;
            ev = s->timing ? timing_begin(s, st) : -1;  // closed behind the solving kernel, before the fallback launch
            HIP_TRY(dfa::solve_pcg(v, s->state, p->linear_iter, p->pcg_tol, s->host_flag, &s->mb_graphs, &s->team,
                                   ev >= 0 ? s->events[ev + 1] : nullptr, st));
            if (ev >= 0) s->ev_pcg.push_back(ev);
        }
    }
    if (s->overlap_fn && gn_launched == 0) s->overlap_fn(s->overlap_user, stream, -1);  // no iteration ran: the caller's work still goes out
    if (not_launched) HIP_TRY(dfa::solve_count_noop(s->state, not_launched, st));
    // final cost at the solved t; weights re-evaluated only if no iteration ever did
    const bool no_weights = p->num_iter == 0 || p->nonlinear_iter == 0;
    if (!huber_done) HIP_TRY(dfa::solve_huber(v, p->psi_reg, st));  // no outer iteration, or the host stopped launching before the last
    HIP_TRY(dfa::solve_linearise(v, s->state, s->cost_partials, s->ticket, no_weights ? 1 : 0, 2, 0.f,
                                 p->tukey_offset, p->psi_data, w_reg_sq, 0.f, s->timing ? s->iters_total : nullptr, st));
    // (postSingleSolve -> copyResultToCPUFromFloat3, opt_solver.cpp:133,270-285: composed once, by that same launch)
    return DFA_OK;
}

int dfa_solver_set_overlap_callback(dfa_solver* s, dfa_overlap_fn fn, void* user) {
    REQUIRE(s, "null plan");
    s->overlap_fn = fn, s->overlap_user = user;
    return DFA_OK;
}

const float* dfa_solver_translations(const dfa_solver* s) { return s ? s->v.t : nullptr; }
const float* dfa_solver_node_dq(const dfa_solver* s) { return s ? s->v.node_dq_out : nullptr; }
const float* dfa_solver_tukey_weights(const dfa_solver* s) { return s ? s->v.rtau : nullptr; }
const float* dfa_solver_huber_weights(const dfa_solver* s) { return s ? s->v.huber : nullptr; }
const int32_t* dfa_solver_data_graph(const dfa_solver* s) { return s ? s->v.ridx : nullptr; }
const int32_t* dfa_solver_reg_graph(const dfa_solver* s) { return s ? s->v.reg_idx : nullptr; }
const float* dfa_solver_matrix_entries(const dfa_solver* s) { return s ? (const float*)s->v.ell : nullptr; }
const int32_t* dfa_solver_matrix_row_lengths(const dfa_solver* s) { return s ? s->v.ell_cnt : nullptr; }
const float* dfa_solver_gradient(const dfa_solver* s) { return s ? s->v.g : nullptr; }

int dfa_solver_warp_to_live(dfa_solver* s, const float* normals, float* out_vertices, float* out_normals,
                            dfa_stream_t stream) {
    REQUIRE(s && s->has_problem, "no problem set");
    REQUIRE(out_vertices || s->v.N == 0, "null output");
    HIP_TRY(dfa::launch_warp_graph(s->v.node_pos, s->v.node_dq_out, s->v.node_w, s->k, s->v.ridx, s->v.canon, normals,
                                   s->v.N, out_vertices, out_normals, S(stream)));
    return DFA_OK;
}

int dfa_solver_get_stats(dfa_solver* s, dfa_solve_stats* host_out, dfa_stream_t stream) {
    REQUIRE(s && host_out, "null plan / out");
    dfa::SolveState h;
    HIP_TRY(hipMemcpyAsync(&h, s->state, sizeof(h), hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    host_out->initial_cost = h.initial_cost;
    host_out->final_cost   = h.final_cost;
    host_out->gn_iters     = h.gn_iters;
    host_out->pcg_iters    = h.pcg_iters;
    host_out->max_row_nnz  = h.max_row_nnz;
    host_out->gn_noop      = h.gn_noop;
    if (dfa::dev_env("DFA_PCG_PROFILE_PRINT"))
        fprintf(stderr, "pcg phase cycles: spmv %lld  red_pAp %lld  update %lld  red_rz %lld  p_update+barrier %lld  loop %lld  (iters %d)\n",
                h.prof[0], h.prof[1], h.prof[2], h.prof[3], h.prof[4], h.prof[5], h.pcg_iters);
    if (dfa::dev_env("DFA_PCG_PROFILE_PRINT"))
        fprintf(stderr, "assemble block 7 cycles: init %lld  list+hash %lld  reduce+barrier %lld  compact %lld   (raw prof[6] %lld prof[7] %lld)\n",
                h.prof[6] / 1000000, h.prof[6] % 1000000, h.prof[7] / 1000000, h.prof[7] % 1000000, h.prof[6], h.prof[7]);
    if (h.overflow)
        return fail(DFA_ERR_CAPACITY, "normal-matrix row wider than the plan's ELL capacity (%d > %d)", h.max_row_nnz,
                    s->ell_cap);
    return DFA_OK;
}

int dfa_solver_enable_timing(dfa_solver* s, int enable) {
    REQUIRE(s, "null plan");
    s->timing = enable != 0;
    if (enable == 1) {  // a new measurement: forget the brackets collected so far (2 = resume, keeps them)
        s->ev_used = 0, s->timed_solves = 0;
        s->ev_pcg.clear(), s->ev_asm.clear();
        if (s->iters_total) HIP_TRY(hipMemset(s->iters_total, 0, sizeof(long long)));
    }
    return DFA_OK;
}

int dfa_solver_team_pcg_info(dfa_solver* s, int* launches, int* aborts, int* disabled) {
    REQUIRE(s, "null plan");
    if (launches) *launches = (int)std::min<long>(s->team.launches, 0x7fffffffL);
    if (aborts) *aborts = s->team.host_abort ? *(volatile int*)s->team.host_abort : 0;
    if (disabled) *disabled = (!s->team.ctl || s->team.disabled) ? 1 : 0;
    return DFA_OK;
}

int dfa_solver_get_timing(dfa_solver* s, dfa_solve_timing* out, dfa_stream_t stream) {
    REQUIRE(s && out, "null plan / out");
    HIP_TRY(hipStreamSynchronize(S(stream)));
    std::memset(out, 0, sizeof(*out));
    for (int idx : s->ev_pcg) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->events[idx], s->events[idx + 1]));
        out->pcg_ms += ms;
    }
    for (int idx : s->ev_asm) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->events[idx], s->events[idx + 1]));
        out->assemble_ms += ms;
    }
    out->pcg_launches      = (int)s->ev_pcg.size();
    out->assemble_launches = (int)s->ev_asm.size();
    out->solves            = s->timed_solves;
    if (s->iters_total) HIP_TRY(hipMemcpy(&out->pcg_iters, s->iters_total, sizeof(long long), hipMemcpyDeviceToHost));
    if (s->has_problem && s->v.D > 0) {
        std::vector<int32_t> cnt((size_t)s->v.D);
        HIP_TRY(hipMemcpy(cnt.data(), s->v.ell_cnt, sizeof(int32_t) * cnt.size(), hipMemcpyDeviceToHost));
        long long nnz = 0;
        for (int32_t c : cnt) nnz += c;
        out->matrix_nnz = nnz;
    }
    return DFA_OK;
}

// -------------------------------------------------------------------- north-star solver seam

int dfa_compute_points_normals(const uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                               float cy, float* points, int points_step, float* normals, int normals_step,
                               dfa_stream_t stream) {
    REQUIRE(depth && points && normals && cols > 0 && rows > 0, "bad images");
    REQUIRE(depth_step >= cols * 2 && points_step >= cols * 16 && normals_step >= cols * 16, "row step smaller than a row");
    REQUIRE(fx != 0.f && fy != 0.f, "zero focal length");
    HIP_TRY(dfa::launch_points_normals(depth, depth_step, cols, rows, fx, fy, cx, cy, points, points_step, normals,
                                       normals_step, S(stream)));
    return DFA_OK;
}

int dfa_solver6_create(int max_D, int max_N, int k, dfa_solver6** out) {
    REQUIRE(out, "null out");
    *out = nullptr;
    REQUIRE(max_D > 0 && max_N >= 0, "bad sizes");
    REQUIRE(k >= 1 && k <= 8, "k out of range 1..8");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DFA_ERR_NO_GPU, "no HIP device");
    dfa_solver6* s = new (std::nothrow) dfa_solver6();
    REQUIRE(s, "out of host memory");
    s->max_D = max_D, s->max_N = max_N, s->k = k, s->has_problem = false, s->node_dq = nullptr;
    std::memset(&s->v, 0, sizeof(s->v));
    s->v.cap = 48;  // = S6_MAXSLOT of solve6.hip
    const size_t N = (size_t)max_N, D = (size_t)max_D, cap = (size_t)s->v.cap;
    int rc = DFA_OK;
#define A(field, count) \
    if (rc == DFA_OK) rc = plan6_alloc(s, &s->v.field, (count))
    A(idx, N * k);
    A(wn, N * k);
    A(idx_nat, N * k);
    A(near, N);
    A(vptr, D + 1);
    A(vlist, N);
    A(vperm, N);
    A(canon_own, N * 3);
    A(canon_n_own, N * 3);
    A(reg_idx, D * k);
    A(blk_hist, D * (dfa::SOLVE_TG_BLOCKS + 1));
    A(node_ptr, D + 1);
    A(node_list, N * k + 1);  // (+ 1: the assembly reads one entry even of an empty list)
    A(rnode_ptr, D + 1);
    A(rnode_list, D * k + 1);  // (+ 1: the assembly reads one entry even of an empty list)
    A(dq, D * 8);
    A(dq_prev, D * 8);
    A(ghat, D * 3);
    A(rec, N * (12 + (k <= 4 ? 4 : 8)));
    A(cost_part, (N + 255) / 256 + (D * k + 255) / 256 + 1);
    A(valid_part, (N + 255) / 256 + (D * k + 255) / 256 + 1);
    A(mnode, D * 48);
    A(rho, N);
    A(rres, D * k * 3);
    A(rvec, D * k * 18);
    A(rhub, D * k);
    A(bcols, D * cap);
    A(bcnt, D);
    A(bfu, D);
    A(rslot, D * cap);
    A(utab, D * 256);
    A(eslot, N * k * k);
    A(pair_list, N * k * k);
    A(pair_ptr, D * (cap + 1));
    A(bvals, D * cap * 36);
    A(minv, D * 36);
    A(g, D * 6);
    A(x, D * 6);
    A(r, D * 6);
    A(p, D * 6);
    A(s, D * 6);
    A(w, D * 6);
    A(u[0], D * 6);
    A(u[1], D * 6);
    A(t[0], D * 6);
    A(t[1], D * 6);
    A(m[0], D * 6);
    A(m[1], D * 6);
    A(g_part[0], (size_t)dfa::s6_matvec_blocks(max_D));
    A(g_part[1], (size_t)dfa::s6_matvec_blocks(max_D));
    A(d_part[0], (size_t)dfa::s6_matvec_blocks(max_D));
    A(d_part[1], (size_t)dfa::s6_matvec_blocks(max_D));
#undef A
    if (rc == DFA_OK) rc = plan6_alloc(s, &s->raw_w, N * k);
    if (rc == DFA_OK) rc = plan6_alloc(s, &s->raw_reg, D * (k + 1));
    if (rc == DFA_OK) rc = plan6_alloc(s, &s->state, 1);
    if (rc == DFA_OK) {
        bool ok = hipHostMalloc((void**)&s->mirror, dfa_solver6::S6_RING * dfa::S6_HIST * sizeof(int), hipHostMallocDefault) == hipSuccess;
        for (int i = 0; ok && i < dfa_solver6::S6_RING; ++i)
            ok = hipEventCreateWithFlags(&s->done_ev[i], hipEventDisableTiming) == hipSuccess;
        if (ok) {
            std::memset(s->mirror, 0, dfa_solver6::S6_RING * dfa::S6_HIST * sizeof(int));
        } else {  // no mirror: every PCG gets its full budget
            (void)hipGetLastError();
            if (s->mirror) (void)hipHostFree(s->mirror);
            s->mirror = nullptr;
        }
    }
    if (rc == DFA_OK) {
        hipError_t e = s->grid.reserve(max_D);
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc (node grid)");
    }
    if (rc != DFA_OK) {
        dfa_solver6_destroy(s);
        return rc;
    }
    *out = s;
    return DFA_OK;
}

void dfa_solver6_destroy(dfa_solver6* s) {
    if (!s) return;
    for (auto& g : s->pcg_graphs) (void)hipGraphExecDestroy(g.second);
    if (s->mirror) (void)hipHostFree(s->mirror);
    for (hipEvent_t e : s->done_ev)
        if (e) (void)hipEventDestroy(e);
    if (s->capture_stream) (void)hipStreamDestroy(s->capture_stream);
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    for (void* p : s->blocks) (void)hipFree(p);
    if (s->xcd_perm) (void)hipFree(s->xcd_perm);
    s->grid.release();
    delete s;
}

int dfa_solver6_set_problem(dfa_solver6* s, const float* node_pos, const float* node_dq, const float* node_w, int D,
                            const float* canon_vertices, const float* canon_normals, int N, dfa_stream_t stream) {
    REQUIRE(s, "null plan");
    REQUIRE(node_pos && node_dq && node_w && D > 0, "no nodes");
    REQUIRE(N >= 0 && (N == 0 || canon_vertices), "bad vertex arrays");
    if (D > s->max_D || N > s->max_N) return fail(DFA_ERR_CAPACITY, "problem larger than the plan");
    dfa::Solve6View& v = s->v;
    v.N = N, v.D = D, v.k = s->k;
    // (the solver reads its own, sorted copies of the vertices: s6_build_graph)
    v.node_pos = node_pos, v.node_w = node_w, v.canon = v.canon_own, v.canon_n = canon_normals ? v.canon_n_own : nullptr;
    s->node_dq = node_dq;
    const dfa::KnnGridView* grid = nullptr;
    if (want_grid(D, N)) {
        HIP_TRY(dfa::knn_grid_build(s->grid.v, node_pos, D, S(stream)));
        grid = &s->grid.v;
    }
    HIP_TRY(dfa::launch_knn(node_pos, node_w, D, canon_vertices, N, s->k, v.idx_nat, s->raw_w, grid, S(stream)));
    const int kreg = s->k + 1;
    HIP_TRY(dfa::launch_knn(node_pos, node_w, D, node_pos, D, kreg, s->raw_reg, nullptr, grid, S(stream)));
    HIP_TRY(dfa::s6_build_graph(v, s->state, canon_vertices, canon_normals, s->raw_w, s->raw_reg, kreg, S(stream)));
    v.xcd_perm = nullptr;
    if (dfa::kDevAB && dfa::dev_env_int("DFA_XCD_MAP", 0) == 2) {
        // (experiment only, profiles/r06_xcd_map.md: the nodes along a Morton curve of their positions, sorted on the host —
        // a synchronisation per graph build that a product form would replace by a device sort)
        std::vector<float> pos((size_t)3 * D);
        HIP_TRY(hipMemcpyAsync(pos.data(), node_pos, sizeof(float) * pos.size(), hipMemcpyDeviceToHost, S(stream)));
        HIP_TRY(hipStreamSynchronize(S(stream)));
        float lo[3] = {pos[0], pos[1], pos[2]}, hi[3] = {pos[0], pos[1], pos[2]};
        for (int i = 0; i < D; ++i)
            for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], pos[3 * i + a]), hi[a] = std::max(hi[a], pos[3 * i + a]);
        auto spread = [](uint64_t x) {  // 21 bits -> every third bit
            x &= 0x1fffff, x = (x | x << 32) & 0x1f00000000ffffull, x = (x | x << 16) & 0x1f0000ff0000ffull;
            x = (x | x << 8) & 0x100f00f00f00f00full, x = (x | x << 4) & 0x10c30c30c30c30c3ull, x = (x | x << 2) & 0x1249249249249249ull;
            return x;
        };
        std::vector<std::pair<uint64_t, int32_t>> key((size_t)D);
        for (int i = 0; i < D; ++i) {
            uint64_t m = 0;
            for (int a = 0; a < 3; ++a) {
                const float u = (pos[3 * i + a] - lo[a]) / std::max(hi[a] - lo[a], 1e-12f);
                m |= spread((uint64_t)(u * 1048575.f)) << a;
            }
            key[i] = {m, i};
        }
        std::sort(key.begin(), key.end());
        std::vector<int32_t> perm((size_t)D);
        for (int i = 0; i < D; ++i) perm[i] = key[i].second;
        if (!s->xcd_perm) HIP_TRY(hipMalloc((void**)&s->xcd_perm, sizeof(int32_t) * (size_t)s->max_D));
        HIP_TRY(hipMemcpy(s->xcd_perm, perm.data(), sizeof(int32_t) * perm.size(), hipMemcpyHostToDevice));
        v.xcd_perm = s->xcd_perm;
    }
    s->has_problem = true;
    return DFA_OK;
}

int dfa_solver6_set_node_transforms(dfa_solver6* s, const float* node_dq) {
    REQUIRE(s && s->has_problem, "no problem set");
    REQUIRE(node_dq, "null transforms");
    s->node_dq = node_dq;
    return DFA_OK;
}

int dfa_solver6_solve(dfa_solver6* s, const float* live_vertex_map, int vertex_step, const float* live_normal_map,
                      int normal_step, int cols, int rows, float fx, float fy, float cx, float cy,
                      const dfa_solve6_params* prm, dfa_stream_t stream) {
    REQUIRE(s && s->has_problem, "no problem set");
    REQUIRE(prm, "null params");
    REQUIRE(live_vertex_map && live_normal_map && cols > 0 && rows > 0, "bad live maps");
    REQUIRE(vertex_step >= cols * 16 && normal_step >= cols * 16, "row step smaller than a row");
    REQUIRE(prm->num_iter >= 0 && prm->gn_iter >= 0 && prm->linear_iter >= 0, "negative iteration count");
    REQUIRE(prm->tukey_offset > 0.f && prm->psi_data > 0.f && prm->psi_reg > 0.f, "non-positive robust parameter");
    REQUIRE(prm->damping >= 0.f && prm->lambda >= 0.f, "negative damping / lambda");
    REQUIRE(prm->pcg_tol_first <= 0.f || prm->pcg_tol_adapt > 0.f || (prm->pcg_tol_decay > 0.f && prm->pcg_tol_decay <= 1.f),
            "forcing decay outside (0, 1]");
    REQUIRE(prm->pcg_tol_adapt >= 0.f, "negative adaptive forcing factor");
    REQUIRE(prm->gn_tol >= 0.f && prm->gn_tol < 1.f, "gn_tol outside [0, 1)");
    dfa::Solve6Params p{prm->num_iter, prm->gn_iter, prm->linear_iter, prm->tukey_offset, prm->psi_data, prm->lambda,
                        prm->psi_reg, prm->dist_thresh, prm->cos_thresh, prm->damping, prm->pcg_tol, prm->pcg_tol_first,
                        prm->pcg_tol_decay, prm->pcg_tol_adapt, prm->gn_tol};
    const bool early = p.gn_tol > 0.f;
    dfa::Solve6Image img{live_vertex_map, live_normal_map, vertex_step, normal_step, cols, rows, fx, fy, cx, cy};
    hipStream_t st = S(stream);
    s->ev_used = 0;
    HIP_TRY(dfa::s6_begin(s->v, s->state, s->node_dq, early ? p.num_iter * p.gn_iter : 0, st));
    s->last_launches = 0;
    const bool no_graph = dfa::dev_env("DFA_S6_NO_GRAPH") != nullptr;  // (development builds: launches issued one by one)
    // ---- launch budget: fold the solves up to n - 2 into the history (in order, each behind its completion event)
    const bool adaptive = prm->adaptive_launch && s->mirror;
    const unsigned long long n = s->solve_seq++;
    const int slot         = (int)(n % dfa_solver6::S6_RING);
    if (adaptive) {
        const dfa_solver6::BudgetKey key{s->v.D, s->v.N, p.num_iter, p.gn_iter, p.linear_iter, p.pcg_tol, p.pcg_tol_first,
                                         p.pcg_tol_decay, p.pcg_tol_adapt, p.gn_tol};
        const dfa_solver6::BudgetKey& old = s->budget_key;
        auto far = [](int a, int b) { return std::abs(a - b) * 8 > std::max(a, b); };  // changed by more than an eighth
        const bool reset = far(key.D, old.D) || far(key.N, old.N) || key.num_iter != old.num_iter || key.gn_iter != old.gn_iter ||
                           key.linear_iter != old.linear_iter || key.tol != old.tol || key.tol_first != old.tol_first ||
                           key.tol_decay != old.tol_decay || key.tol_adapt != old.tol_adapt || key.gn_tol != old.gn_tol;
        s->budget_key = key;
        for (; s->folded + 2 <= n; ++s->folded) {
            const int fs = (int)(s->folded % dfa_solver6::S6_RING);
            HIP_TRY(hipEventSynchronize(s->done_ev[fs]));  // two solves back: complete long ago unless the caller is far ahead
            for (int gi = 0; gi < std::min(s->slot_gn[fs], dfa::S6_HIST); ++gi) {
                const int seen = s->mirror[fs * dfa::S6_HIST + gi];
                int& pred      = s->pred[gi];
                // (an iteration behind the end of its outer iteration ran no PCG: its budget decays like one that needed
                // little — the launches enqueued for it are no-ops every time it is skipped again — but stays known)
                if (seen == dfa::S6_MIRROR_SKIPPED) pred = pred > 1 ? pred - 1 : 1;  // (never back to 0 = unknown = the full cap)
                else if (seen > 0) pred = std::max(seen, pred - 1);
                else if (seen < 0) pred = std::max(pred, -2 * seen);  // cut short: twice as many
            }
        }
        if (reset) {  // another problem (or other stopping rules): what the previous one needed says nothing
            std::memset(s->pred, 0, sizeof(s->pred));
            s->folded = n;  // (solves n - 2, n - 1 of the old problem are never folded)
        }
    } else {
        s->folded = n + 1 >= 2 ? n - 1 : 0;  // nothing to fold later from solves without a budget
        std::memset(s->pred, 0, sizeof(s->pred));
    }
    int* mirror_slot = s->mirror ? s->mirror + slot * dfa::S6_HIST : nullptr;
    if (mirror_slot) {
        // the slot's previous owner, solve n - S6_RING, must have finished writing it (it has, unless the caller runs more
        // than S6_RING - 1 solves ahead of the device)
        if (n >= (unsigned long long)dfa_solver6::S6_RING) HIP_TRY(hipEventSynchronize(s->done_ev[slot]));
        std::memset(mirror_slot, 0, dfa::S6_HIST * sizeof(int));
        s->slot_gn[slot] = p.num_iter * p.gn_iter;
    }
    // the graphs replay launches over the plan's own buffers: only the node count is part of what they captured
    if (s->pcg_key_D != s->v.D) {
        for (auto& g : s->pcg_graphs) (void)hipGraphExecDestroy(g.second);
        s->pcg_graphs.clear();
        s->pcg_key_D = s->v.D;
    }
    for (int outer = 0; outer < p.num_iter; ++outer)
        for (int gn = 0; gn < p.gn_iter; ++gn) {
            auto mark = [&]() {  // 4 events per Gauss-Newton iteration: | linearise | assemble | pcg |
                if (!s->timing) return;
                if (s->ev_used == s->events.size()) {
                    hipEvent_t e;
                    if (hipEventCreate(&e) != hipSuccess) return;
                    s->events.push_back(e);
                }
                (void)hipEventRecord(s->events[s->ev_used++], st);
            };
            const int gi = outer * p.gn_iter + gn;
            mark();
            // gn_tol > 0: the launch's last workgroup applies the stopping rule on the device — launches behind the end of an
            // outer iteration return at entry (nothing comes back to the host: the launches of the whole solve are enqueued
            // regardless)
            HIP_TRY(dfa::s6_linearise(s->v, s->state, img, p, gn == 0, gi, gn, 0, st));
            mark();
            HIP_TRY(dfa::s6_assemble(s->v, s->state, p, gn, st));
            mark();
            // Launches of this PCG: the caller's cap, or (adaptive_launch) what this Gauss-Newton iteration needed in the
            // plan's earlier solves (a maximum that decays by one per solve) plus a quarter, at least two.  (Measured at C2 /
            // C3 over 30-frame sequences: consecutive frames move a count by up to 2 where it is small and by up to a
            // quarter where it is 30-40; one launch of slack instead of two cut 1-2 PCGs short in a fifth of the frames.)
            int launches = p.linear_iter;
            // (VERDICT r05 item 7 — one launch of slack where a slot's count had repeated in two solves running, one launch for a
            // slot skipped twice running — was built and measured in round 6: 58 launches for 41 iterations instead of 60 for 36
            // at C2, 83 for 53 instead of 92 for 47 at C3, and the frames/s inside the run-to-run spread (863.6 against 860,
            // 331 against 343 over the whole period): a launch that returns at entry costs 3.5 us, a dozen of them 3 % of a
            // C2 frame.  Not kept.)
            if (adaptive && gi < dfa::S6_HIST && s->pred[gi] > 0)
                launches = std::min(p.linear_iter, s->pred[gi] + std::max(2, s->pred[gi] / 4));
            // the PCG launches of one Gauss-Newton iteration are replayed as a HIP graph (one per launch count); if capture
            // is not possible here (it never is on some stream configurations) the launches are issued one by one
            bool replayed = false;
            if (!s->graph_disabled && !no_graph) {
                auto it = s->pcg_graphs.find(launches);
                if (it == s->pcg_graphs.end()) {
                    hipGraph_t g = nullptr;
                    hipGraphExec_t ge = nullptr;
                    bool ok = s->capture_stream || hipStreamCreateWithFlags(&s->capture_stream, hipStreamNonBlocking) == hipSuccess;
                    ok = ok && hipStreamBeginCapture(s->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                    if (ok) {
                        const hipError_t le = dfa::s6_pcg_n(s->v, s->state, launches, s->capture_stream);
                        const hipError_t ce = hipStreamEndCapture(s->capture_stream, &g);
                        ok = le == hipSuccess && ce == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess;
                        if (g) (void)hipGraphDestroy(g);
                    }
                    if (ok) {
                        it = s->pcg_graphs.emplace(launches, ge).first;
                    } else {
                        (void)hipGetLastError();  // clear the sticky error of the failed attempt
                        s->graph_disabled = true;
                    }
                }
                if (it != s->pcg_graphs.end()) {
                    HIP_TRY(hipGraphLaunch(it->second, st));
                    replayed = true;
                }
            }
            if (!replayed) HIP_TRY(dfa::s6_pcg_n(s->v, s->state, launches, st));
            s->last_launches += launches + 1;
            mark();
            HIP_TRY(dfa::s6_update(s->v, s->state, launches, p.linear_iter, gi < dfa::S6_HIST ? mirror_slot : nullptr, gi, 1, st));
        }
    if (early && p.num_iter * p.gn_iter > 0) {
        // the last step of the solve, if its outer iteration ran to the cap: one closing linearisation decides whether it stays
        const int gi = p.num_iter * p.gn_iter;
        HIP_TRY(dfa::s6_linearise(s->v, s->state, img, p, 0, gi, p.gn_iter, 1, st));
        HIP_TRY(dfa::s6_update(s->v, s->state, 0, p.linear_iter, nullptr, gi, 0, st));
    }
    if (mirror_slot) HIP_TRY(hipEventRecord(s->done_ev[slot], st));
    return DFA_OK;
}

int dfa_solver6_enable_timing(dfa_solver6* s, int enable) {
    REQUIRE(s, "null plan");
    s->timing = enable != 0;
    return DFA_OK;
}

int dfa_solver6_get_timing(dfa_solver6* s, dfa_solve6_timing* out, dfa_stream_t stream) {
    REQUIRE(s && out, "null plan / out");
    HIP_TRY(hipStreamSynchronize(S(stream)));
    std::memset(out, 0, sizeof(*out));
    for (size_t i = 0; i + 3 < s->ev_used; i += 4) {
        float a = 0.f, b = 0.f, c = 0.f;
        HIP_TRY(hipEventElapsedTime(&a, s->events[i], s->events[i + 1]));
        HIP_TRY(hipEventElapsedTime(&b, s->events[i + 1], s->events[i + 2]));
        HIP_TRY(hipEventElapsedTime(&c, s->events[i + 2], s->events[i + 3]));
        out->linearise_ms += a, out->assemble_ms += b, out->pcg_ms += c;
        out->gn_iterations += 1;
    }
    if (s->has_problem && s->v.D > 0) {
        std::vector<int32_t> cnt((size_t)s->v.D);
        HIP_TRY(hipMemcpy(cnt.data(), s->v.bcnt, sizeof(int32_t) * cnt.size(), hipMemcpyDeviceToHost));
        for (int32_t c : cnt) out->matrix_blocks += c;
    }
    return DFA_OK;
}

const float* dfa_solver6_node_dq(const dfa_solver6* s) { return s ? s->v.dq : nullptr; }

int dfa_solver6_warp(dfa_solver6* s, float* out_vertices, float* out_normals, dfa_stream_t stream) {
    REQUIRE(s && s->has_problem, "no problem set");
    REQUIRE(out_vertices, "null output");
    HIP_TRY(dfa::s6_warp(s->v, s->v.dq, out_vertices, out_normals, S(stream)));
    return DFA_OK;
}

int dfa_solver6_get_stats(dfa_solver6* s, dfa_solve6_stats* out, dfa_stream_t stream) {
    REQUIRE(s && out, "null argument");
    dfa::Solve6State h;
    HIP_TRY(hipMemcpyAsync(&h, s->state, sizeof(h), hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    out->initial_cost = h.initial_cost, out->final_cost = h.final_cost;
    out->gn_iters = h.gn_iters, out->pcg_iters = h.pcg_iters;
    out->gn_solves = h.gn_solves, out->gn_rejected = h.gn_rejected, out->gn_converged = h.gn_converged, out->hist_n = h.hist_n;
    out->valid_first = (long long)h.valid_first, out->valid_last = (long long)h.valid_last;
    out->max_row_blocks = h.max_row_blocks, out->overflow = h.overflow;
    out->pcg_short = h.pcg_short, out->pcg_launches = s->last_launches;
    static_assert(DFA_SOLVE6_HIST == dfa::S6_HIST, "history length of the C ABI and of the state block");
    for (int i = 0; i < DFA_SOLVE6_HIST; ++i) {
        const bool in = i < h.hist_n;
        out->valid_hist[i] = in ? (long long)h.valid_hist[i] : 0ll;
        out->stop_hist[i] = in ? h.stop_hist[i] : 0;
        out->cost_hist[i] = in ? h.cost_hist[i] : 0.0;
        out->pcg_rel_hist[i] = in ? h.pcg_rel_hist[i] : 0.f;
        out->pcg_tol_hist[i] = in ? h.pcg_tol_hist[i] : 0.f;
        out->pcg_it_hist[i] = in ? h.pcg_it_hist[i] : 0;
    }
    return h.overflow ? fail(DFA_ERR_CAPACITY, "a block row of the normal matrix exceeded the plan's capacity") : DFA_OK;
}

}  // extern "C"

#ifdef DFA_S6_DEBUG  // development builds only: device pointers of the north-star plan (tools/_dbg*.py)
extern "C" __attribute__((visibility("default"))) int dfa_dev_solver6_ptrs(dfa_solver6* s, void** out) {
    const dfa::Solve6View& v = s->v;
    void* p[] = {v.utab, v.pair_ptr, v.pair_list, v.bcnt, v.bfu, (void*)v.node_ptr, (void*)v.node_list, v.bcols, v.bvals, v.g,
                 (void*)v.idx, v.rec, nullptr, v.mnode, v.rslot};
    out[18] = v.minv;
    for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); ++i) out[i] = p[i];
    out[15] = (void*)(size_t)v.cap, out[16] = (void*)(size_t)v.D, out[17] = (void*)(size_t)v.N;
    return 0;
}
#endif
