// solve.hip — gfx950 kernels of the warp-field solve: residual rows, Tukey / Huber
// re-weighting, assembly of the sparse normal equations, block-Jacobi PCG, write-back.
//
// Replaces what the reference runs through Opt (include/dynfu/utils/terra/energy.t, driven by
// src/dynfu/utils/opt_solver.cpp) plus the CPU loops updateTukeyBiweights / updateHuberWeights.
//
// Formulation.  Unknown t_i in R^3 per node.  Every residual of
// energy.t is a "row"  r = sqrt(tau) * (b - sum_j w_j t_{n_j})  with at most k node slots:
//   data row v   (energy.t:50-55): slots = k-NN of the canonical vertex, w = RBF weights,
//                                  b = live - canonical, tau = Tukey biweight;
//   reg row (n,i)(energy.t:75-78): slots = {v_i: -1, n: +1}, b = 0, tau = w_reg^2
//                                  (r = w_reg (t_{v_i} - t_n)); the self edge is empty.
// J^T J has the block structure D x D with blocks s*I_3, so ONE scalar sparse matrix A
// (D x D) serves the x, y and z systems:  A = sum_rows tau w w^T,  g = sum_rows tau w e,
// e = b - sum w t.
//
// MI355X mapping
//   * Opt re-walks all N*k graph edges with global atomics in every PCG iteration; here A is
//     assembled once per linearisation and the PCG iterates on ~20 D non-zeros that never
//     leave the chip's caches.
//   * assembly is a gather, not a scatter: a node -> rows transpose graph is built once per
//     frame, then ONE WAVE PER NODE reduces its ~128 k rows into an LDS hash keyed by column
//     (LDS float atomics stay on the CU; no global atomics on the matrix at all) and writes
//     one ELL row + one rhs entry + the Jacobi diagonal.
//   * the PCG is a single persistent 1024-thread workgroup: at 2 k - 8 k nodes the solve is
//     bound by synchronisation latency, not bandwidth — a workgroup barrier
//     costs ~0.1 us where a grid-wide barrier costs 4-5 us — with the direction vector in
//     LDS (16 B / node, ds_read_b128 gathers), x / r / p of the thread's own rows in
//     registers and dot products reduced by wave shuffles in double.
#include <hip/hip_runtime.h>
#include <cstring>

#include <algorithm>
#include <map>
#include <mutex>

#include <float.h>

#include "dq_device.hpp"
#include "dev_switch.hpp"
#include "kernels.hpp"
#include "solve.hpp"

namespace dfa {

// ------------------------------------------------------------------------------------------
// transpose graph node -> (row, slot).  A counting sort of the R*k slot entries by node id with
// workgroup-private histograms in LDS: TG_BLOCKS workgroups each own a contiguous chunk of the
// entries, count into LDS (no global atomics: 1 M device-scope atomics on 2 k counters were
// the whole cost of the first version), publish their histogram, a thread per node turns the
// [block][node] table into per-block bases + node totals, and the fill pass (which scans the totals
// itself) replays the chunk with an LDS cursor per node.  Deterministic up to the order inside one chunk.

constexpr int TG_BLOCKS = SOLVE_TG_BLOCKS;

__global__ __launch_bounds__(1024) void tg_count_kernel(const int32_t* __restrict__ ridx, size_t total, int D,
                                                        int32_t* __restrict__ blk_hist /* [TG_BLOCKS][D] */) {
    extern __shared__ int32_t hist[];
    for (int i = threadIdx.x; i < D; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const size_t chunk = (total + TG_BLOCKS - 1) / TG_BLOCKS;
    const size_t beg = chunk * blockIdx.x, end = min(beg + chunk, total);
    for (size_t e = beg + threadIdx.x; e < end; e += blockDim.x) {
        const int n = ridx[e];
        if (n >= 0) atomicAdd(&hist[n], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += blockDim.x) blk_hist[(size_t)blockIdx.x * D + i] = hist[i];
}

// per node (one thread each): exclusive prefix over the TG_BLOCKS workgroup counts -> per-block bases RELATIVE to the
// node's segment, and the node's total in row TG_BLOCKS of the table.  (Round 1 did this and the scan of the totals in
// one 1024-thread workgroup: two passes of 64 dependent loads per thread, 19 us at C2; now 64 independent loads.)
__global__ __launch_bounds__(256) void tg_colscan_kernel(int32_t* __restrict__ blk_hist /* [TG_BLOCKS + 1][D] */, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    constexpr int G = 32;  // counts in flight per thread (all TG_BLOCKS at once would be 256 registers)
    static_assert(TG_BLOCKS % G == 0, "workgroups of the counting sort in groups");
    int run = 0;
    for (int b0 = 0; b0 < TG_BLOCKS; b0 += G) {
        int h[G];
#pragma unroll
        for (int b = 0; b < G; ++b) h[b] = blk_hist[(size_t)(b0 + b) * D + i];
#pragma unroll
        for (int b = 0; b < G; ++b) {
            blk_hist[(size_t)(b0 + b) * D + i] = run;
            run += h[b];
        }
    }
    blk_hist[(size_t)TG_BLOCKS * D + i] = run;
}

// Every fill workgroup scans the D node totals itself (LDS, a few microseconds) instead of waiting for a scan launch;
// workgroup 0 publishes node_ptr.
__global__ __launch_bounds__(1024) void tg_fill_kernel(const int32_t* __restrict__ ridx, size_t total, int D,
                                                       const int32_t* __restrict__ blk_base /* [TG_BLOCKS + 1][D] */,
                                                       int32_t* __restrict__ node_ptr, uint32_t* __restrict__ node_list) {
    extern __shared__ int32_t cursor[];
    __shared__ int32_t wave_tot[16];
    __shared__ int32_t carry_sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    for (int base = 0; base < D; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < D ? blk_base[(size_t)TG_BLOCKS * D + i] : 0;
        const int incl = wave_inclusive_scan(v);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = carry_sh + incl - v;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (i < D) {
            cursor[i] = off + blk_base[(size_t)blockIdx.x * D + i];
            if (blockIdx.x == 0) node_ptr[i] = off;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_sh = off + v;
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) node_ptr[D] = carry_sh;
    const size_t chunk = (total + TG_BLOCKS - 1) / TG_BLOCKS;
    const size_t beg = chunk * blockIdx.x, end = min(beg + chunk, total);
    for (size_t e = beg + threadIdx.x; e < end; e += blockDim.x) {
        const int n = ridx[e];
        if (n >= 0) node_list[atomicAdd(&cursor[n], 1)] = (uint32_t)e;
    }
}

// ------------------------------------------------------------------------------------------
// linearisation, one launch per Gauss-Newton iteration:
//   [outer-iteration start only] robust weights: Tukey biweight of the current warp error for
//       the data rows (opt_solver.cpp:204-231; warp(c) = calcDQB(c)(c) with node transforms
//       DQ(t_i) * dg_se3_i, :270-285 + node.cpp:19-23), w_reg^2 for the regularisation rows (:30);
//   residual  e_r = b_r - sum_j w_rj t_{n_rj}  -> tail (e, tau) of the row's packed record;
//   cost      sum tau |e|^2: one partial per workgroup, the LAST workgroup to arrive (agent-scope
//       release/acquire around a ticket counter) adds the partials in index order (deterministic)
//       and runs the Gauss-Newton control logic — no separate control launch.

// a row's k node ids and weights: 16-byte loads when k is the template's K (the common case), k dwords else; absent: -1 / 0
template <int K>
__device__ __forceinline__ void load_row_graph(const SolveView& s, size_t r, int (&n)[K], float (&w)[K]) {
    if (s.k == K) {  // (uniform)
#pragma unroll
        for (int q = 0; q < K / 4; ++q) {
            const int4 iv   = reinterpret_cast<const int4*>(s.ridx + r * K)[q];
            const float4 wv = reinterpret_cast<const float4*>(s.rw + r * K)[q];
            n[4 * q] = iv.x, n[4 * q + 1] = iv.y, n[4 * q + 2] = iv.z, n[4 * q + 3] = iv.w;
            w[4 * q] = wv.x, w[4 * q + 1] = wv.y, w[4 * q + 2] = wv.z, w[4 * q + 3] = wv.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) n[j] = j < s.k ? s.ridx[r * s.k + j] : -1, w[j] = j < s.k ? s.rw[r * s.k + j] : 0.f;
    }
}

template <int K>
__device__ __forceinline__ float tukey_weight(const SolveView& s, size_t v, float tukey_offset, float psi_data) {
    const f3 c = mk3(s.canon[3 * v], s.canon[3 * v + 1], s.canon[3 * v + 2]);
    DQ sum     = dq_identity();
    int n[K];
    float w[K];
    load_row_graph<K>(s, v, n, w);
    // the neighbours' translations and transforms four at a time, by unconditional loads (an absent neighbour reads node 0
    // and is skipped): loads inside the `if` were k dependent round trips
#pragma unroll
    for (int h = 0; h < K; h += 4) {
        float tx[4], ty[4], tz[4];
        DQ q[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int nn = n[h + jj] >= 0 ? n[h + jj] : 0;
            tx[jj] = s.t[3 * nn], ty[jj] = s.t[3 * nn + 1], tz[jj] = s.t[3 * nn + 2];
            q[jj]  = dq_load(s.node_dq + 8 * (size_t)nn);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (n[h + jj] >= 0) {
                const DQ cur = dq_mul(dq_from_translation(tx[jj], ty[jj], tz[jj]), q[jj]);
                sum          = dq_mul(sum, dq_scale(cur, w[h + jj]));
            }
        }
    }
    const f3 warped = dq_transform(dq_normalize(sum), c);
    const float ex = s.live[3 * v] - warped.x, ey = s.live[3 * v + 1] - warped.y, ez = s.live[3 * v + 2] - warped.z;
    // calcTukeyBiweight (:204-212)
    const float d = sqrtf(ex * ex + ey * ey + ez * ez) / tukey_offset;
    if (d < psi_data) {
        const double q = 1.0 - ((double)d * (double)d) / ((double)psi_data * (double)psi_data);
        return (float)(q * q);
    }
    return 0.f;
}

// Huber weights (opt_solver.cpp:233-268): computed for interface parity, energy.t:70 never uses them
__device__ __forceinline__ void huber_node(const SolveView& s, int i, float psi_reg) {
    const DQ dq_i = dq_mul(dq_from_translation(s.t[3 * i], s.t[3 * i + 1], s.t[3 * i + 2]),
                           dq_load(s.node_dq + 8 * (size_t)i));
    float h = 0.f;
    for (int j = 0; j < s.k; ++j) {
        const int m = s.reg_idx[(size_t)i * s.k + j];
        if (m < 0) break;
        const f3 pm   = mk3(s.node_pos[3 * m], s.node_pos[3 * m + 1], s.node_pos[3 * m + 2]);
        const DQ dq_m = dq_mul(dq_from_translation(s.t[3 * m], s.t[3 * m + 1], s.t[3 * m + 2]),
                               dq_load(s.node_dq + 8 * (size_t)m));
        const f3 pa = dq_transform(dq_i, pm), pb = dq_transform(dq_m, pm);
        const float ex = pa.x - pb.x, ey = pa.y - pb.y, ez = pa.z - pb.z;
        const float err = sqrtf(ex * ex + ey * ey + ez * ez);
        h               = fabsf(err) <= psi_reg ? 1.f : psi_reg / fabsf(err);  // last neighbour wins (:263)
    }
    s.huber[i] = h;
}
__global__ __launch_bounds__(256) void huber_kernel(SolveView s, float psi_reg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < s.D) huber_node(s, i, psi_reg);
}

struct LineariseArgs {
    int update_weights;  // first linearisation of an outer iteration
    int mode;            // 0 first of outer, 1 later GN iteration, 2 final cost only
    float gn_tol, tukey_offset, psi_data, w_reg_sq;
    long long* iters_total;  // mode 2, optional (device): += the solve's PCG iterations
    float huber_psi;  // > 0: also evaluate the nodes' Huber weights at the current t (the last outer iteration's
                      // preNonlinearSolve, opt_solver.cpp:135-140; a launch of its own before)
};

constexpr int LIN_SHARDS     = 32;    // ticket counters: one device-scope atomic costs ~11 ns when
constexpr int LIN_MAX_BLOCKS = 1024;  // serialised on one word, so arrivals are sharded 2-level

template <int K>
__global__ __launch_bounds__(256) void linearise_kernel(SolveView s, SolveState* __restrict__ st,
                                                        double* __restrict__ cost_partials,
                                                        unsigned int* __restrict__ ticket /*[LIN_SHARDS+1]*/,
                                                        LineariseArgs a) {
    __shared__ double wsum[4];
    __shared__ int is_last;
    if (a.huber_psi > 0.f)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s.D; i += gridDim.x * blockDim.x) huber_node(s, i, a.huber_psi);
    if (a.mode == 2) {
        // the closing evaluation also composes the result: dg_se3_i <- DQ(0,0,0,t_i) * dg_se3_i (opt_solver.cpp:270-285,
        // node.cpp:19-23) — t is final here; a launch of its own before
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s.D; i += gridDim.x * blockDim.x) {
            const DQ out = dq_mul(dq_from_translation(s.t[3 * i], s.t[3 * i + 1], s.t[3 * i + 2]),
                                  dq_load(s.node_dq + 8 * (size_t)i));
            dq_store(s.node_dq_out + 8 * (size_t)i, out);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.iters_total) *a.iters_total += st->pcg_iters;
    }
    // after convergence t no longer changes: weights, residual records and cost of this linearisation exist already.
    // That includes the solve's closing evaluation (mode 2) when an iteration ran: the flag is set by a PCG that found
    // its gradient at the floor and left t where the linearisation before it had evaluated the cost.
    if (st->converged == 1 || (st->converged && !a.update_weights)) {
        if (a.mode != 2 || (st->have_initial && !st->cost_stale)) return;
    }
    const size_t R = (size_t)s.N + (size_t)s.D * s.k;
    double c       = 0.0;
    float amax     = 0.f;  // re-weighting linearisations: the largest addend tau w_a w_b any row brings to the normal matrix
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (size_t)gridDim.x * blockDim.x) {
        float tau;
        if (a.update_weights) {
            tau       = r < (size_t)s.N ? tukey_weight<K>(s, r, a.tukey_offset, a.psi_data) : a.w_reg_sq;
            s.rtau[r] = tau;
        } else {
            tau = s.rtau[r];
        }
        float sx = 0.f, sy = 0.f, sz = 0.f;
        {   // (ids and weights by 16-byte loads, then the k translations by unconditional loads — an absent neighbour reads
            // node 0 and is skipped —: two rounds of loads; with the loads inside `if (n >= 0)` it was k dependent ones)
            int n[K];
            float w[K], tx[K], ty[K], tz[K];
            load_row_graph<K>(s, r, n, w);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int nn = n[j] >= 0 ? n[j] : 0;
                tx[j] = s.t[3 * nn], ty[j] = s.t[3 * nn + 1], tz[j] = s.t[3 * nn + 2];
            }
            float wm = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (n[j] >= 0) sx += w[j] * tx[j], sy += w[j] * ty[j], sz += w[j] * tz[j], wm = fmaxf(wm, fabsf(w[j]));
            amax = fmaxf(amax, tau * wm * wm);
        }
        const float ex = s.rb[3 * r] - sx, ey = s.rb[3 * r + 1] - sy, ez = s.rb[3 * r + 2] - sz;
        // tail of the packed row record (head = k ids + k weights, written once per frame)
        *(float4*)(s.re + r * (size_t)solve_rec_words(s.k) + solve_rec_tail(s.k)) = make_float4(ex, ey, ez, tau);
        c += (double)tau * ((double)ex * ex + (double)ey * ey + (double)ez * ez);
    }
    c = wave_sum_all(c);
    __shared__ float wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c, wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        // publish the partial write-through (sc1) — no release fence, which would write back the
        // whole L2's dirty record tails — then arrive: shard counter first, top counter for the
        // last arriver of each shard
        __hip_atomic_store(&cost_partials[blockIdx.x], (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        if (a.update_weights)  // (the second half of the array: one maximum per workgroup)
            __hip_atomic_store(&cost_partials[LIN_MAX_BLOCKS + blockIdx.x], (double)fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int shard   = blockIdx.x % LIN_SHARDS;
        const unsigned int members = (gridDim.x - shard + LIN_SHARDS - 1) / LIN_SHARDS;
        const unsigned int nshards = min((unsigned int)LIN_SHARDS, gridDim.x);
        int last                   = 0;
        if (__hip_atomic_fetch_add(&ticket[shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(&ticket[shard], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
            if (__hip_atomic_fetch_add(&ticket[LIN_SHARDS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                nshards - 1) {
                __hip_atomic_store(&ticket[LIN_SHARDS], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;

    // last workgroup: ordered sum of the partials + Gauss-Newton control
    __shared__ double sm[256];
    __shared__ double smx[256];
    double acc = 0.0, mx = 0.0;
    {
        // these loads go past the L2 (~2 us each) and this is the one workgroup the launch — and the assembly behind it — waits
        // for: all of a thread's partials in flight together (LIN_MAX_BLOCKS / 256 = 4 per array; a load-wait-add loop was four
        // dependent round trips), clamped addresses, masked sums in the same order as before
        constexpr int Q = LIN_MAX_BLOCKS / 256;
        double cq[Q], mq[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const unsigned int i = min(threadIdx.x + 256u * q, gridDim.x - 1);
            cq[q] = __hip_atomic_load(&cost_partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mq[q] = a.update_weights ? __hip_atomic_load(&cost_partials[LIN_MAX_BLOCKS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (threadIdx.x + 256u * q < gridDim.x) acc += cq[q], mx = fmax(mx, mq[q]);
    }
    sm[threadIdx.x] = acc, smx[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o], smx[threadIdx.x] = fmax(smx[threadIdx.x], smx[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double cost = sm[0];
        if (a.update_weights) st->amax = (float)smx[0];
        if (!st->have_initial) st->initial_cost = cost, st->have_initial = 1;
        if (a.mode == 0) st->done = 0;
        // Gauss-Newton early-out: relative cost decrease of the previous step below gn_tol
        if (a.mode == 1 && !st->done && a.gn_tol > 0.f && (st->cost - cost) <= (double)a.gn_tol * st->cost)
            st->done = 1;
        st->cost       = cost;
        st->final_cost = cost;
        st->cost_stale = 0;
        // robust weights evaluated at THIS t: a gradient at the floor now means the whole solve has converged (with
        // stale weights it only ends the current outer iteration: the next one re-weights at the moved t)
        if (a.mode != 2) st->weights_fresh = a.update_weights;
        if (a.mode != 2 && a.update_weights && st->converged == 2) st->converged = 0;  // a new outer iteration
    }
}

// start of a solve: unknowns, state block and arrival tickets zeroed by ONE launch
__global__ __launch_bounds__(256) void reset_kernel(float* __restrict__ t, int n3, SolveState* __restrict__ st,
                                                    unsigned int* __restrict__ ticket, int nticket) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) t[i] = 0.f;
    if (blockIdx.x == 0) {
        unsigned int* w = (unsigned int*)st;
        for (int j = threadIdx.x; j < (int)(sizeof(SolveState) / 4); j += blockDim.x) w[j] = 0u;
        for (int j = threadIdx.x; j < nticket; j += blockDim.x) ticket[j] = 0u;
    }
}

// The per-problem row set-up as ONE launch, a thread per row: regularisation rows (opt_solver.cpp:74-105), right-hand
// sides of the data rows (energy.t:55), packed record heads and the zeroing of the
// unknowns / state block / tickets (reset_kernel) — four launches of 5-16 us each at C2 in round 1.
template <int K>
__global__ __launch_bounds__(256) void prepare_rows_kernel(SolveView s, SolveState* __restrict__ st,
                                                           unsigned int* __restrict__ ticket, int nticket) {
    const size_t R = (size_t)s.N + (size_t)s.D * s.k;
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < (size_t)3 * s.D) s.t[r] = 0.f;
    if (blockIdx.x == 0) {
        unsigned int* w = (unsigned int*)st;
        for (int j = threadIdx.x; j < (int)(sizeof(SolveState) / 4); j += blockDim.x) w[j] = 0u;
        for (int j = threadIdx.x; j < nticket; j += blockDim.x) ticket[j] = 0u;
    }
    if (r >= R) return;
    const int k = s.k;
    int ids[K];
    float ws[K];
    const bool wide = k == K && K % 4 == 0;  // (uniform) the common case: every row's ids / weights / record head by 16-byte accesses
    if (r < (size_t)s.N) {  // data row: k-NN + RBF weights already in ridx / rw; b = live - canonical (energy.t:55)
#pragma unroll
        for (int c = 0; c < 3; ++c) s.rb[3 * r + c] = s.live[3 * r + c] - s.canon[3 * r + c];
        if (wide) load_row_graph<K>(s, r, ids, ws);
        else {
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < k) ids[j] = s.ridx[r * k + j], ws[j] = s.rw[r * k + j];
        }
    } else {  // regularisation row N + n k + i <- {reg_idx[n][i]: -1, n: +1} (opt_solver.cpp:74-105, energy.t:75-78)
        const int e = (int)(r - (size_t)s.N), n = e / k, m = s.reg_idx[e];
#pragma unroll
        for (int j = 0; j < K; ++j) ids[j] = -1, ws[j] = 0.f;
        if (m >= 0 && m != n) ids[0] = m, ws[0] = -1.f, ids[1] = n, ws[1] = +1.f;  // k >= 2 whenever a non-self neighbour exists
        if (wide) {
#pragma unroll
            for (int q = 0; q < K / 4; ++q) {
                reinterpret_cast<int4*>(s.ridx + r * K)[q]  = make_int4(ids[4 * q], ids[4 * q + 1], ids[4 * q + 2], ids[4 * q + 3]);
                reinterpret_cast<float4*>(s.rw + r * K)[q] = make_float4(ws[4 * q], ws[4 * q + 1], ws[4 * q + 2], ws[4 * q + 3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < k) s.ridx[r * k + j] = ids[j], s.rw[r * k + j] = ws[j];
        }
        s.rb[3 * r + 0] = s.rb[3 * r + 1] = s.rb[3 * r + 2] = 0.f;
    }
    float* rec = s.re + r * (size_t)solve_rec_words(k);
    if (wide) {  // the record's head (the words before (e, tau)) as float4 stores: a lane's record is 48 or 64 contiguous bytes
        float4* rec4 = reinterpret_cast<float4*>(rec);
        if (solve_rec_ids16(k)) {  // K / 2 words of 16-bit id pairs, then K weights
            uint32_t pk[K / 2];
#pragma unroll
            for (int j = 0; j < K / 2; ++j) {
                const uint32_t lo = ids[2 * j] < 0 ? 0xffffu : (uint32_t)ids[2 * j], hi = ids[2 * j + 1] < 0 ? 0xffffu : (uint32_t)ids[2 * j + 1];
                pk[j]             = (lo & 0xffffu) | (hi << 16);
            }
#pragma unroll
            for (int q = 0; q < K / 8; ++q)
                rec4[q] = make_float4(__uint_as_float(pk[4 * q]), __uint_as_float(pk[4 * q + 1]), __uint_as_float(pk[4 * q + 2]), __uint_as_float(pk[4 * q + 3]));
#pragma unroll
            for (int q = 0; q < K / 4; ++q) rec4[K / 8 + q] = make_float4(ws[4 * q], ws[4 * q + 1], ws[4 * q + 2], ws[4 * q + 3]);
        } else {  // K ids, then K weights
#pragma unroll
            for (int q = 0; q < K / 4; ++q)
                rec4[q] = make_float4(__int_as_float(ids[4 * q]), __int_as_float(ids[4 * q + 1]), __int_as_float(ids[4 * q + 2]), __int_as_float(ids[4 * q + 3]));
#pragma unroll
            for (int q = 0; q < K / 4; ++q) rec4[K / 4 + q] = make_float4(ws[4 * q], ws[4 * q + 1], ws[4 * q + 2], ws[4 * q + 3]);
        }
        return;
    }
    if (solve_rec_ids16(k)) {  // k / 2 words of 16-bit ids (0xffff = empty slot), then k weights
        uint16_t* h = reinterpret_cast<uint16_t*>(rec);
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < k) h[j] = ids[j] < 0 ? (uint16_t)0xffffu : (uint16_t)ids[j], rec[k / 2 + j] = ws[j];
        return;
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < k) rec[j] = __int_as_float(ids[j]), rec[k + j] = ws[j];
}

// ------------------------------------------------------------------------------------------
// assembly: one 256-thread workgroup per node.  Its ~128 k rows (transpose graph) are spread
// over the 4 waves, every row contributes k (column, tau w_a w_b) pairs to an LDS hash keyed by
// column; wave 0 then compacts the hash into the node's ELL row.  Each row is one packed record
// (k ids, k weights, e, tau) read with 16-byte loads from a single cache line.

constexpr int HASH      = 512;
constexpr int HASH_MASK = HASH - 1;

// The hash's sums are 64-bit FIXED-POINT integers, not floats.  On gfx950 an LDS float add (ds_add_f32) executes one lane
// after the other whatever the addresses — 192 cycles per wave instruction against 8 for ds_add_u64 without conflicts
// (tools/microbench_lds_atomic.hip) — and 7 of them per row were 56 % of this kernel's time (C4: 345 -> 152 us with the adds
// taken out).  An addend tau w_a w_b is a float whose magnitude is at most `amax` = the largest tau (max_j |w_j|)^2 of any
// row under the current robust weights (data rows: RBF and Tukey weights <= 1; regularisation rows: +-1 x w_reg^2) — found
// by the linearisation that evaluates those weights (SolveState::amax), so the grid follows the PROBLEM's scale: weights
// that are all tiny (sparse nodes, a narrow dg_w) or a huge lambda cost no bits.  Scaled by 2^40 / (amax rounded up to a
// power of two) an addend is an exact integer unless it is below 2^-17 of that bound (then it is cut to the grid: 2^-41 of
// the bound per addend).  The sum of up to 2^22 rows fits 63 bits, is EXACT otherwise, and does not depend on the order of
// the adds.
struct FixedScale {
    float up;     // float -> fixed: a power of two
    double down;  // fixed -> float
};
__device__ __forceinline__ FixedScale solve_fixed_scale(float amax) {
    int e = 0;
    (void)frexpf(fmaxf(amax, 1e-30f), &e);  // amax < 2^e
    e = e < -80 ? -80 : e > 100 ? 100 : e;
    FixedScale f;
    f.up   = ldexpf(1.f, 40 - e);
    f.down = ldexp(1.0, e - 40);
    return f;
}
// a node with more than 2^22 rows (a plan of millions of vertices on a handful of nodes) gives up one bit of the grid per
// doubling of its list instead of overflowing
__device__ __forceinline__ FixedScale fixed_scale_for_rows(FixedScale f, int rows) {
    const int extra = 32 - __clz((unsigned)max(rows, 1) >> 22);  // 0 up to 2^22 - 1 rows
    if (extra > 0) f.up = ldexpf(f.up, -extra), f.down = ldexp(f.down, extra);
    return f;
}
// The magnitude |v| up goes to `cell` (v >= 0: every data row) or to the cell HASH entries further on (v < 0: the
// off-diagonal entries of regularisation rows); the sum is their difference.  Converting a NON-NEGATIVE integer-valued
// float x to 64 bits takes 7 instructions — hi = floor(x / 2^32) and lo = x - hi 2^32 in [0, 2^32) are exact (a power-of-two
// scaling; x with its high bits removed has no more significant bits than x), and the integer is the register pair
// {lo, hi} — where the compiler's signed conversion takes 13 (absolute value, two floors, a sign fix-up with carries).
__device__ __forceinline__ void fixed_add(long long* cell, float v, float up) {
    const float x  = truncf(fabsf(v) * up);
    const float hf = floorf(x * 2.3283064365386963e-10f);  // 2^-32
    const uint32_t hi = (uint32_t)hf, lo = (uint32_t)fmaf(hf, -4294967296.f, x);
    atomicAdd(reinterpret_cast<unsigned long long*>(v < 0.f ? cell + HASH : cell), ((unsigned long long)hi << 32) | lo);
}

// one row record = solve_rec_words(k) consecutive words (head: prepare_rows_kernel, tail: linearise_kernel)
template <int K>
__device__ __forceinline__ float4 load_record(const SolveView& s, size_t r, int (&idx)[K], float (&w)[K]) {
    const float* rec = s.re + r * (size_t)solve_rec_words(s.k);
    if (s.k == K && (K % 8) == 0) {  // 16-bit ids: K / 8 + K / 4 + 1 aligned 16-byte loads (K = 8: one cache line)
        const float4* v = (const float4*)rec;
#pragma unroll
        for (int q = 0; q < K / 8; ++q) {
            const float4 i4 = v[q];
            const uint32_t u[4] = {__float_as_uint(i4.x), __float_as_uint(i4.y), __float_as_uint(i4.z), __float_as_uint(i4.w)};
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int lo = (int)(u[h] & 0xffffu), hi = (int)(u[h] >> 16);
                idx[8 * q + 2 * h]     = lo == 0xffff ? -1 : lo;
                idx[8 * q + 2 * h + 1] = hi == 0xffff ? -1 : hi;
            }
        }
#pragma unroll
        for (int q = 0; q < K / 4; ++q) {
            const float4 w4 = v[K / 8 + q];
            w[4 * q] = w4.x, w[4 * q + 1] = w4.y, w[4 * q + 2] = w4.z, w[4 * q + 3] = w4.w;
        }
        return v[K / 8 + K / 4];
    }
    if (solve_rec_ids16(s.k)) {  // (k a multiple of 8 below the kernel's K)
        const uint16_t* ids = reinterpret_cast<const uint16_t*>(rec);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            idx[j] = j < s.k ? (ids[j] == 0xffffu ? -1 : (int)ids[j]) : -1;
            w[j]   = j < s.k ? rec[s.k / 2 + j] : 0.f;
        }
        const float* tl = rec + solve_rec_tail(s.k);
        return make_float4(tl[0], tl[1], tl[2], tl[3]);
    }
    if (s.k == K && (K % 4) == 0) {
        const float4* v = (const float4*)rec;  // (2K+4)*4 bytes is a multiple of 16
#pragma unroll
        for (int q = 0; q < K / 4; ++q) {
            const float4 i4 = v[q], w4 = v[K / 4 + q];
            idx[4 * q] = __float_as_int(i4.x), idx[4 * q + 1] = __float_as_int(i4.y);
            idx[4 * q + 2] = __float_as_int(i4.z), idx[4 * q + 3] = __float_as_int(i4.w);
            w[4 * q] = w4.x, w[4 * q + 1] = w4.y, w[4 * q + 2] = w4.z, w[4 * q + 3] = w4.w;
        }
        return v[K / 2];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        idx[j] = j < s.k ? __float_as_int(rec[j]) : -1;
        w[j]   = j < s.k ? rec[s.k + j] : 0.f;
    }
    return make_float4(rec[2 * s.k], rec[2 * s.k + 1], rec[2 * s.k + 2], rec[2 * s.k + 3]);
}

#ifdef DFA_PCG_PROFILE  // development builds: a workgroup's life in the assembly (tools/ref_assemble_phases.py)
__device__ unsigned long long asm_tbuf[32768 * 8];
extern "C" __attribute__((visibility("default"))) int dfa_dev_asm_timing(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(asm_tbuf), sizeof(unsigned long long) * 8 * (size_t)n);
}
#endif

template <int K>
__global__ __launch_bounds__(256) void assemble_kernel(SolveView s, SolveState* __restrict__ st, int save_base, int xcd_map,
                                                       float amax_unset) {
    __shared__ int key[HASH];
    __shared__ long long val[2 * HASH];  // [0, HASH): sums of the non-negative addends, [HASH, 2 HASH): of the negative ones' magnitudes
    __shared__ float gpart[4][3];
    __shared__ int wave_cnt[4];
    __shared__ int ovf;
    if (st->done || st->converged) return;
    // (workgroup -> node in launch order.  A contiguous node range per XCD — so that the rows a node shares with its
    // neighbours are fetched into one L2 instead of up to eight — left the launch at 345 us at C4: it was never bound by
    // the fetches.)
    int a = blockIdx.x;
#ifdef DFA_DEV_AB  // DFA_XCD_MAP=1: the experiment above, kept for its counters (profiles/r06_xcd_map.md)
    if (xcd_map && (s.D & 7) == 0) a = (a & 7) * (s.D >> 3) + (a >> 3);
#endif
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef DFA_PCG_PROFILE
    long long t0_ = clock64(), t1_, t2_, t3_, t4_;
    const unsigned long long w0_ = wall_clock64();
#endif
    for (int i = threadIdx.x; i < HASH; i += 256) key[i] = -1, val[i] = val[i + HASH] = 0ll;
    if (threadIdx.x == 0) ovf = 0;
    __syncthreads();
#ifdef DFA_PCG_PROFILE
    t1_ = clock64();
#endif

    const int beg = s.node_ptr[a], end = s.node_ptr[a + 1];
    const FixedScale fx = fixed_scale_for_rows(solve_fixed_scale(st->amax > 0.f ? st->amax : amax_unset), end - beg);
    float gx = 0.f, gy = 0.f, gz = 0.f, dsum = 0.f;
    for (int p = beg + (int)threadIdx.x; p < end; p += 256) {
        const uint32_t e = s.node_list[p];
        const size_t r   = e / (uint32_t)s.k;
        const int slot   = (int)(e - (uint32_t)r * (uint32_t)s.k);
        int idx[K];
        float w[K];
        const float4 et = load_record<K>(s, r, idx, w);  // (e.x, e.y, e.z, tau)
        float wa = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) wa = (j == slot) ? w[j] : wa;
        const float tw = et.w * wa;
        gx += tw * et.x, gy += tw * et.y, gz += tw * et.z;
        if (et.w != 0.f) {
            // first probe of all k columns read together (keys never change once set): the common
            // case "column already present" costs one LDS read + one fire-and-forget ds_add instead
            // of a returning CAS per column
            // (measured and not kept, tools/ref_assemble_phases.py at C4 / C3: every lane taking its columns in the order
            // (lane + t) mod K, so that a step's adds spread over K addresses — 149 / 86 us against 141 / 81, the selects cost
            // more than the conflicts; a thread's rows 2 or 4 at a time with their loads in flight together — a workgroup
            // lives 18 us instead of 23 but fewer are resident: 140-163 / 81-91 us)
            uint32_t h0[K];
            int k0[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                h0[j] = ((uint32_t)idx[j] * 2654435761u) >> (32 - 9);
                k0[j] = key[h0[j]];
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int b = idx[j];
                if (b < 0) continue;
                const float v = tw * w[j];
                if (b == a) {  // the diagonal is hit by every row: kept in a register
                    dsum += v;
                    continue;
                }
                if (k0[j] == b) {  // the column is in the table where its first probe looks: nearly every pair after the first rows
                    fixed_add(&val[h0[j]], v, fx.up);
                    continue;
                }
                uint32_t h = h0[j];
                int cur    = k0[j];
#pragma unroll 1
                for (int probes = 0;; ++probes) {
                    if (cur == -1) cur = atomicCAS(&key[h], -1, b), cur = cur == -1 ? b : cur;
                    if (cur == b) {
                        fixed_add(&val[h], v, fx.up);
                        break;
                    }
                    if (probes >= HASH) {
                        ovf = 1;
                        break;
                    }
                    h   = (h + 1) & HASH_MASK;
                    cur = key[h];
                }
            }
        }
    }
    // the diagonal: one add per wave into its (pre-inserted) slot
    dsum = wave_total(dsum);
    if (lane == 0 && dsum != 0.f) {
        uint32_t h = ((uint32_t)a * 2654435761u) >> (32 - 9);
        for (int probes = 0; probes < HASH; ++probes, h = (h + 1) & HASH_MASK) {
            const int cur = atomicCAS(&key[h], -1, a);
            if (cur == -1 || cur == a) {
                fixed_add(&val[h], dsum, fx.up);
                break;
            }
        }
    }
#ifdef DFA_PCG_PROFILE
    t2_ = clock64();
#endif
    gx = wave_total(gx), gy = wave_total(gy), gz = wave_total(gz);
    if (lane == 0) gpart[wave][0] = gx, gpart[wave][1] = gy, gpart[wave][2] = gz;
    __syncthreads();
#ifdef DFA_PCG_PROFILE
    t3_ = clock64();
#endif
    // compact the hash into the ELL row (slot-major: entry q of row a at [q*D + a]: a slot of all rows is
    // one contiguous 4*D-byte segment, which the sorted-row PCG prologues re-read from L1/L2); each wave owns
    // HASH/4 consecutive hash slots, wave offsets come from a 4-entry LDS prefix
    constexpr int PER_WAVE = HASH / 4;
    int wcnt               = 0;
    for (int base = 0; base < PER_WAVE; base += 64)
        wcnt += __popcll(__ballot(key[wave * PER_WAVE + base + lane] >= 0));
    if (lane == 0) wave_cnt[wave] = wcnt;
    __syncthreads();
    int pos0 = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        pos0 += w < wave ? wave_cnt[w] : 0;
        total += wave_cnt[w];
    }
    float diag = 0.f;
    for (int base = 0; base < PER_WAVE; base += 64) {
        const int kk     = key[wave * PER_WAVE + base + lane];
        const int hq     = wave * PER_WAVE + base + lane;
        const float vv   = (float)((double)(val[hq] - val[hq + HASH]) * fx.down);
        const bool valid = kk >= 0;
        const uint64_t m = __ballot(valid);
        const int pos    = pos0 + __popcll(m & ((1ull << lane) - 1ull));
        if (valid) {
            if (pos < s.ell_cap) {
                s.ell[(size_t)pos * s.D + a] = make_float2(vv, __int_as_float(kk));
            }
            if (kk == a) s.diag[a] = vv, diag = 1.f;
        }
        pos0 += __popcll(m);
    }
    const bool has_diag = __syncthreads_or(diag != 0.f);
    if (threadIdx.x == 0) {
        s.ell_cnt[a] = min(total, s.ell_cap);
        if (!has_diag) s.diag[a] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gc = (gpart[0][c] + gpart[1][c]) + (gpart[2][c] + gpart[3][c]);
            s.g[3 * a + c] = gc;
            if (save_base) s.g_base[3 * a + c] = gc, s.t_base[3 * a + c] = s.t[3 * a + c];
        }
        // 2048 device-scope atomics on one word cost ~11 ns each: only the (few) blocks that raise
        // the running maximum issue one
        if (total > *(volatile int*)&st->max_row_nnz) atomicMax(&st->max_row_nnz, total);
        if (total > s.ell_cap || ovf) st->overflow = 1;
#ifdef DFA_PCG_PROFILE
        t4_ = clock64();
        if (a == 7 && !s.team_ctl) st->prof[6] = (t1_ - t0_) * 1000000 + (t2_ - t1_), st->prof[7] = (t3_ - t2_) * 1000000 + (t4_ - t3_);  // (plans with a team PCG: its own counters)
        if (a < 32768) {
            unsigned long long* o = asm_tbuf + 8 * (size_t)a;
            o[0] = w0_, o[1] = wall_clock64() - w0_, o[2] = (unsigned long long)(end - beg), o[3] = (unsigned long long)total;
            o[4] = t1_ - t0_, o[5] = t2_ - t1_, o[6] = t3_ - t2_, o[7] = t4_ - t3_;
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------
// Order-stable variant (SolveView::deterministic).  Three things make two runs of the default path differ in the last
// bits: the transposition fills a node's row list in the order its LDS cursor atomics land, the assembly above adds
// into LDS with float atomics from four waves and compacts the hash in slot order (which depends on who inserted a key
// first), and the PCG kernels place rows of equal length by an atomic cursor (which thread owns which row decides the
// order of the inner products' partial sums).  Here: lists sorted, per-wave private sums added in wave order, rows of
// the matrix sorted by column, equal-length rows in index order.
constexpr int DET_SORT_MAX = 4096;

__global__ __launch_bounds__(256) void sort_node_lists_kernel(const int32_t* __restrict__ node_ptr, uint32_t* __restrict__ node_list) {
    __shared__ uint32_t buf[DET_SORT_MAX];
    const int a = blockIdx.x, tid = threadIdx.x;
    const int beg = node_ptr[a], len = node_ptr[a + 1] - beg;
    if (len < 2 || len > DET_SORT_MAX) return;  // (longer lists keep the order of the transposition)
    int n2 = 1;
    while (n2 < len) n2 <<= 1;
    for (int i = tid; i < n2; i += 256) buf[i] = i < len ? node_list[beg + i] : 0xffffffffu;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < n2 / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t x = buf[lo], y = buf[hi];
                if ((x > y) == up) buf[lo] = y, buf[hi] = x;
            }
            __syncthreads();
        }
    for (int i = tid; i < len; i += 256) node_list[beg + i] = buf[i];
}

template <int K>
__global__ __launch_bounds__(256) void assemble_det_kernel(SolveView s, SolveState* __restrict__ st, int save_base, float amax_unset) {
    __shared__ int key[HASH];
    __shared__ long long val[2 * HASH];  // fixed-point sums (see FixedScale, fixed_add): integer adds commute, any order gives the same bits
    __shared__ float gpart[4][3], dpart[4];
    __shared__ int ovf, nkeys;
    if (st->done || st->converged) return;
    const int a    = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < HASH; i += 256) key[i] = -1, val[i] = val[i + HASH] = 0ll;
    if (threadIdx.x == 0) ovf = 0, nkeys = 0;
    __syncthreads();
    const int beg = s.node_ptr[a], end = s.node_ptr[a + 1];
    const FixedScale fx = fixed_scale_for_rows(solve_fixed_scale(st->amax > 0.f ? st->amax : amax_unset), end - beg);
    // pass 1: the set of columns (keys only), the gradient and the diagonal
    float gx = 0.f, gy = 0.f, gz = 0.f, dsum = 0.f;
    for (int p = beg + (int)threadIdx.x; p < end; p += 256) {
        const uint32_t e = s.node_list[p];
        const size_t r   = e / (uint32_t)s.k;
        const int slot   = (int)(e - (uint32_t)r * (uint32_t)s.k);
        int idx[K];
        float w[K];
        const float4 et = load_record<K>(s, r, idx, w);
        float wa = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) wa = (j == slot) ? w[j] : wa;
        const float tw = et.w * wa;
        gx += tw * et.x, gy += tw * et.y, gz += tw * et.z;
        if (et.w != 0.f) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int b = idx[j];
                if (b < 0) continue;
                if (b == a) {
                    dsum += tw * w[j];
                    continue;
                }
                uint32_t h = ((uint32_t)b * 2654435761u) >> (32 - 9);
                for (int probes = 0;; ++probes) {
                    const int cur = atomicCAS(&key[h], -1, b);
                    if (cur == -1 || cur == b) break;
                    if (probes >= HASH) {
                        ovf = 1;
                        break;
                    }
                    h = (h + 1) & HASH_MASK;
                }
            }
        }
    }
    dsum = wave_total(dsum), gx = wave_total(gx), gy = wave_total(gy), gz = wave_total(gz);
    if (lane == 0) dpart[wave] = dsum, gpart[wave][0] = gx, gpart[wave][1] = gy, gpart[wave][2] = gz;
    __syncthreads();
    const float dtot = (dpart[0] + dpart[1]) + (dpart[2] + dpart[3]);
    if (threadIdx.x == 0 && dtot != 0.f) {  // the diagonal is a column like the others (as in assemble_kernel: only if non-zero)
        uint32_t h = ((uint32_t)a * 2654435761u) >> (32 - 9);
        for (int probes = 0; probes < HASH; ++probes, h = (h + 1) & HASH_MASK)
            if (key[h] == -1) {
                key[h] = a;
                break;
            }
    }
    __syncthreads();
    // pass 2: the values, into this wave's copy (the slot of a column: read-only probes now)
    for (int p = beg + (int)threadIdx.x; p < end; p += 256) {
        const uint32_t e = s.node_list[p];
        const size_t r   = e / (uint32_t)s.k;
        const int slot   = (int)(e - (uint32_t)r * (uint32_t)s.k);
        int idx[K];
        float w[K];
        const float4 et = load_record<K>(s, r, idx, w);
        if (et.w == 0.f) continue;
        float wa = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) wa = (j == slot) ? w[j] : wa;
        const float tw = et.w * wa;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int b = idx[j];
            if (b < 0 || b == a) continue;
            uint32_t h = ((uint32_t)b * 2654435761u) >> (32 - 9);
            for (int probes = 0; probes <= HASH && key[h] != b; ++probes) h = (h + 1) & HASH_MASK;
            if (key[h] == b) fixed_add(&val[h], tw * w[j], fx.up);
        }
    }
    __syncthreads();
    // output: entry of column c at the position of c among the row's columns (ascending)
    int total = 0;
    for (int i = threadIdx.x; i < HASH; i += 256) total += key[i] >= 0;
    total = (int)wave_total((float)total);
    if (lane == 0) atomicAdd(&nkeys, total);
    __syncthreads();
    total = nkeys;
    bool has_diag = false;
    for (int i = threadIdx.x; i < HASH; i += 256) {
        const int kk = key[i];
        if (kk < 0) continue;
        int pos = 0;
        for (int q = 0; q < HASH; ++q) pos += key[q] >= 0 && key[q] < kk;
        const float vv = kk == a ? dtot : (float)((double)(val[i] - val[i + HASH]) * fx.down);
        if (pos < s.ell_cap) s.ell[(size_t)pos * s.D + a] = make_float2(vv, __int_as_float(kk));
        if (kk == a) s.diag[a] = vv, has_diag = true;
    }
    const bool any_diag = __syncthreads_or(has_diag);
    if (threadIdx.x == 0) {
        s.ell_cnt[a] = min(total, s.ell_cap);
        if (!any_diag) s.diag[a] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gc = (gpart[0][c] + gpart[1][c]) + (gpart[2][c] + gpart[3][c]);
            s.g[3 * a + c] = gc;
            if (save_base) s.g_base[3 * a + c] = gc, s.t_base[3 * a + c] = s.t[3 * a + c];
        }
        if (total > *(volatile int*)&st->max_row_nnz) atomicMax(&st->max_row_nnz, total);
        if (total > s.ell_cap || ovf) st->overflow = 1;
    }
}

// rows of equal length in index order: `from` holds the permutation as the atomic cursors left it (ranks in [0, D), rows
// grouped by length, hist[b] = end of bin b), `to` receives the order-stable one.  Called by all NT threads.
template <int NT, int R>
__device__ __forceinline__ void stable_equal_runs(const int32_t* from, int32_t* to, const int32_t* cnt_of_row, const int* hist, int D) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int rank = (int)threadIdx.x + NT * i;
        if (rank >= D) continue;
        const int row = from[rank];
        const int bin = 256 - min(cnt_of_row[row], 256);
        const int beg = bin > 0 ? hist[bin - 1] : 0, end = hist[bin];
        int before = 0;
        for (int q = beg; q < end; ++q) before += from[q] < row;
        to[beg + before] = row;
    }
}

// ------------------------------------------------------------------------------------------
// block-Jacobi PCG, one persistent workgroup of 1024 threads; thread owns rows tid + 1024*i.

// Workgroup total: DPP wave totals (float) -> one LDS slot per wave -> ONE barrier -> every
// thread adds the NWAVES partials in double.  `red` must alternate between two buffers on
// successive calls so that no second barrier is needed to protect the slots.
template <int NWAVES>
__device__ __forceinline__ double block_sum(float v, float* red) {
    const float w = wave_total(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i) tot += (double)red[i];
    return tot;
}

// PCG targets never go below the round-off floor of the SOLVE: 1e-12 of the first linearisation's (r0, z0), the level
// at which a whole linearisation is skipped.  A late Gauss-Newton iteration starts from a small gradient, and 1e-12 of
// THAT is out of float's reach — its PCG would polish noise until the iteration cap (C2: the third iteration spent 108
// PCG iterations to move the translations by 2e-7 m).  DFA_PCG_NO_SOLVE_FLOOR=1 (A/B): per-linearisation targets only.
__device__ int g_floor_off = 0;
__device__ __forceinline__ float solve_floor(const SolveState* st) {
    return g_floor_off ? 0.f : (float)(1e-12 * st->grad_first);
}

// DFA_PCG_PROFILE builds accumulate shader cycles per PCG phase (thread 0) into SolveState::prof
#ifdef DFA_PCG_PROFILE
#define PROF_MARK(i)                      \
    do {                                  \
        const long long now_ = clock64(); \
        pc_[i] += now_ - last_;           \
        last_ = now_;                     \
    } while (0)
#else
#define PROF_MARK(i)
#endif

// float flavour (fewer registers; used by the register-resident kernel)
template <int NWAVES>
__device__ __forceinline__ float block_sum_f(float v, float* red) {
    const float w = wave_total(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i) tot += red[i];
    return tot;
}

// Streaming PCG for systems too large for the register-resident kernel (D > 2048 or rows wider than
// its slots): one persistent workgroup, RPT rows per thread, the matrix re-read from L2 every
// iteration.  A single CU moves 64 B/clk through its vector memory path, so the bytes per iteration
// are what matters: the prologue counting-sorts the rows by length (wave-uniform loop bounds with
// almost no padding) and repacks the ELL image rank-major with 16-bit columns — 6 B per non-zero,
// fully coalesced — into the plan's workspace.
// The whole workgroup (NT threads, NT * RPT >= D) runs this: the body of pcg_kernel below, and the way out of the
// register-resident kernel when a row pair does not fit its slots (it used to be a second launch behind every
// register-resident one, which returned at once in the common case: 5 launches per C2 frame).
template <int NT, int RPT>
__device__ __forceinline__ void pcg_stream_body(const SolveView& s, SolveState* __restrict__ st, int max_iter, float pcg_tol,
                                                char* smem) {
    float4* p_s = (float4*)smem;                                      // D entries
    float* red0 = (float*)(smem + sizeof(float4) * (size_t)s.Dpad);  // 2 x 16 wave partials
    float* red1 = red0 + 16;
    int* hist   = (int*)(red1 + 16);                                  // 260 bins
    const int tid = threadIdx.x;
    const int D   = s.D;

    // ---- rows sorted by length (descending): rank -> row in s.pk_perm
    for (int i = tid; i < 260; i += NT) hist[i] = 0;
    __syncthreads();
    int my_cnt[RPT];
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
        const int row = tid + NT * h;
        my_cnt[h]     = row < D ? min(s.ell_cnt[row], 256) : -1;
        if (my_cnt[h] >= 0) atomicAdd(&hist[256 - my_cnt[h]], 1);
    }
    __syncthreads();
    if (tid < 64) {
        int loc[5], sum = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int b = tid * 5 + j;
            loc[j]      = b < 257 ? hist[b] : 0;
            sum += loc[j];
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (tid >= o) incl += t;
        }
        int off = incl - sum;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int b = tid * 5 + j;
            if (b < 257) hist[b] = off;
            off += loc[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < RPT; ++h)
        if (my_cnt[h] >= 0) (s.deterministic ? s.pk_perm2 : s.pk_perm)[atomicAdd(&hist[256 - my_cnt[h]], 1)] = tid + NT * h;
    __syncthreads();  // workgroup-scope visibility of pk_perm
    if (s.deterministic) {  // (hist[b] is now the end of bin b)
        stable_equal_runs<NT, RPT>(s.pk_perm2, s.pk_perm, s.ell_cnt, hist, D);
        __syncthreads();
    }

    // ---- my rows = ranks tid + NT*i; repack them rank-major (coalesced from now on)
    int row[RPT], rcnt[RPT], wmax[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int rank = tid + NT * i;
        row[i]         = rank < D ? s.pk_perm[rank] : -1;
        rcnt[i]        = row[i] >= 0 ? min(s.ell_cnt[row[i]], 256) : 0;
        for (int q = 0; q < rcnt[i]; ++q) {
            const float2 e                  = s.ell[(size_t)q * D + row[i]];
            s.pk_vals[(size_t)q * D + rank] = e.x;
            s.pk_cols[(size_t)q * D + rank] = (uint16_t)__float_as_int(e.y);
        }
        int m = rcnt[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        wmax[i] = m;  // wave-uniform
    }
    __syncthreads();

    float x[RPT][3], r[RPT][3], p[RPT][3], minv[RPT];
    float rz_loc = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        if (row[i] >= 0) {
            const float d = s.diag[row[i]];
            minv[i]       = d > FLT_EPSILON ? 1.0f / d : 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                x[i][c] = 0.f;
                r[i][c] = s.g[3 * row[i] + c];
                p[i][c] = minv[i] * r[i][c];
                rz_loc  = fmaf(r[i][c], p[i][c], rz_loc);
            }
            p_s[row[i]] = make_float4(p[i][0], p[i][1], p[i][2], 0.f);
        } else {
            minv[i] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) x[i][c] = r[i][c] = p[i][c] = 0.f;
        }
    }
    double rz           = block_sum<NT / 64>(rz_loc, red1);  // the barrier inside also publishes p_s
    const double rz0    = rz;
    const double floor_ = 1e-12;  // squared-residual-ratio floor of float arithmetic
    const double tol2   = (double)pcg_tol * (double)pcg_tol > floor_ ? (double)pcg_tol * (double)pcg_tol : floor_;
    int it              = 0;
    const bool skip     = st->grad_first > 0.0 && rz0 <= floor_ * st->grad_first;
    const double target = fmax(tol2 * rz0, (double)solve_floor(st));
    if (!skip) {
        while (it < max_iter) {
            if (!(rz > 0.0)) break;
            float ap[RPT][3];
            float pap_loc = 0.f;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int rank  = tid + NT * i;
                const int rankc = rank < D ? rank : 0;
                float ax = 0.f, ay = 0.f, az = 0.f;
                // 8 (coalesced, rank-major) loads in flight per lane; wmax is wave-uniform.  Loads are
                // unconditional (slots up to the ELL capacity are valid memory) and masked AFTER the
                // load: a load under a per-element condition makes hipcc branch around it and wait
                // vmcnt(0) each time — 16 serial L2 round trips per chunk.
                for (int q0 = 0; q0 < wmax[i]; q0 += 8) {
                    int colv[8];
                    float valv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int q = min(q0 + u, 255);
                        colv[u]     = s.pk_cols[(size_t)q * D + rankc];
                        valv[u]     = s.pk_vals[(size_t)q * D + rankc];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool ok = q0 + u < rcnt[i];
                        colv[u]       = ok ? colv[u] : 0;
                        valv[u]       = ok ? valv[u] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float4 pc = p_s[colv[u]];
                        ax = fmaf(valv[u], pc.x, ax), ay = fmaf(valv[u], pc.y, ay), az = fmaf(valv[u], pc.z, az);
                    }
                }
                ap[i][0] = ax, ap[i][1] = ay, ap[i][2] = az;
                pap_loc = fmaf(p[i][0], ax, fmaf(p[i][1], ay, fmaf(p[i][2], az, pap_loc)));
            }
            const double pAp = block_sum<NT / 64>(pap_loc, red0);
            if (!(pAp > 0.0)) break;
            const float alpha = (float)(rz / pAp);
            float rzn_loc     = 0.f;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    x[i][c] = fmaf(alpha, p[i][c], x[i][c]);
                    r[i][c] = fmaf(-alpha, ap[i][c], r[i][c]);
                    rzn_loc = fmaf(r[i][c], minv[i] * r[i][c], rzn_loc);
                }
            }
            const double rz_new = block_sum<NT / 64>(rzn_loc, red1);
            ++it;
            if (rz_new <= target) break;
            const float beta = (float)(rz_new / rz);
            // every thread has read p_s for this iteration (two barriers passed since the SpMV)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c) p[i][c] = fmaf(beta, p[i][c], minv[i] * r[i][c]);
                if (row[i] >= 0) p_s[row[i]] = make_float4(p[i][0], p[i][1], p[i][2], 0.f);
            }
            rz = rz_new;
            __syncthreads();
        }
    }
    // t += delta
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        if (row[i] >= 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) s.t[3 * row[i] + c] += x[i][c];
        }
    }
    if (tid == 0) {
        if (st->grad_first == 0.0) st->grad_first = rz0;
        st->pcg_iters += it;
        st->gn_iters += 1;
        if (skip) solve_mark_at_floor(st);
    }
}

#ifdef DFA_DEV_AB  // the streaming kernel as a launch of its own (DFA_PCG_VARIANT=0 / 4): development builds only
template <int NT, int RPT>
__global__ __launch_bounds__(NT) void pcg_kernel(SolveView s, SolveState* __restrict__ st, int max_iter,
                                                   float pcg_tol) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (st->done) return;
    if (st->converged) {  // no-op iteration (see SolveState::converged)
        if (threadIdx.x == 0) st->gn_iters += 1, st->gn_noop += 1;
        return;
    }
    pcg_stream_body<NT, RPT>(s, st, max_iter, pcg_tol, smem);
}
#endif  // DFA_DEV_AB

// ------------------------------------------------------------------------------------------
// PCG with the WHOLE matrix in registers (D <= 2 * NT rows).
//
// The matrix is constant over the PCG iterations, and a single CU can stream at most 64 B/clk
// through its vector memory path: re-reading a padded ELL image every iteration costs more than
// everything else in the loop together (measured: 6.1 k of 10.5 k cycles per iteration).  Here a
// thread keeps its rows' entries in registers for the whole solve — E values and E/2 words of
// packed 16-bit columns (as LDS byte offsets) — so the only per-iteration memory traffic left is the LDS gather of p
// (one ds_read_b128 per non-zero).
//
// Register arrays need compile-time indices, so row lengths must be (nearly) uniform across the
// lanes of a wave or the padding eats the gain.  The prologue therefore counting-sorts the rows
// by length in LDS and gives thread t the t-th LONGEST row ("A", slots 0.. upwards) and the t-th
// SHORTEST row ("B", slots E-1.. downwards): lengths vary slowly along a wave, nA + nB is about
// the true row-pair length, and both loop bounds are wave-uniform (no divergence, no selects).
// Entries of B that do not fit (rare) are streamed from L2 each iteration.
//
// NC = 3: one workgroup, the three coordinates share alpha / beta (CG on A (x) I3 as one system).
// NC = 1: JtJ = A (x) I3 is three INDEPENDENT scalar systems with the same matrix — workgroup c of three
// solves coordinate c on its own CU.  The gather shrinks from one ds_read_b128 + 3 FMAs per non-zero to one
// ds_read_b32 + 1 FMA (the LDS pipe moves 128 B/clk: 8 clocks per wave-wide b128 read, 2 per b32 read).  Every
// coordinate stops at (r, z)_c <= tol^2 (r0, z0)_joint / 3, which implies the joint stopping rule.
template <int NT, int P, int E, int NC>
__global__ __launch_bounds__(NT) void pcg_paired_kernel(SolveView s, SolveState* __restrict__ st, int max_iter,
                                                        float pcg_tol) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D     = s.D;
    const int c0    = NC == 1 ? (int)blockIdx.x : 0;  // first coordinate of this workgroup
    float4* p_s     = (float4*)smem;  // D x 16 B (NC = 3) / D x 4 B (NC = 1) in the same 16 B x Dpad region
    float* p_s1     = (float*)smem;
    float* red0     = (float*)(smem + sizeof(float4) * (size_t)s.Dpad);
    float* red1     = red0 + 16;
    int* hist       = (int*)(red1 + 16);  // 258 bins
    int* perm       = hist + 260;         // D row ids, longest row first
    if (st->done) return;
    const int tid = threadIdx.x;
    if (st->converged) {  // no-op iteration (see SolveState::converged); booked once, by whoever solves this plan
        if (tid == 0 && blockIdx.x == 0) st->gn_iters += 1, st->gn_noop += 1;
        return;
    }
    constexpr int R = 2 * P;  // rows per thread

    // ---- (r0, z0) of the joint system first, rows in natural order: it scales both stopping rules, and a gradient at
    // the round-off floor ends the launch here, before the sort and the 25 us of loading the matrix into registers
    float rzj_loc = 0.f;
#pragma unroll
    for (int h = 0; h < R; ++h) {
        const int row = tid + NT * h;
        if (row < D) {
            const float d    = s.diag[row];
            const float minv = d > FLT_EPSILON ? 1.0f / d : 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float g = s.g[3 * row + c];
                rzj_loc       = fmaf(g, minv * g, rzj_loc);
            }
        }
    }
    const float rz0 = block_sum_f<NT / 64>(rzj_loc, red1);
    const bool skip = st->grad_first > 0.0 && (double)rz0 <= 1e-12 * st->grad_first;
    if (skip) {  // the same decision in every workgroup; the first one books the (empty) iteration
        if (tid == 0 && blockIdx.x == 0) {
            st->gn_iters += 1;
            solve_mark_at_floor(st);
        }
        return;
    }

    // ---- rows sorted by length (descending), counting sort in LDS
    for (int i = tid; i < 260; i += NT) hist[i] = 0;
    __syncthreads();
    int my_cnt[R];
#pragma unroll
    for (int h = 0; h < R; ++h) {
        const int row = tid + NT * h;
        my_cnt[h]     = row < D ? min(s.ell_cnt[row], 256) : -1;
        if (my_cnt[h] >= 0) atomicAdd(&hist[256 - my_cnt[h]], 1);  // bin 0 = longest
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan of 257 bins by one wave (5 bins per lane)
        int loc[5], sum = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int b = tid * 5 + j;
            loc[j]      = b < 257 ? hist[b] : 0;
            sum += loc[j];
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (tid >= o) incl += t;
        }
        int off = incl - sum;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int b = tid * 5 + j;
            if (b < 257) hist[b] = off;
            off += loc[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < R; ++h)
        if (my_cnt[h] >= 0) (s.deterministic ? perm + s.Dpad : perm)[atomicAdd(&hist[256 - my_cnt[h]], 1)] = tid + NT * h;
    __syncthreads();
    if (s.deterministic) {  // (the launcher sized the LDS for a second array of row ids; hist[b] is now the end of bin b)
        stable_equal_runs<NT, R>(perm + s.Dpad, perm, s.ell_cnt, hist, D);
        __syncthreads();
    }

    // ---- this thread's pairs: pair j = rank j*NT + t (long, "A") and rank D-1-j*NT-t (short, "B")
    int rowA[P], rowB[P], nA[P], nB[P], cntA_[P], cntB_[P];
    bool unfit_any = false;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int ia = j * NT + tid, ib = D - 1 - j * NT - tid;
        rowA[j]      = (ia < D && ia <= ib) ? perm[ia] : -1;
        rowB[j]      = (ib >= 0 && ib > ia) ? perm[ib] : -1;
        const int cntA = rowA[j] >= 0 ? min(s.ell_cnt[rowA[j]], 256) : 0;
        const int cntB = rowB[j] >= 0 ? min(s.ell_cnt[rowB[j]], 256) : 0;
        cntA_[j] = cntA, cntB_[j] = cntB;
        int na         = min(cntA, E);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) na = max(na, __shfl_xor(na, o, 64));
        na             = (na + 1) & ~1;  // even: a packed column word never mixes A and B slots
        const int capB = E - na;         // wave-uniform
        const int regB = min(cntB, capB);
        int nb         = regB;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nb = max(nb, __shfl_xor(nb, o, 64));
        // wave-uniform by construction; readfirstlane tells the compiler, so that the slot-range tests of the
        // PCG loop become scalar branches instead of per-lane selects over both accumulators
        nA[j] = __builtin_amdgcn_readfirstlane(na), nB[j] = __builtin_amdgcn_readfirstlane(min((nb + 1) & ~1, capB));
        unfit_any |= cntA > E || cntB > capB;
    }
    // A pair that does not fit E slots (k = 8 graphs always, k = 4 hardly ever): the system is streamed from L2 instead,
    // all three coordinates by the first workgroup — decided from the row lengths alone, before any matrix entry is
    // loaded, and inside this launch.
    if (__syncthreads_or(unfit_any)) {
        if (NC == 3 || blockIdx.x == 0) pcg_stream_body<NT, 2 * P>(s, st, max_iter, pcg_tol, smem);
        return;
    }
    float mval[P][E];
    uint32_t mcol[P][E / 2];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int cntA = cntA_[j], cntB = cntB_[j];
        const int na = nA[j];
        const int regB = min(cntB, E - na);
        const int rA = rowA[j] >= 0 ? rowA[j] : 0, rB = rowB[j] >= 0 ? rowB[j] : 0;
        // values and columns -> registers for the whole solve (slot q: entry q of A for q < nA, entry
        // E-1-q of B otherwise); two 16-bit columns per register
#pragma unroll
        for (int q2 = 0; q2 < E / 2; ++q2) {
            uint32_t packed = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int q     = 2 * q2 + h;
                const bool isA  = q < na;
                const int ent   = isA ? q : E - 1 - q;
                const int r     = isA ? rA : rB;
                const bool live = isA ? (q < cntA) : (E - 1 - q < regB);
                const float2 e  = s.ell[(size_t)ent * D + r];  // unconditional 8-byte load, masked after
                float v         = e.x;
                int col         = __float_as_int(e.y);
                if (!live) v = 0.f, col = 0;
                mval[j][q] = v;
                packed |= (uint32_t)(col << (NC == 3 ? 4 : 2)) << (16 * h);  // byte offset of p[col] in LDS
            }
            mcol[j][q2] = packed;
            if ((q2 & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the loads in flight
        }
    }
    float xA[P][NC], rA_[P][NC], pA[P][NC], xB[P][NC], rB_[P][NC], pB[P][NC], minvA[P], minvB[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        minvA[j] = minvB[j] = 0.f;
        if (rowA[j] >= 0) {
            const float d = s.diag[rowA[j]];
            minvA[j]      = d > FLT_EPSILON ? 1.0f / d : 1.0f;
        }
        if (rowB[j] >= 0) {
            const float d = s.diag[rowB[j]];
            minvB[j]      = d > FLT_EPSILON ? 1.0f / d : 1.0f;
        }
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const int c    = c0 + cc;
            const float ga = rowA[j] >= 0 ? s.g[3 * rowA[j] + c] : 0.f;
            const float gb = rowB[j] >= 0 ? s.g[3 * rowB[j] + c] : 0.f;
            xA[j][cc] = xB[j][cc] = 0.f;
            rA_[j][cc] = ga, rB_[j][cc] = gb;
            pA[j][cc] = minvA[j] * ga, pB[j][cc] = minvB[j] * gb;
        }
        if (NC == 3) {
            if (rowA[j] >= 0) p_s[rowA[j]] = make_float4(pA[j][0], pA[j][1], pA[j][NC - 1], 0.f);
            if (rowB[j] >= 0) p_s[rowB[j]] = make_float4(pB[j][0], pB[j][1], pB[j][NC - 1], 0.f);
        } else {
            if (rowA[j] >= 0) p_s1[rowA[j]] = pA[j][0];
            if (rowB[j] >= 0) p_s1[rowB[j]] = pB[j][0];
        }
    }
    __syncthreads();  // p in LDS
    float rz = rz0;   // (NC = 1 forms its own (r, z) inside the loop)
    const float floor_ = 1e-12f;
    const float tol2   = pcg_tol * pcg_tol > floor_ ? pcg_tol * pcg_tol : floor_;
    // (never below the solve's round-off floor, see solve_floor)
    const float joint  = fmaxf(tol2 * rz0, solve_floor(st));
    // NC = 1: this coordinate's share of the joint target; a coordinate already below it does no iteration
    const float target = NC == 3 ? joint : joint * (1.0f / 3.0f);
    const float rz_min = NC == 3 ? 0.f : target;
    int it             = 0;
    const char* pbase   = (const char*)p_s;
#ifdef DFA_PCG_PROFILE
    long long pc_[6] = {0, 0, 0, 0, 0, 0};
    long long last_  = clock64();
#endif
    // a = A p for this thread's rows (p gathered from LDS)
    auto matvec = [&](float (&aA)[P][NC], float (&aB)[P][NC]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
#pragma unroll
            for (int c = 0; c < NC; ++c) aA[j][c] = aB[j][c] = 0.f;
            // Two slots share a packed column word.  The empty asm makes the word opaque per iteration (otherwise
            // the compiler hoists the unpacking out of the PCG loop and doubles the registers the columns occupy)
            // and, being volatile, keeps the slot-range tests below real scalar branches: if-converted they cost
            // a select per slot AND the FMAs of both accumulators (measured: 4.5 VALU per non-zero, now 2).
            auto slots2 = [&](const int q, float (&acc)[NC]) __attribute__((always_inline)) {
                uint32_t cw = mcol[j][q / 2];
                asm volatile("" : "+v"(cw));
                const float v0 = mval[j][q], v1 = mval[j][q + 1];
                if (NC == 3) {
                    const float4 g0 = *(const float4*)(pbase + (cw & 0xffffu));
                    const float4 g1 = *(const float4*)(pbase + (cw >> 16));
                    asm volatile("" ::"v"(g0.w), "v"(g1.w));  // keep ds_read_b128
                    acc[0]      = fmaf(v1, g1.x, fmaf(v0, g0.x, acc[0]));
                    acc[NC / 2] = fmaf(v1, g1.y, fmaf(v0, g0.y, acc[NC / 2]));
                    acc[NC - 1] = fmaf(v1, g1.z, fmaf(v0, g0.z, acc[NC - 1]));
                } else {
                    const float g0 = *(const float*)(pbase + (cw & 0xffffu));
                    const float g1 = *(const float*)(pbase + (cw >> 16));
                    acc[0]         = fmaf(v1, g1, fmaf(v0, g0, acc[0]));
                }
            };
            auto slots4 = [&](const int q, float (&acc)[NC]) __attribute__((always_inline)) {  // 4 gathers in flight
                uint32_t c01 = mcol[j][q / 2], c23 = mcol[j][q / 2 + 1];
                asm volatile("" : "+v"(c01), "+v"(c23));
                const float v0 = mval[j][q], v1 = mval[j][q + 1], v2 = mval[j][q + 2], v3 = mval[j][q + 3];
                if (NC == 3) {
                    const float4 g0 = *(const float4*)(pbase + (c01 & 0xffffu));
                    const float4 g1 = *(const float4*)(pbase + (c01 >> 16));
                    const float4 g2 = *(const float4*)(pbase + (c23 & 0xffffu));
                    const float4 g3 = *(const float4*)(pbase + (c23 >> 16));
                    asm volatile("" ::"v"(g0.w), "v"(g1.w), "v"(g2.w), "v"(g3.w));
                    acc[0]      = fmaf(v3, g3.x, fmaf(v2, g2.x, fmaf(v1, g1.x, fmaf(v0, g0.x, acc[0]))));
                    acc[NC / 2] = fmaf(v3, g3.y, fmaf(v2, g2.y, fmaf(v1, g1.y, fmaf(v0, g0.y, acc[NC / 2]))));
                    acc[NC - 1] = fmaf(v3, g3.z, fmaf(v2, g2.z, fmaf(v1, g1.z, fmaf(v0, g0.z, acc[NC - 1]))));
                } else {
                    const float g0 = *(const float*)(pbase + (c01 & 0xffffu));
                    const float g1 = *(const float*)(pbase + (c01 >> 16));
                    const float g2 = *(const float*)(pbase + (c23 & 0xffffu));
                    const float g3 = *(const float*)(pbase + (c23 >> 16));
                    acc[0]         = fmaf(v3, g3, fmaf(v2, g2, fmaf(v1, g1, fmaf(v0, g0, acc[0]))));
                }
            };
            // row A: slots [0, nA) upwards; row B: slots [E - nB, E) from the top (nA, nB even, wave-uniform).
            // (8 gathers in flight per step measured no faster: the loop is bound by LDS bank conflicts — a
            // random 4-byte gather costs ~6.5 clocks per wave instruction against 2 conflict-free.)
#pragma unroll
            for (int q0 = 0; q0 < E; q0 += 4) {
                if (q0 + 4 <= nA[j]) {
                    slots4(q0, aA[j]);
                } else {
                    if (q0 + 2 <= nA[j]) slots2(q0, aA[j]);
                    break;
                }
            }
#pragma unroll
            for (int q0 = E - 4; q0 >= 0; q0 -= 4) {
                if (q0 >= E - nB[j]) {
                    slots4(q0, aB[j]);
                } else {
                    if (q0 + 2 >= E - nB[j]) slots2(q0 + 2, aB[j]);
                    break;
                }
            }
        }
    };
    auto publish = [&](const float (&vA)[P][NC], const float (&vB)[P][NC]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            if (NC == 3) {
                if (rowA[j] >= 0) p_s[rowA[j]] = make_float4(vA[j][0], vA[j][NC / 2], vA[j][NC - 1], 0.f);
                if (rowB[j] >= 0) p_s[rowB[j]] = make_float4(vB[j][0], vB[j][NC / 2], vB[j][NC - 1], 0.f);
            } else {
                if (rowA[j] >= 0) p_s1[rowA[j]] = vA[j][0];
                if (rowB[j] >= 0) p_s1[rowB[j]] = vB[j][0];
            }
        }
    };
    if (NC == 3) {
        // textbook PCG: two reductions and the publication of p = three barriers per iteration
        while (it < max_iter) {
            if (!(rz > rz_min)) break;
            PROF_MARK(5);
            float aA[P][NC], aB[P][NC];
            float pap_loc = 0.f;
            matvec(aA, aB);
#pragma unroll
            for (int j = 0; j < P; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) pap_loc = fmaf(pA[j][c], aA[j][c], fmaf(pB[j][c], aB[j][c], pap_loc));
            PROF_MARK(0);
            const float pAp = block_sum_f<NT / 64>(pap_loc, red0);
            PROF_MARK(1);
            if (!(pAp > 0.f)) break;
            const float alpha = rz / pAp;
            float rzn_loc     = 0.f;
#pragma unroll
            for (int j = 0; j < P; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    xA[j][c]  = fmaf(alpha, pA[j][c], xA[j][c]);
                    xB[j][c]  = fmaf(alpha, pB[j][c], xB[j][c]);
                    rA_[j][c] = fmaf(-alpha, aA[j][c], rA_[j][c]);
                    rB_[j][c] = fmaf(-alpha, aB[j][c], rB_[j][c]);
                    rzn_loc = fmaf(rA_[j][c], minvA[j] * rA_[j][c], fmaf(rB_[j][c], minvB[j] * rB_[j][c], rzn_loc));
                }
            PROF_MARK(2);
            const float rz_new = block_sum_f<NT / 64>(rzn_loc, red1);
            PROF_MARK(3);
            ++it;
            if (rz_new <= target) break;
            const float beta = rz_new / rz;
#pragma unroll
            for (int j = 0; j < P; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    pA[j][c] = fmaf(beta, pA[j][c], minvA[j] * rA_[j][c]);
                    pB[j][c] = fmaf(beta, pB[j][c], minvB[j] * rB_[j][c]);
                }
            publish(pA, pB);
            rz = rz_new;
            __syncthreads();
            PROF_MARK(4);
        }
    }
    if (NC == 1) {
        // Chronopoulos-Gear form of the same recurrence: the matrix multiplies u = M^-1 r, both inner products
        // (r, u) and (A u, u) come out of ONE reduction, and s = A p follows by recurrence — two barriers per
        // iteration instead of three (a reduction costs ~550 clocks of a ~4 700-clock iteration here).
        // In LDS: u (the prologue stored M^-1 r0).  pA / pB start as the zero direction.
        float sA[P][NC], sB[P][NC], uA[P][NC], uB[P][NC];
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                uA[j][c] = pA[j][c], uB[j][c] = pB[j][c];
                pA[j][c] = pB[j][c] = sA[j][c] = sB[j][c] = 0.f;
            }
        float gamma_old = 1.f, alpha_old = 1.f;
        __syncthreads();  // every wave has read the prologue's sums before red0 / red1 are written again
        while (it < max_iter) {
            PROF_MARK(5);
            float wA[P][NC], wB[P][NC];
            matvec(wA, wB);
            float g_loc = 0.f, d_loc = 0.f;
#pragma unroll
            for (int j = 0; j < P; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    g_loc = fmaf(rA_[j][c], uA[j][c], fmaf(rB_[j][c], uB[j][c], g_loc));
                    d_loc = fmaf(wA[j][c], uA[j][c], fmaf(wB[j][c], uB[j][c], d_loc));
                }
            PROF_MARK(0);
            const float gw = wave_total(g_loc), dw = wave_total(d_loc);
            if ((tid & 63) == 0) red0[tid >> 6] = gw, red1[tid >> 6] = dw;
            __syncthreads();
            float gamma = 0.f, delta = 0.f;
#pragma unroll
            for (int i = 0; i < NT / 64; ++i) gamma += red0[i], delta += red1[i];
            PROF_MARK(1);
            if (!(gamma > rz_min)) break;  // converged: (r, M^-1 r) of the iterate in x
            const float beta  = it == 0 ? 0.f : gamma / gamma_old;
            const float denom = it == 0 ? delta : delta - beta * gamma / alpha_old;
            if (!(denom > 0.f)) break;
            const float alpha = gamma / denom;
#pragma unroll
            for (int j = 0; j < P; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    pA[j][c]  = fmaf(beta, pA[j][c], uA[j][c]);
                    pB[j][c]  = fmaf(beta, pB[j][c], uB[j][c]);
                    sA[j][c]  = fmaf(beta, sA[j][c], wA[j][c]);
                    sB[j][c]  = fmaf(beta, sB[j][c], wB[j][c]);
                    xA[j][c]  = fmaf(alpha, pA[j][c], xA[j][c]);
                    xB[j][c]  = fmaf(alpha, pB[j][c], xB[j][c]);
                    rA_[j][c] = fmaf(-alpha, sA[j][c], rA_[j][c]);
                    rB_[j][c] = fmaf(-alpha, sB[j][c], rB_[j][c]);
                    uA[j][c]  = minvA[j] * rA_[j][c];
                    uB[j][c]  = minvB[j] * rB_[j][c];
                }
            ++it;
            gamma_old = gamma, alpha_old = alpha;
            PROF_MARK(2);
            publish(uA, uB);  // every wave is past this iteration's gather (the reduction's barrier)
            __syncthreads();
            PROF_MARK(4);
        }
    }
#ifdef DFA_PCG_PROFILE
    if (tid == 0 && c0 == 0)
        for (int i = 0; i < 6; ++i) st->prof[i] += pc_[i];
#endif
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (rowA[j] >= 0) s.t[3 * rowA[j] + c0 + c] += xA[j][c];
            if (rowB[j] >= 0) s.t[3 * rowB[j] + c0 + c] += xB[j][c];
        }
    if (tid == 0) {
        if (NC == 3) {
            if (st->grad_first == 0.0) st->grad_first = (double)rz0;
            st->pcg_iters += it;
            st->gn_iters += 1;
        } else {
            // iterations of this launch = those of its slowest coordinate; the last workgroup to arrive books them
            atomicMax(&st->split_iters, it);
            __threadfence();
            if (atomicAdd(&st->split_ticket, 1u) == 2u) {
                __threadfence();
                st->pcg_iters += atomicExch(&st->split_iters, 0);
                st->split_ticket = 0u;
                if (st->grad_first == 0.0) st->grad_first = (double)rz0;
                st->gn_iters += 1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers

#define KDISPATCH(kernel, k, ...)                      \
    do {                                               \
        if ((k) <= 4) kernel<4> __VA_ARGS__;           \
        else if ((k) <= 8) kernel<8> __VA_ARGS__;      \
        else kernel<16> __VA_ARGS__;                   \
    } while (0)

// node -> (row, slot) lists of any R x k index array (shared with solve6.hip)
hipError_t solve_transpose_graph(const int32_t* ridx, size_t total, int D, int32_t* blk_hist, int32_t* node_ptr,
                                 uint32_t* node_list, hipStream_t st) {
    const size_t lds = sizeof(int32_t) * (size_t)D;
    tg_count_kernel<<<TG_BLOCKS, 1024, lds, st>>>(ridx, total, D, blk_hist);
    tg_colscan_kernel<<<(D + 255) / 256, 256, 0, st>>>(blk_hist, D);
    tg_fill_kernel<<<TG_BLOCKS, 1024, lds, st>>>(ridx, total, D, blk_hist, node_ptr, node_list);
    return hipGetLastError();
}

hipError_t solve_build_graph(const SolveView& s, SolveState* state, unsigned int* ticket, int nticket, hipStream_t st) {
    const int D = s.D, N = s.N, k = s.k;
    const size_t R = (size_t)N + (size_t)D * k, total = R * k;
    const size_t threads = R > (size_t)3 * D ? R : (size_t)3 * D;
    KDISPATCH(prepare_rows_kernel, k, <<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(s, state, ticket, nticket));
    const size_t lds = sizeof(int32_t) * (size_t)D;
    tg_count_kernel<<<TG_BLOCKS, 1024, lds, st>>>(s.ridx, total, D, s.blk_hist);
    tg_colscan_kernel<<<(D + 255) / 256, 256, 0, st>>>(s.blk_hist, D);
    tg_fill_kernel<<<TG_BLOCKS, 1024, lds, st>>>(s.ridx, total, D, s.blk_hist, s.node_ptr, s.node_list);
    if (s.deterministic) sort_node_lists_kernel<<<D, 256, 0, st>>>(s.node_ptr, s.node_list);
    return hipGetLastError();
}

int solve_residual_blocks(const SolveView& s) {
    const size_t R  = (size_t)s.N + (size_t)s.D * s.k;
    const size_t nb = (R + 255) / 256;
    return (int)(nb < (size_t)LIN_MAX_BLOCKS ? nb : (size_t)LIN_MAX_BLOCKS);
}

hipError_t solve_linearise(const SolveView& s, SolveState* state, double* cost_partials, unsigned int* ticket,
                           int update_weights, int mode, float gn_tol, float tukey_offset, float psi_data,
                           float w_reg_sq, float huber_psi, long long* iters_total, hipStream_t st) {
    const int nb = solve_residual_blocks(s);
    LineariseArgs a{update_weights, mode, gn_tol, tukey_offset, psi_data, w_reg_sq, iters_total, huber_psi};
    KDISPATCH(linearise_kernel, s.k, <<<nb, 256, 0, st>>>(s, state, cost_partials, ticket, a));
    return hipGetLastError();
}

hipError_t solve_reset(const SolveView& s, SolveState* state, unsigned int* ticket, int nticket, hipStream_t st) {
    reset_kernel<<<(3 * s.D + 255) / 256, 256, 0, st>>>(s.t, 3 * s.D, state, ticket, nticket);
    return hipGetLastError();
}

hipError_t solve_huber(const SolveView& s, float psi_reg, hipStream_t st) {
    huber_kernel<<<(s.D + 255) / 256, 256, 0, st>>>(s, psi_reg);
    return hipGetLastError();
}

hipError_t solve_assemble(const SolveView& s, SolveState* state, int save_base, float w_reg_sq, hipStream_t st) {
    // The rows carry w_reg^2 as their tau; the scale of the fixed-point sums comes from SolveState::amax, which the
    // re-weighting linearisation in front of this launch has found.  Should no such linearisation have stored one (amax is
    // still the 0 of solve_reset — nothing in the driver does that today), the sums take the bound every addend obeys,
    // max(1, w_reg^2), instead of a grid for addends of 1e-30 that the first real one would overflow.
    const float amax_unset = std::max(1.0f, w_reg_sq);
    if (s.deterministic) KDISPATCH(assemble_det_kernel, s.k, <<<s.D, 256, 0, st>>>(s, state, save_base, amax_unset));
    else KDISPATCH(assemble_kernel, s.k, <<<s.D, 256, 0, st>>>(s, state, save_base, dev_env_int("DFA_XCD_MAP", 0), amax_unset));
    return hipGetLastError();
}

// g = g_base - A (t - t_base): a thread per row over the slot-major ELL (entry q of row a at [q * D + a]: coalesced)
__global__ __launch_bounds__(256) void regradient_kernel(SolveView s, SolveState* __restrict__ st) {
    if (st->done || st->converged) return;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a == 0) st->cost_stale = 1, st->weights_fresh = 0;  // (a floor hit now ends this outer iteration only)
    if (a >= s.D) return;
    const int cnt = s.ell_cnt[a];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int q = 0; q < cnt; ++q) {
        const float2 e = s.ell[(size_t)q * s.D + a];
        const int col  = __float_as_int(e.y);
        sx = fmaf(e.x, s.t[3 * col] - s.t_base[3 * col], sx);
        sy = fmaf(e.x, s.t[3 * col + 1] - s.t_base[3 * col + 1], sy);
        sz = fmaf(e.x, s.t[3 * col + 2] - s.t_base[3 * col + 2], sz);
    }
    s.g[3 * a] = s.g_base[3 * a] - sx, s.g[3 * a + 1] = s.g_base[3 * a + 1] - sy, s.g[3 * a + 2] = s.g_base[3 * a + 2] - sz;
}

hipError_t solve_regradient(const SolveView& s, SolveState* state, hipStream_t st) {
    regradient_kernel<<<(s.D + 255) / 256, 256, 0, st>>>(s, state);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// PCG across many workgroups, for plans with more than 2048 nodes (above 8192 the single-workgroup kernels cannot
// hold p in one CU's LDS at all; between 2048 and 8192 they spend ~1 ms per launch sorting and repacking the matrix).
// Textbook preconditioned CG, same stopping rules as the kernels above, two launches per iteration — kernel
// boundaries are the grid barriers (a software barrier over 512 workgroups costs 9 - 41 us on this part, a boundary
// ~4 us: tools/microbench_gridbarrier.hip):
//   A(it)  beta from the partial r.z sums of the two previous updates; q = A p with p = z + beta p_old formed in the
//          gather (16 lanes per row over the slot-major ELL as assembled, no repacking); partial p.q
//   B(it)  alpha; x += alpha p; r -= alpha q; z = M^-1 r; partial r.z
// (The one-launch Chronopoulos-Gear form used for the 6x6-block system in solve6.hip was tried here first: in fp32 its
// recurrences stall just above the 1e-6 residual target and it took ~30 iterations where this form takes ~11.)
// Vector roles in the plan's mb_* buffers: x, r, z = mb_u[0], q = mb_w, p ping-pong = mb_p / mb_s.
constexpr int MB_LPR = 16;  // lanes per row

__device__ __forceinline__ float sum_partials_mb(const float* __restrict__ part, int n) {
    float acc = 0.f;
    for (int i = threadIdx.x & 63; i < n; i += 64) acc += part[i];
    return wave_sum_all(acc);  // the same value, the same order, in every wave
}

__global__ __launch_bounds__(256) void pcg_mb_init_kernel(SolveView s, SolveState* __restrict__ st) {
    __shared__ float sh[4];
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a == 0) {
        st->mb_done = st->done || st->converged ? 1 : 0, st->mb_skip = st->done ? 1 : 0, st->mb_iters = 0, st->mb_rz0 = 0.f;
        if (st->converged) st->gn_noop += 1;  // (the finish kernel books the iteration itself)
    }
    float rz = 0.f;
    if (a < s.D) {
        const float d    = s.diag[a];
        const float minv = d > FLT_EPSILON ? 1.0f / d : 1.0f;
        const float4 r   = make_float4(s.g[3 * a], s.g[3 * a + 1], s.g[3 * a + 2], 0.f);
        const float4 z   = make_float4(minv * r.x, minv * r.y, minv * r.z, 0.f);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        s.mb_r[a] = r, s.mb_u[0][a] = z, s.mb_x[a] = zero;
        s.mb_p[a] = s.mb_s[a] = s.mb_t[0][a] = s.mb_t[1][a] = zero;  // (the one-launch form multiplies them by beta_0 = 0)
        rz = fmaf(r.z, z.z, fmaf(r.y, z.y, r.x * z.x));
    }
    rz = wave_sum_all(rz);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = rz;
    __syncthreads();
    if (threadIdx.x == 0) s.mb_gpart[0][blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#ifdef DFA_DEV_AB  // the textbook form (two launches per iteration, DFA_MB_FORM=2): development builds only
__global__ __launch_bounds__(256) void pcg_mb_matvec_kernel(SolveView s, SolveState* __restrict__ st, int it, float pcg_tol) {
    __shared__ float sh[4];
    if (st->mb_done) return;
    const int nbu = (s.D + 255) / 256;  // workgroups of the init / update kernels
    const float rz_cur = sum_partials_mb(s.mb_gpart[it & 1], nbu);
    const float floor_ = 1e-12f;
    const float tol2   = pcg_tol * pcg_tol > floor_ ? pcg_tol * pcg_tol : floor_;
    float beta = 0.f;
    bool stop  = !(rz_cur > 0.f);
    if (it == 0) {
        const bool at_floor = st->grad_first > 0.0 && (double)rz_cur <= (double)floor_ * st->grad_first;
        stop                = stop || at_floor;  // nothing left to solve
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->mb_rz0 = rz_cur;
            if (at_floor) solve_mark_at_floor(st);
        }
    } else {
        const float rz_prev = sum_partials_mb(s.mb_gpart[(it + 1) & 1], nbu);
        beta                = rz_cur / rz_prev;
        stop                = stop || rz_cur <= fmaxf(tol2 * st->mb_rz0, solve_floor(st));
    }
    if (stop) {  // the same decision in every workgroup
        if (blockIdx.x == 0 && threadIdx.x == 0) st->mb_done = 1;
        return;
    }
    const int lane16 = threadIdx.x & (MB_LPR - 1);
    const int a      = (blockIdx.x * 256 + threadIdx.x) / MB_LPR;
    const float4* z    = s.mb_u[0];
    const float4* pold = (it & 1) ? s.mb_p : s.mb_s;  // written by launch it - 1
    float4* pnew       = (it & 1) ? s.mb_s : s.mb_p;
    float ax = 0.f, ay = 0.f, az = 0.f;
    const bool row_ok = a < s.D;
    const int cnt     = row_ok ? s.ell_cnt[a] : 0;
    for (int q = lane16; q < cnt; q += MB_LPR) {
        const float2 e  = s.ell[(size_t)q * s.D + a];
        const int col   = __float_as_int(e.y);
        const float val = e.x;
        float4 pv       = z[col];
        if (it > 0) {
            const float4 po = pold[col];
            pv.x = fmaf(beta, po.x, pv.x), pv.y = fmaf(beta, po.y, pv.y), pv.z = fmaf(beta, po.z, pv.z);
        }
        ax = fmaf(val, pv.x, ax), ay = fmaf(val, pv.y, ay), az = fmaf(val, pv.z, az);
    }
    ax = group16_sum(ax), ay = group16_sum(ay), az = group16_sum(az);
    float pq = 0.f;
    if (row_ok && lane16 == 0) {
        float4 pv = z[a];
        if (it > 0) {
            const float4 po = pold[a];
            pv.x = fmaf(beta, po.x, pv.x), pv.y = fmaf(beta, po.y, pv.y), pv.z = fmaf(beta, po.z, pv.z);
        }
        pnew[a]   = pv;
        s.mb_w[a] = make_float4(ax, ay, az, 0.f);
        pq        = fmaf(pv.z, az, fmaf(pv.y, ay, pv.x * ax));
    }
    pq = wave_sum_all(pq);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = pq;
    __syncthreads();
    if (threadIdx.x == 0) s.mb_dpart[0][blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void pcg_mb_update_kernel(SolveView s, SolveState* __restrict__ st, int it) {
    __shared__ float sh[4];
    if (st->mb_done) return;
    const float rz_cur = sum_partials_mb(s.mb_gpart[it & 1], (s.D + 255) / 256);
    const float pq     = sum_partials_mb(s.mb_dpart[0], solve_mb_blocks(s.D));
    if (!(pq > 0.f)) {  // breakdown: keep x
        if (blockIdx.x == 0 && threadIdx.x == 0) st->mb_done = 1;
        return;
    }
    const float alpha = rz_cur / pq;
    const int a       = blockIdx.x * blockDim.x + threadIdx.x;
    float rz          = 0.f;
    if (a < s.D) {
        const float4 p = ((it & 1) ? s.mb_s : s.mb_p)[a], q = s.mb_w[a];
        float4 x = s.mb_x[a], r = s.mb_r[a];
        x.x = fmaf(alpha, p.x, x.x), x.y = fmaf(alpha, p.y, x.y), x.z = fmaf(alpha, p.z, x.z);
        r.x = fmaf(-alpha, q.x, r.x), r.y = fmaf(-alpha, q.y, r.y), r.z = fmaf(-alpha, q.z, r.z);
        const float d    = s.diag[a];
        const float minv = d > FLT_EPSILON ? 1.0f / d : 1.0f;
        const float4 z   = make_float4(minv * r.x, minv * r.y, minv * r.z, 0.f);
        s.mb_x[a] = x, s.mb_r[a] = r, s.mb_u[0][a] = z;
        rz = fmaf(r.z, z.z, fmaf(r.y, z.y, r.x * z.x));
    }
    rz = wave_sum_all(rz);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = rz;
    __syncthreads();
    if (threadIdx.x == 0) {
        s.mb_gpart[(it + 1) & 1][blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        if (blockIdx.x == 0) st->mb_iters += 1;
    }
}
#endif  // DFA_DEV_AB

// ---- Chronopoulos-Gear form: ONE launch per iteration (the two inner products are taken together after the matrix
// product, so an iteration needs one grid-wide synchronisation; the textbook form above needs two).  As in
// s6_pcg_step_kernel:  u = M^-1 r, w = A u, m = M^-1 w, t kept by t_i = m_i + beta_i t_(i-1);
//   p_i = u_i + beta_i p_(i-1);  s_i = w_i + beta_i s_(i-1);  x += alpha_i p_i;  r -= alpha_i s_i;  u_(i+1) = u_i - alpha_i t_i;
//   w_(i+1) = A u_(i+1) = A u_i - alpha_i (A m_i + beta_i A t_(i-1))  — gathered from the vectors of launch i - 1;
//   gamma = (r, u), delta = (w, u);  beta_(i+1) = gamma_(i+1) / gamma_i;  alpha_(i+1) = gamma_(i+1) / (delta_(i+1) - beta_(i+1) gamma_(i+1) / alpha_i).
// launch it = -1: w_0 = A u_0, m_0, gamma_0, delta_0 (u_0 = M^-1 g, x = 0 from pcg_mb_init_kernel).  16 lanes per row.
// Same iterates as the textbook form in exact arithmetic; the stopping rules are evaluated on gamma = (r, M^-1 r).
__global__ __launch_bounds__(256) void pcg_mb_step_kernel(SolveView s, SolveState* __restrict__ st, int it, float pcg_tol) {
    __shared__ float sh[2][4];
    __shared__ float scal[2];
    const int nb  = solve_mb_blocks(s.D);
    const int cur = it >= 0 ? (it & 1) : 0, nxt = cur ^ 1;  // u, m: read [cur], write [nxt]; t: read [nxt], write [cur]
    const float4* ucur  = s.mb_u[cur];
    const float4* mcur  = s.mb_m[cur];
    const float4* tprev = s.mb_t[nxt];
    // The launch is a chain of dependent round trips and little else.  The row's length and each lane's FIRST matrix entry
    // depend on nothing but the launch arguments: they are requested here, with the stop flag and the partial inner products,
    // and their gathers go out before the scalars are summed — two rounds instead of three (scalars, then entries, then
    // gathers), also for the second entry of a lane (rows of 17-32 entries: every k = 8 row).  A launch that turns out to have nothing to do has loaded a few values in vain.
    const int lane16  = threadIdx.x & (MB_LPR - 1);
    const int a       = (blockIdx.x * 256 + threadIdx.x) / MB_LPR;
    const bool row_ok = a < s.D;
    const int cnt     = row_ok ? s.ell_cnt[a] : 0;
    const float2 e0   = row_ok ? s.ell[(size_t)lane16 * s.D + a] : make_float2(0.f, 0.f);  // (rows shorter than 16: not used)
    const float2 e1   = row_ok ? s.ell[(size_t)(lane16 + MB_LPR) * s.D + a] : make_float2(0.f, 0.f);  // (k = 8 rows have ~27 entries)
    if (st->mb_done) return;
    const bool has0 = lane16 < cnt, has1 = lane16 + MB_LPR < cnt;
    const int col0  = has0 ? __float_as_int(e0.y) : 0, col1 = has1 ? __float_as_int(e1.y) : 0;
    const float4 uu0 = ucur[col0], uu1 = ucur[col1];
    float4 mm0 = make_float4(0.f, 0.f, 0.f, 0.f), tt0 = mm0, mm1 = mm0, tt1 = mm0;
    if (it >= 0) mm0 = mcur[col0], tt0 = tprev[col0], mm1 = mcur[col1], tt1 = tprev[col1];
    float alpha = 0.f, beta = 0.f;
    if (it >= 0) {
        // the inner products of the launch before: every workgroup adds the partials in the same order
        float g = 0.f, d = 0.f;
        for (int i = threadIdx.x; i < nb; i += 256) g += s.mb_gpart[it & 1][i], d += s.mb_dpart[it & 1][i];
        g = wave_sum_all(g), d = wave_sum_all(d);
        if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = g, sh[1][threadIdx.x >> 6] = d;
        __syncthreads();
        const float gamma = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]), delta = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
        const float floor_ = 1e-12f;
        const float tol2   = pcg_tol * pcg_tol > floor_ ? pcg_tol * pcg_tol : floor_;
        float rz0 = gamma, denom = delta;
        bool stop = !(gamma > 0.f);
        if (it == 0) {
            const bool at_floor = st->grad_first > 0.0 && (double)gamma <= (double)floor_ * st->grad_first;
            stop                = stop || at_floor;  // nothing left to solve
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                st->mb_rz0 = gamma;
                if (at_floor) solve_mark_at_floor(st);
            }
        } else {
            rz0   = st->mb_rz0;
            beta  = gamma / st->mb_gamma_prev[(it + 1) & 1];
            denom = delta - beta * gamma / st->mb_alpha_prev[(it + 1) & 1];
            stop  = stop || gamma <= fmaxf(tol2 * rz0, solve_floor(st));
        }
        stop = stop || !(denom > 0.f);  // converged, or breakdown: the same decision in every workgroup
        if (stop) {
            if (blockIdx.x == 0 && threadIdx.x == 0) st->mb_done = 1;
            return;
        }
        alpha = gamma / denom;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->mb_gamma_prev[it & 1] = gamma, st->mb_alpha_prev[it & 1] = alpha;
            st->mb_iters += 1;
        }
    }
    float au[3] = {0.f, 0.f, 0.f}, am[3] = {0.f, 0.f, 0.f}, at[3] = {0.f, 0.f, 0.f};
    for (int q = lane16; q < cnt; q += MB_LPR) {
        const bool first = q == lane16, second = q == lane16 + MB_LPR;
        const float2 e   = first ? e0 : second ? e1 : s.ell[(size_t)q * s.D + a];
        const int col    = __float_as_int(e.y);
        const float val  = e.x;
        const float4 uu  = first ? uu0 : second ? uu1 : ucur[col];
        au[0] = fmaf(val, uu.x, au[0]), au[1] = fmaf(val, uu.y, au[1]), au[2] = fmaf(val, uu.z, au[2]);
        if (it >= 0) {
            const float4 mm = first ? mm0 : second ? mm1 : mcur[col], tt = first ? tt0 : second ? tt1 : tprev[col];
            am[0] = fmaf(val, mm.x, am[0]), am[1] = fmaf(val, mm.y, am[1]), am[2] = fmaf(val, mm.z, am[2]);
            at[0] = fmaf(val, tt.x, at[0]), at[1] = fmaf(val, tt.y, at[1]), at[2] = fmaf(val, tt.z, at[2]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) au[c] = group16_sum(au[c]), am[c] = group16_sum(am[c]), at[c] = group16_sum(at[c]);
    float gpart = 0.f, dpart = 0.f;
    if (row_ok && lane16 == 0) {
        const float dg   = s.diag[a];
        const float minv = dg > FLT_EPSILON ? 1.0f / dg : 1.0f;
        float4 u = ucur[a], r = s.mb_r[a];
        float w[3] = {au[0], au[1], au[2]};
        if (it >= 0) {
            const float4 m = mcur[a], tp = tprev[a], p = s.mb_p[a], sv = s.mb_s[a], wv = s.mb_w[a];
            float4 x = s.mb_x[a];
            const float4 tn = make_float4(fmaf(beta, tp.x, m.x), fmaf(beta, tp.y, m.y), fmaf(beta, tp.z, m.z), 0.f);
            const float4 pn = make_float4(fmaf(beta, p.x, u.x), fmaf(beta, p.y, u.y), fmaf(beta, p.z, u.z), 0.f);
            const float4 sn = make_float4(fmaf(beta, sv.x, wv.x), fmaf(beta, sv.y, wv.y), fmaf(beta, sv.z, wv.z), 0.f);
            u = make_float4(fmaf(-alpha, tn.x, u.x), fmaf(-alpha, tn.y, u.y), fmaf(-alpha, tn.z, u.z), 0.f);
            r = make_float4(fmaf(-alpha, sn.x, r.x), fmaf(-alpha, sn.y, r.y), fmaf(-alpha, sn.z, r.z), 0.f);
            x.x = fmaf(alpha, pn.x, x.x), x.y = fmaf(alpha, pn.y, x.y), x.z = fmaf(alpha, pn.z, x.z);
#pragma unroll
            for (int c = 0; c < 3; ++c) w[c] = au[c] - alpha * (am[c] + beta * at[c]);
            s.mb_t[cur][a] = tn, s.mb_p[a] = pn, s.mb_s[a] = sn, s.mb_x[a] = x, s.mb_r[a] = r, s.mb_u[nxt][a] = u;
        }
        s.mb_w[a]                     = make_float4(w[0], w[1], w[2], 0.f);
        s.mb_m[it >= 0 ? nxt : 0][a] = make_float4(minv * w[0], minv * w[1], minv * w[2], 0.f);
        gpart = fmaf(r.z, u.z, fmaf(r.y, u.y, r.x * u.x));
        dpart = fmaf(w[2], u.z, fmaf(w[1], u.y, w[0] * u.x));
    }
    gpart = wave_sum_all(gpart), dpart = wave_sum_all(dpart);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = gpart, sh[1][threadIdx.x >> 6] = dpart;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int slot = it >= 0 ? ((it + 1) & 1) : 0;
        s.mb_gpart[slot][blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        s.mb_dpart[slot][blockIdx.x] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
    (void)scal;
}

// t += delta and the counters the single-workgroup kernels keep
__global__ __launch_bounds__(256) void pcg_mb_finish_kernel(SolveView s, SolveState* __restrict__ st) {
    if (st->mb_skip) return;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < s.D) {
        const float4 x = s.mb_x[a];
        s.t[3 * a] += x.x, s.t[3 * a + 1] += x.y, s.t[3 * a + 2] += x.z;
    }
    if (a == 0) {
        if (st->grad_first == 0.0) st->grad_first = (double)st->mb_rz0;
        st->pcg_iters += st->mb_iters;
        st->gn_iters += 1;
    }
}

// The iteration count is only known on the device.  With a pinned host word the launches go out in chunks (16, 32,
// 64, ...) and the stop flag is read back between chunks — one stream synchronisation per chunk instead of up to
// max_iter launches that return at once (3 us each: 0.7 ms per Gauss-Newton iteration at the usual cap of 256).
void MbGraphCache::release() {
    for (int i = 0; i < used; ++i)
        if (e[i].exec) (void)hipGraphExecDestroy(e[i].exec);
    used = 0;
    if (capture) (void)hipStreamDestroy(capture), capture = nullptr;
}

// development builds: DFA_MB_FORM=2 is the textbook form, two launches per iteration
static bool mb_one_launch() { return dev_env_int("DFA_MB_FORM", 1) != 2; }

// iterations [it0, it1): from the cache's graph of that range when there is (or can be) one, else launch by launch
static hipError_t launch_mb_range(const SolveView& s, SolveState* state, int it0, int it1, float pcg_tol,
                                  MbGraphCache* gc, hipStream_t st) {
    const int nb = solve_mb_blocks(s.D);
#ifdef DFA_DEV_AB
    const int nbu = (s.D + 255) / 256;
#endif
    auto direct = [&](hipStream_t q) {
        for (int it = it0; it < it1; ++it) {
#ifdef DFA_DEV_AB
            if (!mb_one_launch()) {
                pcg_mb_matvec_kernel<<<nb, 256, 0, q>>>(s, state, it, pcg_tol);
                pcg_mb_update_kernel<<<nbu, 256, 0, q>>>(s, state, it);
                continue;
            }
#endif
            pcg_mb_step_kernel<<<nb, 256, 0, q>>>(s, state, it, pcg_tol);
        }
        return hipGetLastError();
    };
    const bool no_graph = dev_env("DFA_MB_NO_GRAPH") != nullptr;  // A/B (development builds)
    if (!gc || gc->disabled || no_graph || it1 - it0 < 4) return direct(st);
    MbGraphCache::Entry* hit = nullptr;
    for (int i = 0; i < gc->used && !hit; ++i) {
        MbGraphCache::Entry& c = gc->e[i];
        // (the plan's own buffers never move; the borrowed pointers of the view change from frame to frame but these
        // kernels read none of them)
        if (c.it0 == it0 && c.it1 == it1 && c.tol == pcg_tol && c.state == state && c.view.D == s.D && c.view.ell == s.ell)
            hit = &c;
    }
    if (!hit) {
        if (gc->used == (int)(sizeof(gc->e) / sizeof(gc->e[0]))) {  // other problem sizes / chunk sizes: start over
            for (int i = 0; i < gc->used; ++i)
                if (gc->e[i].exec) (void)hipGraphExecDestroy(gc->e[i].exec);
            gc->used = 0;
        }
        hipGraph_t g = nullptr;
        hipGraphExec_t exec = nullptr;
        bool ok = gc->capture || hipStreamCreateWithFlags(&gc->capture, hipStreamNonBlocking) == hipSuccess;
        ok      = ok && hipStreamBeginCapture(gc->capture, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const hipError_t le = direct(gc->capture);
            const hipError_t ce = hipStreamEndCapture(gc->capture, &g);
            ok = le == hipSuccess && ce == hipSuccess && g && hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess;
            if (g) (void)hipGraphDestroy(g);
        }
        if (!ok) {
            (void)hipGetLastError();  // clear the sticky error of the failed attempt
            gc->disabled = true;
            return direct(st);
        }
        hit        = &gc->e[gc->used++];
        hit->it0 = it0, hit->it1 = it1, hit->tol = pcg_tol, hit->view = s, hit->state = state, hit->exec = exec;
    }
    return hipGraphLaunch(hit->exec, st);
}

static hipError_t launch_mb_pcg(const SolveView& s, SolveState* state, int max_iter, float pcg_tol, int* host_flag,
                                MbGraphCache* gc, hipStream_t st) {
    const int nb = solve_mb_blocks(s.D), nbu = (s.D + 255) / 256;
    pcg_mb_init_kernel<<<nbu, 256, 0, st>>>(s, state);
    if (mb_one_launch()) pcg_mb_step_kernel<<<nb, 256, 0, st>>>(s, state, -1, pcg_tol);
    int chunk = 16;
    const int ci = gc ? std::min(gc->call++, 63) : 0;
    if (mb_one_launch() && gc && host_flag && gc->pred[ci] > 0) chunk = std::max(8, (gc->pred[ci] + 4 + 7) & ~7);
    for (int it = 0; it < max_iter;) {
        const int end = host_flag ? std::min(max_iter, it + chunk) : max_iter;
        {
            const hipError_t e = launch_mb_range(s, state, it, end, pcg_tol, host_flag ? gc : nullptr, st);
            if (e != hipSuccess) return e;
            it = end;
        }
        if (mb_one_launch() && host_flag && it < max_iter) {
            // step `it` first evaluates the stopping rule on the residual the chunk left, then iterates
            pcg_mb_step_kernel<<<nb, 256, 0, st>>>(s, state, it, pcg_tol);
            // mb_done, converged, mb_skip, mb_iters
            hipError_t e = hipMemcpyAsync(host_flag, &state->mb_done, 4 * sizeof(int), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            if (gc) gc->pred[ci] = host_flag[3];  // (the total when the flag is set, a lower bound otherwise)
            if (*host_flag) break;
            ++it;
            chunk = 16;  // the prediction fell short: go on in small chunks
            continue;
        }
#ifdef DFA_DEV_AB
        if (host_flag && it < max_iter) {
            // one more matvec launch evaluates the stopping rule on the last update's residual
            pcg_mb_matvec_kernel<<<nb, 256, 0, st>>>(s, state, it, pcg_tol);
            // mb_done and, directly behind it, converged (the caller stops launching Gauss-Newton iterations on it)
            hipError_t e = hipMemcpyAsync(host_flag, &state->mb_done, 2 * sizeof(int), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            if (*host_flag) break;
            pcg_mb_update_kernel<<<nbu, 256, 0, st>>>(s, state, it);  // the matvec above was iteration `it`
            ++it;
            chunk *= 2;
        }
#endif
    }
    pcg_mb_finish_kernel<<<nbu, 256, 0, st>>>(s, state);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// PCG by three TEAMS of persistent workgroups, one coordinate per team, every team confined to ONE XCD
// (plans of 2 049 .. ~19 000 nodes: C3, C4, the adaptor's frames).
//
// The launched form above pays a kernel boundary per iteration: ~5.8 us for an iteration whose arithmetic takes a fraction of
// a microsecond, and the host has to guess how many launches to enqueue (192 launches for 105 iterations per C3 frame).
// A grid barrier across the chip costs more than the boundary (L2 write-back + invalidate between XCDs: ~15 us); a barrier
// among workgroups that share ONE L2 does not (tools/microbench_xcd_barrier.hip).  J^T J = A (x) I_3 is three independent
// scalar systems with the same matrix (as the register-resident kernel above solves them): coordinate c is solved by the
// TEAM_W workgroups that the dispatcher placed on XCD c, one per CU (the launch asks for more than half a CU's LDS) — every
// workgroup reads its XCC_ID, takes a rank in its team by an atomic counter and leaves if the team is full or the XCD is
// not 0..2: the placement is counted, never assumed.
//
// One synchronisation per iteration (Chronopoulos-Gear form, the recurrences of pcg_mb_step_kernel), one gathered vector
// pair: a member owns R = ceil(D / TEAM_W) rows; J = TEAM_NT / R threads share a row and keep their entries' values,
// columns and the REPLICA of u at the entry's column in registers for the whole solve:
//   wait(round i)            gamma_i, delta_i = sums of the members' partials (carried by the flag words themselves)
//   beta_i, alpha_i          the same bits in every member (same words, same summation order)
//   LDS <- (m_i, t_(i-1))    the pair every row owner published before it raised its flag: D x 8 bytes, coalesced, from L2
//   per entry                t_i[col] = m_i[col] + beta_i t_(i-1)[col];  u_(i+1)[col] = u_i[col] - alpha_i t_i[col]  (replica: the
//                            owner of row col does the same arithmetic on the same numbers);  w_(i+1)[a] += val u_(i+1)[col]
//   row owner                p, s, x, r, t, u as in pcg_mb_step_kernel;  m_(i+1) = M^-1 w_(i+1);  partial (r, u), (w, u)
//   publish                  (m_(i+1), t_i) of the own rows, s_waitcnt vmcnt(0), workgroup barrier, then the member's flag
//                            words {round i + 1, partial}
// Where the longest row of the matrix fits TEAM_E_TREG slots per thread, t's replica at the entry's column lives in a register
// as well and m ALONE is exchanged (half the copy, 4-byte gathers): team_member<16, true>; longer rows: team_member<20, false>.
//
// How the exchange stays inside the XCD's L2.  Agent-scope atomics (sc1) are the textbook tool and were the first version:
// every such load is a trip over the fabric (1.2-1.5 us measured here; 2 MB of them per team and iteration for the vector
// copy) and an iteration cost 6.7 us — no better than a launch.  An agent-scope acquire fence + plain loads: buffer_inv sc1
// from 1 500 waves, 30 us per iteration.  `buffer_inv sc0` + plain loads: leaves the vector L1 alone outside threadgroup-split
// mode — the pollers never saw a flag, the teams timed out and the guard launch took over (which is how that was found).
// What works: PLAIN stores and PLAIN loads, with NO ADDRESS READ TWICE by a CU inside a launch.  A plain store is in the
// XCD's L2 once acknowledged (the vector L1 writes through and does not allocate on stores); a plain load of an address this CU
// has not read since the kernel began (the L1 starts a kernel empty; one workgroup per CU: nobody else fills it) misses the
// L1 and is served by that same L2.  So every barrier round of a launch has an exchange area of its own, and a flag word
// is stored TEAM_K times: poll attempt k reads copy k, a fresh line, and only a wait that outlasts TEAM_K attempts goes on
// with agent-scope loads.  Flag words are self-validating ({round, value} in one 64-bit store; rounds grow from launch to
// launch, so nothing is ever reset), and the vectors are complete when the flag is stored because every wave has waited for
// its stores' acknowledgements (vmcnt(0): stores count in vmcnt on gfx9) before the workgroup barrier in front of it.
// What makes this enough is that writer and reader share the L2 — which the XCC_ID census guarantees and nothing else does.
// Every spin is bounded by the wall clock (s_memrealtime): a team that cannot assemble (placement, starvation by other
// kernels) or a row that does not fit the register slots ABORTS before it has changed anything, and the guard launch behind
// (pcg_team_guard_kernel: one workgroup per coordinate, returns at entry otherwise) solves every coordinate nobody has dealt
// with — a team that gave up, a team no workgroup ever joined — by itself; the host sees the count in pinned memory at its
// next call and goes back to the launched form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "pcg_team_kernel orders its stores with s_waitcnt vmcnt(0): gfx9 only (gfx10+ count stores in vscnt)"
#endif
constexpr int TEAM_W = 32;        // members per team: ONE 1024-thread workgroup per CU of a 32-CU XCD
constexpr int TEAM_NT = 1024;
constexpr int TEAM_E = 20;        // register slots per thread (rows of up to J x 20 entries) ...
constexpr int TEAM_E_TREG = 16;   // ... and of the form that keeps t's replica in registers (rows of up to J x 16)
constexpr int TEAM_K = 4;         // copies of a flag word = poll attempts served by plain loads
constexpr int TEAM_ROUNDS = 257;  // exchange areas per launch: barrier rounds 0 .. 256 (the reference's linearIter, dyn_fusion.cpp:186)
constexpr size_t TEAM_MIN_LDS = 82 * 1024;  // more than half a CU's LDS: one member per CU, nobody else fills its L1
constexpr long long TEAM_TICKS_FIRST = 2000000, TEAM_TICKS = 500000;  // 20 ms / 5 ms of the 100 MHz wall clock
// flag words of one team: [round][copy][kind: (gamma | delta), joint][2 x TEAM_W]
__host__ __device__ constexpr size_t team_words_per_round() { return (size_t)TEAM_K * 2 * 2 * TEAM_W; }
size_t solve_team_pcg_words() { return 3 * (size_t)TEAM_ROUNDS * team_words_per_round(); }

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }  // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ void team_give_up(TeamCtl* ctl, int c, int* host_abort) {
    if (atomicExch(&ctl->abort[c], 1u) == 0u && host_abort) atomicAdd_system(host_abort, 1);
}

// The first wave of a member polls the words of `round`: lane l reads member l & 31's gamma (l < 32) or delta word; with
// JOINT the lanes below 32 also read the joint (r0, z0) word.  Sums in lane order by the same butterfly in every member; the
// workgroup meets at a barrier behind it.  Called by every thread; false = timed out / the team has given up.  The wall clock
// (s_memrealtime: a microsecond by itself) is only consulted once a wait has lasted 64 polls.
template <bool JOINT>
__device__ __forceinline__ bool team_wait(const unsigned long long* __restrict__ wr /* this round's words */, unsigned round,
                                          float (&sum)[3], const unsigned* abort_flag, long long ticks, float* bc /* LDS [4] */,
                                          long long* prof = nullptr, int rank = 0) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        long long t0 = 0;
        unsigned long long w0 = 0, w1 = 0;
        bool good = true;
#ifdef DFA_PCG_PROFILE
        const long long c0_ = clock64();
        bool own_seen = false;
#endif
        for (unsigned spins = 0;; ++spins) {
            // attempt k < TEAM_K: copy k by a plain load (a line this CU has never read: from the L2); later: copy 0, agent scope
            const unsigned long long* p = wr + (size_t)(spins < (unsigned)TEAM_K ? spins : 0u) * (4 * TEAM_W) + lane;
            if (spins < (unsigned)TEAM_K) {
                // (wavefront scope = no cache-policy bits on the load; `volatile` would make it a system-scope one)
                w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (JOINT) w1 = __hip_atomic_load(p + 2 * TEAM_W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else {
                w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (JOINT) w1 = __hip_atomic_load(p + 2 * TEAM_W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const bool ok = (unsigned)(w0 >> 32) == round && (!JOINT || lane >= TEAM_W || (unsigned)(w1 >> 32) == round);
#ifdef DFA_PCG_PROFILE
            if (prof) {  // how long until this member's OWN words come back (store -> L2 -> load: no skew in it), and the polls
                const bool mine_ok = __builtin_amdgcn_readlane((int)ok, rank);
                if (mine_ok && !own_seen) own_seen = true, prof[0] += clock64() - c0_;
                prof[1] += 1;
            }
#endif
            if (__all((int)ok)) break;
            if (spins >= 64u && (spins & 63u) == 0u) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > ticks || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    good = false;
                    break;
                }
            }
            if (spins >= (unsigned)TEAM_K) __builtin_amdgcn_s_sleep(1);
        }
        float v0 = __uint_as_float((unsigned)w0), v1 = JOINT && lane < TEAM_W ? __uint_as_float((unsigned)w1) : 0.f;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v0 += __shfl_xor(v0, o, 64), v1 += __shfl_xor(v1, o, 64);  // sums of each half of the wave
        if (lane == 0) bc[0] = v0, bc[2] = v1, bc[3] = good ? 1.f : 0.f;
        if (lane == TEAM_W) bc[1] = v0;
    }
    __syncthreads();
    sum[0] = bc[0], sum[1] = bc[1], sum[2] = bc[2];
    return bc[3] != 0.f;
}

// a member's solve: E register slots per thread; TREG: the replica of t at the entry's column lives in a register too, and the
// exchange carries m alone (4 bytes per row instead of the (m, t) pair: half the copy, 4-byte LDS gathers) — the form for
// plans whose longest row fits 16 slots per thread
template <int E, bool TREG>
__device__ __forceinline__ void team_member(const SolveView& s, SolveState* __restrict__ st, TeamCtl* ctl, char* smem, float (&red)[3][TEAM_NT / 64],
                                            float* bc, int c, int rank, unsigned epoch0, int max_iter, float pcg_tol, int* host_abort) {
    const int tid = threadIdx.x, D = s.D;
    float2* mt_s = (float2*)smem;                                      // Dpad x (m, t) ...
    float* m_s   = (float*)smem;                                       // ... TREG: Dpad x m
    float* part  = (float*)(smem + sizeof(float2) * (size_t)s.Dpad);  // TEAM_NT partial row sums
    const int R = (D + TEAM_W - 1) / TEAM_W, J = TEAM_NT / R;          // rows per member, threads per row
    const int r0 = rank * R, nrows = max(0, min(R, D - r0));
    const int a_loc = tid % R, j = tid / R;
    const bool active = j < J && a_loc < nrows, owner = active && j == 0;
    const int a = active ? r0 + a_loc : 0;  // (a valid row for the unconditional loads of the others)

    // ---- this thread's entries -> registers: slot e holds entry q = j + e J of row a
    const int cnt  = active ? min(s.ell_cnt[a], s.ell_cap) : 0;
    const int mine = cnt > j ? (cnt - j + J - 1) / J : 0;
    if (__syncthreads_or(mine > E)) {  // a row that does not fit: before anything has been published
        if (tid == 0) team_give_up(ctl, c, host_abort);
        return;
    }
    int emax = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) emax = max(emax, __shfl_xor(emax, o, 64));
    emax = __builtin_amdgcn_readfirstlane(emax);  // wave-uniform slot bound
    // (batches of unconditional loads, masked after: a load under a per-slot condition makes hipcc wait for each one by
    // itself — 48 dependent round trips, 35 us, in the first version of this prologue; in two halves: the unpacked columns of
    // all E slots at once cost registers the loop needs)
    float val[E], ucol[E], tcol[TREG ? E : 1];
    uint32_t colp[E / 2];
#pragma unroll
    for (int e = 0; e < (TREG ? E : 1); ++e) tcol[e] = 0.f;
#pragma unroll
    for (int h0 = 0; h0 < E; h0 += E / 2) {
        int col[E / 2];
#pragma unroll
        for (int i = 0; i < E / 2; ++i) {
            const int e     = h0 + i;
            const float2 en = s.ell[(size_t)min(j + e * J, s.ell_cap - 1) * D + a];
            const bool live = e < mine;
            val[e] = live ? en.x : 0.f;
            col[i] = live ? __float_as_int(en.y) : a;
        }
        float gcol[E / 2];
#pragma unroll
        for (int i = 0; i < E / 2; ++i) ucol[h0 + i] = s.diag[col[i]], gcol[i] = s.g[3 * col[i] + c];  // (both in one round trip)
#pragma unroll
        for (int i = 0; i < E / 2; ++i) ucol[h0 + i] = (ucol[h0 + i] > FLT_EPSILON ? 1.0f / ucol[h0 + i] : 1.0f) * gcol[i];
#pragma unroll
        for (int i = 0; i < E / 2; i += 2) colp[(h0 + i) / 2] = (uint32_t)col[i] | ((uint32_t)col[i + 1] << 16);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- row owners: r = g, u = M^-1 g, x = p = s = t = 0; (r0, z0) of the JOINT system scales the stopping rules
    float minv = 0.f, r = 0.f, u = 0.f, x = 0.f, pv = 0.f, sv = 0.f, tv = 0.f, w = 0.f, m = 0.f, joint_loc = 0.f;
    if (owner) {
        const float d = s.diag[a];
        minv          = d > FLT_EPSILON ? 1.0f / d : 1.0f;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const float g = s.g[3 * a + cc];
            joint_loc     = fmaf(g, minv * g, joint_loc);
            if (cc == c) r = g;
        }
        u = minv * r;
    }
    // exchange area / flag words of barrier round r of THIS launch (r = 0 .. max_iter): never read twice by a CU
    auto mt_at = [&](int rr) __attribute__((always_inline)) { return s.team_mt + ((size_t)rr * 3 + c) * s.team_stride; };
    auto words_at = [&](int rr) __attribute__((always_inline)) {
        return s.team_words + ((size_t)c * TEAM_ROUNDS + rr) * team_words_per_round();
    };

    // w = A u over the replicas; the row's J partial sums meet in LDS (threads of a row are R apart: any R, any J)
    auto row_product = [&]() __attribute__((always_inline)) {
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int e = 0; e < E; e += 2)
            if (e < emax) acc0 = fmaf(val[e], ucol[e], acc0), acc1 = fmaf(val[e + 1], ucol[e + 1], acc1);  // (empty slots hold val = 0)
        part[tid] = acc0 + acc1;
        __syncthreads();
        float tot = 0.f;
        if (owner) {
            for (int j0 = 0; j0 < J; j0 += 8) {  // eight LDS reads in flight
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = part[a_loc + min(j0 + q, J - 1) * R];
#pragma unroll
                for (int q = 0; q < 8; ++q) tot += j0 + q < J ? v[q] : 0.f;
            }
        }
        return tot;
    };
    // (m, t) of the own rows and the member's partial sums for barrier round rr (tag `round`)
    auto publish = [&](int rr, unsigned round, float gp, float dp, float jp, bool with_joint) __attribute__((always_inline)) {
        // (a plain store: in the XCD's L2 once acknowledged)
        if (owner) {
            if (TREG) ((float*)mt_at(rr))[a] = m;
            else mt_at(rr)[a] = make_float2(m, tv);
        }
        const float gw = wave_total(gp), dw = wave_total(dp), jw = with_joint ? wave_total(jp) : 0.f;
        if ((tid & 63) == 0) red[0][tid >> 6] = gw, red[1][tid >> 6] = dw, red[2][tid >> 6] = jw;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have been acknowledged by the L2
        __syncthreads();
        // thread (copy k, kind q): q = 0 gamma, 1 delta, 2 joint — 64-bit stores, one per copy
        if (tid < 3 * TEAM_K) {
            const int q = tid % 3, kk = tid / 3;
            if (q < 2 || with_joint) {
                float tot = 0.f;
#pragma unroll
                for (int i = 0; i < TEAM_NT / 64; ++i) tot += red[q][i];
                unsigned long long* dst = words_at(rr) + (size_t)kk * (4 * TEAM_W) + (q == 2 ? 2 * TEAM_W : q * TEAM_W) + rank;
                __hip_atomic_store(dst, ((unsigned long long)round << 32) | __float_as_uint(tot), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WAVEFRONT);  // (one 64-bit store, no cache-policy bits)
            }
        }
    };

#ifdef DFA_PCG_PROFILE
    long long pc_[6] = {0, 0, 0, 0, 0, 0}, pw_[2] = {0, 0};
    long long last_  = clock64();
#endif
    w = row_product();
    m = minv * w;
    publish(0, epoch0, r * u, w * u, joint_loc, true);
    PROF_MARK(5);  // (prologue's product + first publication)

    const float floor_ = 1e-12f;
    const float tol2   = pcg_tol * pcg_tol > floor_ ? pcg_tol * pcg_tol : floor_;
    float target = 0.f, gamma_old = 1.f, alpha_old = 1.f, rz0 = 0.f;
    int it = 0;
    bool gave_up = false;
    while (it < max_iter) {
        float sm[3];
        const unsigned round = epoch0 + (unsigned)it;
        if (it == 0) {
            if (!team_wait<true>(words_at(0), round, sm, &ctl->abort[c], TEAM_TICKS_FIRST, bc)) { gave_up = true; break; }
            rz0 = sm[2];
            if (st->grad_first > 0.0 && (double)rz0 <= 1e-12 * st->grad_first) {  // the same decision in every team
                if (tid == 0 && rank == 0) {
                    ctl->handled[c] = 1u;
                    if (c == 0) {
                        st->gn_iters += 1;
                        solve_mark_at_floor(st);
                    }
                }
                return;
            }
            target = fmaxf(tol2 * rz0, solve_floor(st)) * (1.0f / 3.0f);  // this coordinate's share of the joint target
        } else {
#ifdef DFA_PCG_PROFILE
            if (!team_wait<false>(words_at(it), round, sm, &ctl->abort[c], TEAM_TICKS, bc, tid == 0 && c == 0 && rank == 0 ? pw_ : nullptr, rank)) { gave_up = true; break; }
#else
            if (!team_wait<false>(words_at(it), round, sm, &ctl->abort[c], TEAM_TICKS, bc)) { gave_up = true; break; }
#endif
        }
        const float gamma = sm[0], delta = sm[1];
        PROF_MARK(0);  // wait
        if (!(gamma > target)) break;  // converged: (r, M^-1 r) of the iterate in x
        const float beta  = it == 0 ? 0.f : gamma / gamma_old;
        const float denom = it == 0 ? delta : delta - beta * gamma / alpha_old;
        if (!(denom > 0.f)) break;
        const float alpha = gamma / denom;
        {   // the published (m_i, t_(i-1)) of every row -> LDS by plain wide loads (first and only read of round i's area by
            // this CU: from the L2)
            const float4* src = (const float4*)mt_at(it);
            float4* dst       = (float4*)mt_s;
            const int n4      = TREG ? s.Dpad / 4 : s.Dpad / 2;
            for (int i0 = tid; i0 < n4; i0 += 4 * TEAM_NT) {  // four loads in flight per thread
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = src[min(i0 + q * TEAM_NT, n4 - 1)];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i0 + q * TEAM_NT < n4) dst[i0 + q * TEAM_NT] = v[q];
            }
        }
        __syncthreads();
        PROF_MARK(1);  // copy
        // four gathers in flight per step (a step under its own wave-uniform branch would wait for each gather by itself)
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += 4)
            if (e0 < emax) {
                uint32_t c01 = colp[e0 / 2], c23 = colp[e0 / 2 + 1];
                asm volatile("" : "+v"(c01), "+v"(c23));
                if (TREG) {  // t's replica in a register: the same fmaf on the same numbers as the row's owner
                    const float g0 = m_s[c01 & 0xffffu], g1 = m_s[c01 >> 16], g2 = m_s[c23 & 0xffffu], g3 = m_s[c23 >> 16];
                    const int t0 = TREG ? e0 : 0;  // (tcol has one element in the other form: never indexed there)
                    tcol[t0]                  = fmaf(beta, tcol[t0], g0);
                    tcol[TREG ? e0 + 1 : 0]   = fmaf(beta, tcol[TREG ? e0 + 1 : 0], g1);
                    tcol[TREG ? e0 + 2 : 0]   = fmaf(beta, tcol[TREG ? e0 + 2 : 0], g2);
                    tcol[TREG ? e0 + 3 : 0]   = fmaf(beta, tcol[TREG ? e0 + 3 : 0], g3);
                    ucol[e0]     = fmaf(-alpha, tcol[t0], ucol[e0]);
                    ucol[e0 + 1] = fmaf(-alpha, tcol[TREG ? e0 + 1 : 0], ucol[e0 + 1]);
                    ucol[e0 + 2] = fmaf(-alpha, tcol[TREG ? e0 + 2 : 0], ucol[e0 + 2]);
                    ucol[e0 + 3] = fmaf(-alpha, tcol[TREG ? e0 + 3 : 0], ucol[e0 + 3]);
                } else {
                    const float2 g0 = mt_s[c01 & 0xffffu], g1 = mt_s[c01 >> 16], g2 = mt_s[c23 & 0xffffu], g3 = mt_s[c23 >> 16];
                    ucol[e0]     = fmaf(-alpha, fmaf(beta, g0.y, g0.x), ucol[e0]);
                    ucol[e0 + 1] = fmaf(-alpha, fmaf(beta, g1.y, g1.x), ucol[e0 + 1]);
                    ucol[e0 + 2] = fmaf(-alpha, fmaf(beta, g2.y, g2.x), ucol[e0 + 2]);
                    ucol[e0 + 3] = fmaf(-alpha, fmaf(beta, g3.y, g3.x), ucol[e0 + 3]);
                }
            }
        if (owner) {
            pv = fmaf(beta, pv, u), sv = fmaf(beta, sv, w);
            x  = fmaf(alpha, pv, x), r = fmaf(-alpha, sv, r);
            tv = fmaf(beta, tv, m);
            u  = fmaf(-alpha, tv, u);
        }
        PROF_MARK(2);  // replicas
        w = row_product();  // (its barrier also orders this iteration's reads of mt_s before the next copy)
        m = minv * w;
        ++it;
        gamma_old = gamma, alpha_old = alpha;
        PROF_MARK(3);  // row product
        publish(it, epoch0 + (unsigned)it, r * u, w * u, 0.f, false);
        PROF_MARK(4);  // publish
    }
#ifdef DFA_PCG_PROFILE
    if (tid == 0 && c == 0 && rank == 0) {
        for (int i = 0; i < 6; ++i) st->prof[i] += pc_[i];
        st->prof[6] += pw_[0], st->prof[7] += pw_[1];
    }
#endif
    if (gave_up) {  // nothing of this coordinate has been written: the guard launch solves it
        if ((tid & 63) == 0) team_give_up(ctl, c, host_abort);
        return;
    }
    if (owner) s.t[3 * a + c] += x;
    if (tid == 0 && rank == 0) {
        ctl->handled[c] = 1u;  // (every member is past the last barrier and adds its rows' x: done when the kernel is)
        // iterations of this launch = those of its slowest coordinate; the last team (or guard workgroup) to arrive books them
        atomicMax(&st->split_iters, it);
        __threadfence();
        if (atomicAdd(&st->split_ticket, 1u) == 2u) {
            __threadfence();
            st->pcg_iters += atomicExch(&st->split_iters, 0);
            st->split_ticket = 0u;
            if (st->grad_first == 0.0) st->grad_first = (double)rz0;
            st->gn_iters += 1;
        }
    }
}

__global__ __launch_bounds__(TEAM_NT) void pcg_team_kernel(SolveView s, SolveState* __restrict__ st, unsigned epoch0, int max_iter,
                                                           float pcg_tol, int* host_abort, int force_abort) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int rank_sh;
    __shared__ float red[3][TEAM_NT / 64];
    __shared__ float bc[4];
    if (st->done) return;
    const unsigned xcc = xcc_id();
    if (xcc >= 3u) return;
    const int c = (int)xcc, tid = threadIdx.x, D = s.D;
    TeamCtl* ctl = s.team_ctl;
    if (tid == 0) rank_sh = (int)atomicAdd(&ctl->count[c], 1u);
    __syncthreads();
    const int rank = rank_sh;
    if (rank >= TEAM_W) return;  // the team is complete without this workgroup
    // (ctl->handled[c]: team c has dealt with its coordinate — solved it, or found there was nothing to solve.  The guard
    // launch solves every coordinate nobody has dealt with: a team that gave up, and a team that never existed — a device
    // whose XCC_IDs are not 0, 1, 2, a partition mode with fewer XCDs)
    if (st->converged) {         // no-op iteration (see SolveState::converged); booked once
        if (tid == 0 && rank == 0) {
            ctl->handled[c] = 1u;
            if (c == 0) st->gn_iters += 1, st->gn_noop += 1;
        }
        return;
    }
    if ((force_abort >> c) & 1) {  // (development builds: the guard launch's test; 8 + mask: leave without a word, as a team that never existed)
        if (tid == 0 && !(force_abort & 8)) team_give_up(ctl, c, host_abort);
        return;
    }
    // which form: the longest row of the matrix (SolveState::max_row_nnz, raised by the assembly in front of this launch: the
    // same value in every workgroup) against the 16 slots per thread of the form that keeps t's replica in registers
    const int rows_ = (D + TEAM_W - 1) / TEAM_W, j_ = TEAM_NT / rows_;
    const bool treg = st->max_row_nnz <= j_ * TEAM_E_TREG && !(force_abort & 16);  // (development builds: 16 = the (m, t) form always)
    if (treg) team_member<TEAM_E_TREG, true>(s, st, ctl, smem, red, bc, c, rank, epoch0, max_iter, pcg_tol, host_abort);
    else team_member<TEAM_E, false>(s, st, ctl, smem, red, bc, c, rank, epoch0, max_iter, pcg_tol, host_abort);
}

// The guard behind every team launch: workgroup c resets team c's arrival counter for the next launch and, if the team
// gave up, solves coordinate c by itself — the same recurrence and stopping rules in one 1024-thread workgroup, u in LDS,
// the matrix streamed from the ELL as assembled, the rows' vectors in the plan's mb_* buffers (component c).  Slow (tens of
// microseconds per iteration) and rare by construction.
__global__ __launch_bounds__(1024) void pcg_team_guard_kernel(SolveView s, SolveState* __restrict__ st, int max_iter, float pcg_tol,
                                                              int* host_abort) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red0[16], red1[16];
    __shared__ unsigned todo_sh;
    const int c = blockIdx.x, tid = threadIdx.x, D = s.D;
    TeamCtl* ctl = s.team_ctl;
    if (tid == 0) {
        // the coordinate is this workgroup's if nobody has dealt with it: its team gave up (abort: counted by the team), or no
        // workgroup of the team launch ever took it (counted here: the plan goes back to the launched form)
        const unsigned handled = ctl->handled[c], gave_up = ctl->abort[c];
        todo_sh = st->done ? 0u : !handled;
        if (todo_sh && !gave_up && host_abort) atomicAdd_system(host_abort, 1);
        ctl->abort[c] = 0u, ctl->count[c] = 0u, ctl->handled[c] = 0u;
    }
    __syncthreads();
    if (!todo_sh) return;
    if (st->converged) {  // no-op iteration, booked once — by whoever deals with coordinate 0
        if (tid == 0 && c == 0) st->gn_iters += 1, st->gn_noop += 1;
        return;
    }
    float* u_s = (float*)smem;  // Dpad
    float *xs = (float*)s.mb_x + c, *rs = (float*)s.mb_r + c, *ps = (float*)s.mb_p + c, *ss = (float*)s.mb_s + c;  // [4 a]
    float joint_loc = 0.f;
    for (int a = tid; a < D; a += 1024) {
        const float d    = s.diag[a];
        const float minv = d > FLT_EPSILON ? 1.0f / d : 1.0f;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const float g = s.g[3 * a + cc];
            joint_loc     = fmaf(g, minv * g, joint_loc);
        }
        const float g = s.g[3 * a + c];
        xs[4 * a] = 0.f, rs[4 * a] = g, ps[4 * a] = 0.f, ss[4 * a] = 0.f;
        u_s[a] = minv * g;
    }
    const float rz0 = block_sum_f<16>(joint_loc, red0);  // (its barrier publishes u_s)
    if (st->grad_first > 0.0 && (double)rz0 <= 1e-12 * st->grad_first) {
        if (tid == 0 && c == 0) {
            st->gn_iters += 1;
            solve_mark_at_floor(st);
        }
        return;
    }
    const float floor_ = 1e-12f;
    const float tol2   = pcg_tol * pcg_tol > floor_ ? pcg_tol * pcg_tol : floor_;
    const float target = fmaxf(tol2 * rz0, solve_floor(st)) * (1.0f / 3.0f);
    float gamma_old = 1.f, alpha_old = 1.f;
    int it = 0;
    __syncthreads();
    while (it < max_iter) {
        float g_loc = 0.f, d_loc = 0.f;
        for (int a = tid; a < D; a += 1024) {
            const int cnt = min(s.ell_cnt[a], s.ell_cap);
            float w = 0.f;
            for (int q = 0; q < cnt; ++q) {
                const float2 en = s.ell[(size_t)q * D + a];
                w = fmaf(en.x, u_s[__float_as_int(en.y)], w);
            }
            ((float*)s.mb_w)[4 * a + c] = w;
            g_loc = fmaf(rs[4 * a], u_s[a], g_loc), d_loc = fmaf(w, u_s[a], d_loc);
        }
        const float gw = wave_total(g_loc), dw = wave_total(d_loc);
        if ((tid & 63) == 0) red0[tid >> 6] = gw, red1[tid >> 6] = dw;
        __syncthreads();
        float gamma = 0.f, delta = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) gamma += red0[i], delta += red1[i];
        if (!(gamma > target)) break;
        const float beta  = it == 0 ? 0.f : gamma / gamma_old;
        const float denom = it == 0 ? delta : delta - beta * gamma / alpha_old;
        if (!(denom > 0.f)) break;
        const float alpha = gamma / denom;
        for (int a = tid; a < D; a += 1024) {
            const float d    = s.diag[a];
            const float minv = d > FLT_EPSILON ? 1.0f / d : 1.0f;
            const float p = fmaf(beta, ps[4 * a], u_s[a]), sn = fmaf(beta, ss[4 * a], ((float*)s.mb_w)[4 * a + c]);
            const float rn = fmaf(-alpha, sn, rs[4 * a]);
            ps[4 * a] = p, ss[4 * a] = sn, xs[4 * a] = fmaf(alpha, p, xs[4 * a]), rs[4 * a] = rn;
            u_s[a] = minv * rn;  // (own row only; the gathers of this iteration are behind the reduction's barrier)
        }
        ++it;
        gamma_old = gamma, alpha_old = alpha;
        __syncthreads();
    }
    for (int a = tid; a < D; a += 1024) s.t[3 * a + c] += xs[4 * a];
    if (tid == 0) {
        atomicMax(&st->split_iters, it);
        __threadfence();
        if (atomicAdd(&st->split_ticket, 1u) == 2u) {
            __threadfence();
            st->pcg_iters += atomicExch(&st->split_iters, 0);
            st->split_ticket = 0u;
            if (st->grad_first == 0.0) st->grad_first = (double)rz0;
            st->gn_iters += 1;
        }
    }
}

// plans the team form can serve: (m, t) of every row + the partial sums in one CU's LDS, a member's rows on its threads (it
// is USED above the register-resident kernels: more than 2 048 nodes)
bool solve_team_pcg_fits(int D) {
    return sizeof(float2) * (size_t)((D + 3) & ~3) + sizeof(float) * TEAM_NT + 1024 <= 158 * 1024 && (D + TEAM_W - 1) / TEAM_W <= TEAM_NT;
}
int solve_team_pcg_rounds() { return TEAM_ROUNDS; }

template <class Kernel>
static hipError_t allow_big_lds(Kernel* k);

static hipError_t launch_team_pcg(const SolveView& s, SolveState* state, int max_iter, float pcg_tol, TeamPcg* tp, hipStream_t st) {
    hipError_t e = allow_big_lds(pcg_team_kernel);
    if (e == hipSuccess) e = allow_big_lds(pcg_team_guard_kernel);
    if (e != hipSuccess) return e;
    const unsigned epoch0 = tp->epoch;
    tp->epoch += (unsigned)max_iter + 8u;
    if (tp->epoch < epoch0 || tp->epoch == 0u) tp->epoch = 1u;  // (wrapped: never round 0, the value of a word nobody has written)
    const size_t lds = std::max(sizeof(float2) * (size_t)s.Dpad + sizeof(float) * TEAM_NT, TEAM_MIN_LDS);
    if (max_iter + 1 > TEAM_ROUNDS) return hipErrorInvalidValue;  // (route_pcg asks solve_team_pcg_fits first)
    {   // The barrier rounds of a launch are numbered from `epoch0`, a kernel ARGUMENT: a captured launch replayed from a HIP
        // graph would meet its own flag words of the replay before and sail through its barriers.  Refused loudly (the
        // launched form above 2 048 nodes synchronises with its stream and was never capturable either).
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return hipErrorStreamCaptureUnsupported;
    }
    // Two team launches of different plans must not share the device: each wants every CU of XCDs 0-2 for its members, and
    // two half-assembled teams would wait for each other until both time out (correct — the guard launches take over — but
    // 20 ms lost).  Launches on ONE stream are ordered anyway, and a process that only ever uses one stream for them pays
    // nothing here.  The first launch on a SECOND stream waits for the device once; from then on every team launch records
    // an event behind itself and a launch on another stream than the one before waits for it.  Per device, under a lock:
    // plans may be driven from several host threads.
    struct Turn {
        hipEvent_t ev      = nullptr;
        hipStream_t stream = nullptr;
        bool any = false, several = false;
    };
    static std::mutex mu;
    static std::map<int, Turn> turns;
    int dev = 0;
    e       = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    Turn& turn = turns[dev];
    if (turn.any && turn.stream != st) {
        if (!turn.several) {
            if ((e = hipEventCreateWithFlags(&turn.ev, hipEventDisableTiming)) != hipSuccess) return e;
            if ((e = hipDeviceSynchronize()) != hipSuccess) return e;  // (once per process and device: no event behind the launches so far)
            turn.several = true;
        } else if ((e = hipStreamWaitEvent(st, turn.ev, 0)) != hipSuccess) {
            return e;
        }
    }
    pcg_team_kernel<<<8 * TEAM_W, TEAM_NT, lds, st>>>(s, state, epoch0, max_iter, pcg_tol, tp->host_abort,
                                                             dev_env_int("DFA_MB_TEAM_ABORT", 0));
    pcg_team_guard_kernel<<<3, 1024, sizeof(float) * (size_t)s.Dpad, st>>>(s, state, max_iter, pcg_tol, tp->host_abort);
    tp->launches += 1;
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (turn.several) e = hipEventRecord(turn.ev, st);
    turn.stream = st, turn.any = true;
    return e;
}

int solve_pcg_max_nodes() { return 32768; }  // bounded by the transposition's LDS histogram (4 B x D)

template <class Kernel>
static hipError_t allow_big_lds(Kernel* k) {
    return allow_dynamic_lds((const void*)k, 160 * 1024 - 1024);  // once per (device, kernel)
}

#ifdef DFA_DEV_AB
// streaming kernel (matrix re-read from L2 every iteration): 1024 threads, RPT rows per thread
template <int RPT>
static hipError_t launch_streaming_pcg(const SolveView& s, SolveState* state, int max_iter, float pcg_tol,
                                       hipStream_t st) {
    hipError_t e = allow_big_lds(pcg_kernel<1024, RPT>);
    if (e != hipSuccess) return e;
    const size_t shmem = sizeof(float4) * (size_t)s.Dpad + 32 * sizeof(float) + 260 * sizeof(int);
    pcg_kernel<1024, RPT><<<1, 1024, shmem, st>>>(s, state, max_iter, pcg_tol);
    return hipGetLastError();
}
#endif  // DFA_DEV_AB

// register-resident kernel: NT threads own 2*P*NT rows, P pairs of E matrix slots per thread
// (NC = 3: one workgroup for the joint system; NC = 1: three workgroups, one coordinate each)
template <int NT, int P, int E, int NC>
static hipError_t launch_paired_pcg(const SolveView& s, SolveState* state, int max_iter, float pcg_tol,
                                    hipStream_t st) {
    hipError_t e = allow_big_lds(pcg_paired_kernel<NT, P, E, NC>);
    if (e != hipSuccess) return e;
    const size_t sh = sizeof(float4) * (size_t)s.Dpad + 32 * sizeof(float) + sizeof(int) * (260 + (size_t)s.Dpad * (s.deterministic ? 2 : 1));
    pcg_paired_kernel<NT, P, E, NC><<<NC == 1 ? 3 : 1, NT, sh, st>>>(s, state, max_iter, pcg_tol);
    return hipGetLastError();
}

#ifdef DFA_DEV_AB
static void sync_floor_switch() {
    static int done = -1;
    const int want = dev_env("DFA_PCG_NO_SOLVE_FLOOR") ? 1 : 0;
    if (done != want) {
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_floor_off), &want, sizeof(int));
        done = want;
    }
}
#endif

// The PCG of one linearisation.  Up to 2 048 nodes: the register-resident kernel, one workgroup per coordinate (a system
// that does not fit the registers is streamed inside the same launch); above: the many-workgroup PCG, which reads the
// assembled ELL directly (the single-workgroup streaming kernel spends ~1 ms per launch sorting and repacking the matrix
// by itself at 8 k nodes).
// Development builds (-DDFA_DEV_AB) also hold the kernels this routing was measured against, selected by
// DFA_PCG_VARIANT (read at every call: the tests switch it): 0 streaming, 1 register-resident with the three coordinates
// in ONE workgroup (shared CG scalars, as the oracle), 2 / 5 its 512-thread flavours, 3 many-workgroup at any size,
// 4 streaming up to 8 192 nodes.
static hipError_t route_pcg(const SolveView& s, SolveState* state, int max_iter, float pcg_tol, int* host_flag, MbGraphCache* gc,
                            TeamPcg* team, hipEvent_t& main_done, hipStream_t st) {
    const int D = s.D;
#ifdef DFA_DEV_AB
    sync_floor_switch();
    const int v2 = dev_env_int("DFA_PCG_VARIANT", -1);
    if (dev_env_int("DFA_MB_TEAM", 1) == 2 && team && team->ctl && !team->disabled && solve_team_pcg_fits(D) && max_iter < TEAM_ROUNDS &&
        !(team->host_abort && *(volatile int*)team->host_abort != 0))
        return launch_team_pcg(s, state, max_iter, pcg_tol, team, st);
    if (v2 == 3) return launch_mb_pcg(s, state, max_iter, pcg_tol, host_flag, gc, st);
    if (D <= 2048 && v2 != 0) {
        // 512 threads leave 256 VGPRs per lane (64 slots per row pair: k = 8 rows fit), 1024 threads 128 VGPRs (32 slots: k = 4)
        if (D <= 1024 && v2 == 1) return launch_paired_pcg<512, 1, 64, 3>(s, state, max_iter, pcg_tol, st);
        if (D <= 1024) return launch_paired_pcg<512, 1, 64, 1>(s, state, max_iter, pcg_tol, st);
        if (v2 == 2) return launch_paired_pcg<512, 2, 32, 3>(s, state, max_iter, pcg_tol, st);
        if (v2 == 1) return launch_paired_pcg<1024, 1, 32, 3>(s, state, max_iter, pcg_tol, st);
        if (v2 == 5) return launch_paired_pcg<512, 2, 32, 1>(s, state, max_iter, pcg_tol, st);
        return launch_paired_pcg<1024, 1, 32, 1>(s, state, max_iter, pcg_tol, st);
    }
    if (D <= 1024) return launch_streaming_pcg<1>(s, state, max_iter, pcg_tol, st);
    if (D <= 2048) return launch_streaming_pcg<2>(s, state, max_iter, pcg_tol, st);
    if (v2 == 4 && D <= 4096) return launch_streaming_pcg<4>(s, state, max_iter, pcg_tol, st);
    if (v2 == 4 && D <= 8192) return launch_streaming_pcg<8>(s, state, max_iter, pcg_tol, st);
#else
    // 512 threads leave 256 VGPRs per lane (64 slots per row pair: k = 8 rows fit), 1024 threads 128 VGPRs (32 slots: k = 4)
    if (D <= 1024) return launch_paired_pcg<512, 1, 64, 1>(s, state, max_iter, pcg_tol, st);
    if (D <= 2048) return launch_paired_pcg<1024, 1, 32, 1>(s, state, max_iter, pcg_tol, st);
#endif
    // (development builds: DFA_MB_TEAM=0 the launched form, =2 the team form at any size)
    if (team && team->ctl && !team->disabled && solve_team_pcg_fits(D) && max_iter < TEAM_ROUNDS && dev_env_int("DFA_MB_TEAM", 1) != 0) {
        // a team that gave up in an earlier launch (placement, starvation, a row too long) has said so in pinned memory: from
        // then on this plan takes the launched form (no synchronisation: the word is read as it stands)
        if (team->host_abort && *(volatile int*)team->host_abort != 0) team->disabled = true;
        else return launch_team_pcg(s, state, max_iter, pcg_tol, team, st);
    }
    return launch_mb_pcg(s, state, max_iter, pcg_tol, host_flag, gc, st);
}

// does the PCG of this plan run without any host synchronisation (the register-resident kernels, the team form)?  The
// launched many-workgroup form reads its stop flag back once per chunk of launches — and the plan's `converged` flag with it.
bool solve_pcg_is_async(const SolveView& s, const TeamPcg* team, int max_iter) {
    if (s.D <= 2048) return dev_env_int("DFA_PCG_VARIANT", -1) != 3;
    return team && team->ctl && !team->disabled && solve_team_pcg_fits(s.D) && max_iter < TEAM_ROUNDS &&
           dev_env_int("DFA_MB_TEAM", 1) != 0 && dev_env_int("DFA_PCG_VARIANT", -1) != 3 &&
           !(team->host_abort && *(volatile int*)team->host_abort != 0);
}

hipError_t solve_pcg(const SolveView& s, SolveState* state, int max_iter, float pcg_tol, int* host_flag, MbGraphCache* gc,
                     TeamPcg* team, hipEvent_t main_done, hipStream_t st) {
    const hipError_t e = route_pcg(s, state, max_iter, pcg_tol, host_flag, gc, team, main_done, st);
    if (main_done) (void)hipEventRecord(main_done, st);  // paths without a fallback launch
    return e;
}

__global__ void count_noop_kernel(SolveState* __restrict__ st, int n) { st->gn_iters += n, st->gn_noop += n; }
hipError_t solve_count_noop(SolveState* state, int n, hipStream_t st) {
    count_noop_kernel<<<1, 1, 0, st>>>(state, n);
    return hipGetLastError();
}


}  // namespace dfa
