// solve6.hip — the NORTH-STAR solve on gfx950: 6-DoF twist per deformation node, dual-quaternion
// blend of the k nearest nodes with normalised weights, projective point-to-plane data term
// against the live vertex / normal maps, ARAP-style edge regulariser, Gauss-Newton with a
// block-Jacobi (6x6) preconditioned CG on the block-sparse normal equations.
//
// Not in the reference's code: BASELINE.json:north_star / SURVEY.md App. B.2 ask for it, DESIGN.md
// §4.5 fixes the formulas, oracle/solve6_oracle.c states them on the CPU in double precision.  The
// reference pieces this follows where they apply: the point-to-plane row [s x n, n | n.(d - s)] and
// the nearest-pixel projective lookup of the rigid ICP (src/kfusion/cuda/proj_icp.cu:72-98,343-350),
// the RBF weight (src/dynfu/utils/node.cpp:29-36), the Tukey / Huber weights
// (src/dynfu/utils/opt_solver.cpp:204-268), w_reg^2 = lambda / (D k) (opt_solver.cpp:30).
//
// Once per frame (s6_build_graph): k-NN + normalised weights, regularisation graph, node -> rows transposition
// (sorted: reproducible sums), and s6_pattern — the block row pattern of the normal matrix and, per row of the
// energy, the slot of each of its neighbours in that pattern.
// Per Gauss-Newton iteration (all launches on one stream, nothing returns to the host):
//   s6_nodes      g^_i = T_i(g_i), M_i (the node's twist as dual-quaternion increments), clears the cost accumulators
//   s6_linearise  one lane per vertex: blend, project, gate, residual, Tukey weight; the row's 6-vector for
//                 neighbour j factors as f_j M_j l — l (8 numbers) and the k scalars f are written entry-major
//   s6_reg        one lane per regularisation edge
//   s6_assemble   one workgroup per node: streams the node's rows, accumulates the 8x8 moments sum rho f_a f_b l l^T
//                 per column slot in per-wave private LDS copies (no atomics), then H_ab = M_a S_ab M_b^T, the
//                 regulariser, damping, the inverse of the diagonal block and the gradient
//   s6_pcg_step   ONE launch per PCG iteration (Chronopoulos-Gear form), one wave per block row; replayed as a HIP
//                 graph.  Scalars are re-summed from per-workgroup partials by every wave (deterministic).
//   s6_update     T_i <- twist about g^_i applied on the left
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "dq_device.hpp"
#include "dev_switch.hpp"
#include "kernels.hpp"
#include "solve.hpp"
#include "solve6.hpp"

namespace dfa {

namespace {

__device__ __forceinline__ Quat qconj(Quat a) { return Quat{a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ float qdot(Quat a, Quat b) { return a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Quat pureq(f3 v) { return Quat{0.f, v.x, v.y, v.z}; }
__device__ __forceinline__ f3 qvec(Quat a) { return mk3(a.x, a.y, a.z); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// e_c^ (x) q for the basis vectors e_x, e_y, e_z
__device__ __forceinline__ Quat basis_mul(int c, Quat q) {
    return c == 0 ? Quat{-q.x, q.w, -q.z, q.y} : c == 1 ? Quat{-q.y, q.z, q.w, -q.x} : Quat{-q.z, -q.y, q.x, q.w};
}
// rotation + translation of a unit dual quaternion
__device__ __forceinline__ f3 dq_point(DQ q, f3 c) {
    const f3 rc = qvec(qmul(qmul(q.r, pureq(c)), qconj(q.r)));
    const f3 t  = qvec(qmul(q.d, qconj(q.r)));
    return mk3(rc.x + 2.f * t.x, rc.y + 2.f * t.y, rc.z + 2.f * t.z);
}

// ------------------------------------------------------------------------------------ graphs
// nearest node of every vertex (the key of the vertex sort).  A vertex WITHOUT a nearest node (NaN coordinates: the k-NN
// returns -1 for it) is filed under the last node: the sort is a permutation of ALL N vertices — a key the transposition
// skipped would leave positions [vptr[D], N) of the solver's arrays unwritten — and such a vertex keeps its row of -1
// ids and zero weights, i.e. it is passed through unchanged by every kernel, as it was before the sort existed.
__global__ __launch_bounds__(256) void s6_near_kernel(const int32_t* __restrict__ idx_nat, int N, int k, int D,
                                                      int32_t* __restrict__ near) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const int n = idx_nat[(size_t)v * k];
    near[v]     = n >= 0 && n < D ? n : D - 1;
}

constexpr int S6_SORT_MAX = 4096;

// bitonic sort of n <= S6_SORT_MAX keys in LDS by one 256-thread workgroup (buf holds the next power of two, padded with
// 0xffffffff)
// Stages with a stride of at most 64 keep a wave inside its own 128-key blocks (compare-exchange i = tid + 256 m touches the
// 2-stride-aligned block of key 2 i): the wave's LDS operations are ordered among themselves, no workgroup barrier is needed
// between such stages — 10 barriers instead of 66 for 2 048 keys.
__device__ __forceinline__ void s6_sort_lds(uint32_t* buf, int n2, int tid) {
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < n2 / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t x = buf[lo], y = buf[hi];
                if ((x > y) == up) buf[lo] = y, buf[hi] = x;
            }
            // the next stage leaves the wave's blocks (or there is none): everybody waits; else only the compiler does
            const int next = stride > 1 ? stride >> 1 : size;  // (after stride 1 comes size 2 x size with stride = size)
            if (stride > 64 || next > 64 || (stride == 1 && size == n2)) __syncthreads();
            else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"), __builtin_amdgcn_wave_barrier(), __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
}

// One workgroup per node: its vertices (grouped by the transposition in whatever order the atomics landed) sorted by
// index — a reproducible order — and gathered into the solver's arrays at their sorted positions, the radial basis
// weights normalised on the way (weight of node.cpp:29-36 divided by the row sum).
template <int K>
__global__ __launch_bounds__(256) void s6_permute_kernel(Solve6View s, const float* __restrict__ canon_user,
                                                         const float* __restrict__ canon_n_user,
                                                         const float* __restrict__ raw_w) {
    __shared__ uint32_t sortbuf[S6_SORT_MAX];
    const int a = blockIdx.x, tid = threadIdx.x, k = s.k;
    const int beg = s.vptr[a], len = s.vptr[a + 1] - beg;
    const bool sorted = len <= S6_SORT_MAX;
    if (sorted && len > 1) {
        int n2 = 1;
        while (n2 < len) n2 <<= 1;
        for (int i = tid; i < n2; i += 256) sortbuf[i] = i < len ? s.vlist[beg + i] : 0xffffffffu;
        __syncthreads();
        s6_sort_lds(sortbuf, n2, tid);
    } else if (sorted && len == 1 && tid == 0) {
        sortbuf[0] = s.vlist[beg];
    }
    __syncthreads();
    const bool wide = k == K;  // rows of K entries: 16-byte loads and stores
    for (int i = tid; i < len; i += 256) {
        const uint32_t v = sorted ? sortbuf[i] : s.vlist[beg + i];
        const size_t p   = (size_t)(beg + i);
        s.vperm[p]       = v;
#pragma unroll
        for (int c = 0; c < 3; ++c) s.canon_own[3 * p + c] = canon_user[3 * (size_t)v + c];
        if (canon_n_user)
#pragma unroll
            for (int c = 0; c < 3; ++c) s.canon_n_own[3 * p + c] = canon_n_user[3 * (size_t)v + c];
        float w[K];
        int32_t id[K];
        if (wide) {
#pragma unroll
            for (int q = 0; q < K / 4; ++q) {
                const float4 w4 = reinterpret_cast<const float4*>(raw_w + (size_t)v * K)[q];
                const int4 i4   = reinterpret_cast<const int4*>(s.idx_nat + (size_t)v * K)[q];
                w[4 * q] = w4.x, w[4 * q + 1] = w4.y, w[4 * q + 2] = w4.z, w[4 * q + 3] = w4.w;
                id[4 * q] = i4.x, id[4 * q + 1] = i4.y, id[4 * q + 2] = i4.z, id[4 * q + 3] = i4.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) w[j] = j < k ? raw_w[(size_t)v * k + j] : 0.f, id[j] = j < k ? s.idx_nat[(size_t)v * k + j] : -1;
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) sum += j < k ? w[j] : 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) w[j] = sum > 0.f ? w[j] / sum : 0.f;
        if (wide) {
#pragma unroll
            for (int q = 0; q < K / 4; ++q) {
                reinterpret_cast<float4*>(s.wn + p * K)[q] = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
                reinterpret_cast<int4*>(s.idx + p * K)[q]  = make_int4(id[4 * q], id[4 * q + 1], id[4 * q + 2], id[4 * q + 3]);
            }
        } else {
            for (int j = 0; j < k; ++j) s.idx[p * k + j] = id[j], s.wn[p * k + j] = w[j];
        }
    }
}

// k nearest OTHER nodes from a (k + 1)-NN list of the nodes among themselves
__global__ __launch_bounds__(256) void s6_reg_graph_kernel(const int32_t* __restrict__ raw, int D, int kreg, int k,
                                                           int32_t* __restrict__ reg_idx) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= D) return;
    int o = 0;
    for (int j = 0; j < kreg && o < k; ++j) {
        const int m = raw[(size_t)n * kreg + j];
        if (m >= 0 && m != n) reg_idx[(size_t)n * k + o++] = m;
    }
    for (; o < k; ++o) reg_idx[(size_t)n * k + o] = -1;
}

// ------------------------------------------------------------------------------------ blend
template <int K>
struct Blend {
    Quat a, b;   // un-normalised blended real / dual parts
    float m;     // |a|^2
    float s[K];  // hemisphere sign of each neighbour (0 = unused slot)
};

template <int K>
__device__ __forceinline__ void blend(const float* __restrict__ dq, const int32_t* idx, const float* wn, int k,
                                      Blend<K>& B) {
    B.a = Quat{0.f, 0.f, 0.f, 0.f}, B.b = B.a;
    Quat r0   = Quat{1.f, 0.f, 0.f, 0.f};
    bool have = false;
    // the node transforms four at a time, by unconditional loads (a neighbour that is not there reads node 0 and is not
    // used): with the load inside the `if` the k gathers were k dependent round trips to L2
#pragma unroll
    for (int h = 0; h < K; h += 4) {
        DQ q[4];
        bool on[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = h + jj;
            on[jj]      = j < k && idx[j] >= 0 && wn[j] != 0.f;
            q[jj]       = dq_load(dq + 8 * (size_t)(on[jj] ? idx[j] : 0));
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = h + jj;
            B.s[j]      = 0.f;
            if (!on[jj]) continue;
            if (!have) r0 = q[jj].r, have = true;
            const float sg = qdot(q[jj].r, r0) < 0.f ? -1.f : 1.f;
            B.s[j]         = sg;
            const float w  = wn[j] * sg;
            B.a = qadd(B.a, qscale(q[jj].r, w)), B.b = qadd(B.b, qscale(q[jj].d, w));
        }
    }
    B.m = qdot(B.a, B.a);
}

template <int K>
__device__ __forceinline__ f3 blend_point(const Blend<K>& B, f3 c) {
    const Quat ac = qconj(B.a);
    const f3 u    = qvec(qmul(qmul(B.a, pureq(c)), ac));
    const f3 t    = qvec(qmul(B.b, ac));
    const float im = 1.f / B.m;
    return mk3((u.x + 2.f * t.x) * im, (u.y + 2.f * t.y) * im, (u.z + 2.f * t.z) * im);
}
template <int K>
__device__ __forceinline__ f3 blend_normal(const Blend<K>& B, f3 n) {
    const f3 u     = qvec(qmul(qmul(B.a, pureq(n)), qconj(B.a)));
    const float im = 1.f / B.m;
    return mk3(u.x * im, u.y * im, u.z * im);
}

// ------------------------------------------------------------------------------- per iteration
// g^_n = T_n(g_n) and M_n of node n at transform q (part of the launch that produced q: s6_begin / s6_update)
__device__ __forceinline__ void s6_node_now(const Solve6View& s, int n, DQ q) {
    const f3 g = dq_point(q, mk3(s.node_pos[3 * n], s.node_pos[3 * n + 1], s.node_pos[3 * n + 2]));
    s.ghat[3 * n] = g.x, s.ghat[3 * n + 1] = g.y, s.ghat[3 * n + 2] = g.z;
    // M_n (6 x 8): the twist components of node n as dual-quaternion increments (W, Wd).  A data row's
    // 6-vector for a neighbour n is f_vn * M_n l_v with the per-vertex functional l_v = (lW, lD) of
    // s6_linearise — the assembly accumulates 8 x 8 moments of l and applies M afterwards.
    float* M = s.mnode + 48 * (size_t)n;
#pragma unroll
    for (int col = 0; col < 3; ++col) {
        const f3 e    = mk3(col == 0 ? 1.f : 0.f, col == 1 ? 1.f : 0.f, col == 2 ? 1.f : 0.f);
        const Quat W  = basis_mul(col, q.r);                                         // omega = e_col
        const Quat Wd = qadd(basis_mul(col, q.d), qmul(pureq(cross(g, e)), q.r));    // rotation about g^: v0 = g^ x e
        float* r0 = M + 8 * col;
        r0[0] = W.w, r0[1] = W.x, r0[2] = W.y, r0[3] = W.z, r0[4] = Wd.w, r0[5] = Wd.x, r0[6] = Wd.y, r0[7] = Wd.z;
        float* r1 = M + 8 * (3 + col);                                               // translation e_col: (0, e^ r)
        r1[0] = 0.f, r1[1] = 0.f, r1[2] = 0.f, r1[3] = 0.f, r1[4] = W.w, r1[5] = W.x, r1[6] = W.y, r1[7] = W.z;
    }
}

__device__ __forceinline__ float tukey6(float err, float offset, float c) {  // opt_solver.cpp:204-231
    const float e = err / offset;
    if (e < c) {
        const float t = 1.f - (e * e) / (c * c);
        return t * t;
    }
    return 0.f;
}

// The energy of a linearisation: every workgroup leaves ITS sum (and count of valid rows) in cost_part / valid_part, the
// first workgroup of the assembly that follows adds them up in a fixed order (s6_cost_total).  Two atomicAdds per workgroup
// on the state block's two words — 4 000 same-address fp64 atomics per launch at C3 — cost 15-25 us of the launch
// (C2 0.033 -> 0.017 ms, C3 0.061 -> 0.041), and their order decided the last bits of the sum.
__device__ __forceinline__ void block_add_cost(double cost, unsigned int nvalid, const Solve6View& s, bool through = false) {
    __shared__ double sc[8];
    __shared__ unsigned int sn[8];
    cost   = wave_sum_all(cost);
    nvalid = (unsigned int)wave_sum_all((double)nvalid);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sc[wave] = cost, sn[wave] = nvalid;
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0;
        unsigned int n = 0;
        for (int w = 0; w < nw; ++w) c += sc[w], n += sn[w];
        if (through) {  // read by ANOTHER workgroup of this launch (its last one): written through, past this XCD's L2
            __hip_atomic_store(&s.cost_part[blockIdx.x], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&s.valid_part[blockIdx.x], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            s.cost_part[blockIdx.x] = c, s.valid_part[blockIdx.x] = n;
        }
    }
}
__host__ __device__ inline int s6_linearise_blocks(int N, int D, int k) { return (N + 255) / 256 + (D * k + 255) / 256; }
// by all 256 threads of one workgroup: the partial sums of the last linearisation, in a fixed order, into the state block
__device__ __forceinline__ void s6_cost_total(const Solve6View& s, Solve6State* st, bool through = false) {
    __shared__ double tc[4];
    __shared__ unsigned long long tn[4];
    const int n = s6_linearise_blocks(s.N, s.D, s.k), tid = threadIdx.x;
    double c = 0.0;
    unsigned long long v = 0ull;
    if (through) {
        // partials of the launch in flight: loads that do not stop at this XCD's L2 — each a round trip of ~2 us, and this is
        // the one workgroup the whole launch waits for: eight of them in flight per thread (clamped addresses, masked sums;
        // a loop of load-wait-add was five dependent round trips at C2: 9 us per linearisation)
        constexpr int Q = 8;
        for (int base = tid; base < n; base += 256 * Q) {
            double cq[Q];
            unsigned int vq[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int i = min(base + 256 * q, n - 1);
                cq[q] = __hip_atomic_load(&s.cost_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vq[q] = __hip_atomic_load(&s.valid_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (base + 256 * q < n) c += cq[q], v += vq[q];
        }
    } else
    for (int i = tid; i < n; i += 256) c += s.cost_part[i], v += s.valid_part[i];
    c = wave_sum_all(c);
    v = (unsigned long long)wave_sum_all((double)v);  // (counts below 2^53: exact)
    if ((tid & 63) == 0) tc[tid >> 6] = c, tn[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) st->cost = (tc[0] + tc[1]) + (tc[2] + tc[3]), st->valid = (tn[0] + tn[1]) + (tn[2] + tn[3]);
}

__device__ __forceinline__ double s6_reg_edge(const Solve6View& s, int e, float wreg2, float psi_reg, int update_w);

// gn_tol > 0: the Gauss-Newton stopping rule (oracle/oracle.h: orc6_params.gn_tol states it; the reference runs Opt with
// earlyOut = true and nonLinearIter as a cap, src/dynfu/dyn_fusion.cpp:183-189) and the bookkeeping s6_bookkeeping does
// otherwise — by ONE thread of the linearisation's last workgroup, after s6_cost_total: the assembly's workgroups, which run
// next, must already know (a launch behind an ended outer iteration returns at entry).  Slots that are skipped write
// nothing: s6_begin has marked every slot "skipped".
__device__ __forceinline__ void s6_decide(Solve6State* st, const S6Decide& d) {
    const bool in = d.gi < S6_HIST;
    const double cost = st->cost;
    const unsigned long long valid = st->valid;
    if (!st->have_first) st->initial_cost = cost, st->valid_first = valid, st->have_first = 1;
    int stop = 0;
    if (d.gn > 0) {
        const double ref = st->cost_ref;
        if (cost > (1.0 + (double)d.gn_tol) * ref) stop = 2;                       // the step raised the energy: undone
        else if (d.closing || ref - cost <= (double)d.gn_tol * ref) stop = 1;      // converged: kept, no further step
    }
    if (stop == 2) {
        st->final_cost = st->cost_ref, st->valid_last = st->valid_ref, st->gn_rejected += 1;
    } else {
        st->final_cost = cost, st->valid_last = valid;
        if (stop == 1) st->gn_converged += d.closing ? 0 : 1;
        else st->cost_ref = cost, st->valid_ref = valid, st->gn_solves += 1;
    }
    if (in) {
        st->cost_hist[d.gi] = cost, st->valid_hist[d.gi] = (unsigned int)valid, st->pcg_it_hist[d.gi] = 0;
        st->pcg_rel_hist[d.gi] = stop ? 0.f : 1.f, st->pcg_tol_hist[d.gi] = stop ? 0.f : sqrtf(d.f.tol2), st->stop_hist[d.gi] = stop;
        if (st->hist_n < d.gi + 1) st->hist_n = d.gi + 1;
    }
    st->gn_iters += 1, st->cur = d.gi, st->gn_stop = stop;
    st->tol2 = d.f.tol2, st->pcg_last_it = 0, st->pcg_done = stop != 0;
    st->ew_gamma = d.f.ew_gamma, st->ew_min2 = d.f.ew_min2, st->ew_max2 = d.f.ew_max2, st->ew_slot = d.f.ew_slot;
}

// Tail of the linearisation when it decides (gn_tol > 0): every workgroup has published its partial sums write-through
// (block_add_cost); thread 0 waits for that store to leave, then arrives — a counter per shard, a top counter for the last
// arriver of each shard (one word would serialise ~2 000 device-scope atomics at ~11 ns each), re-armed by whoever
// completes them — and the LAST workgroup of the launch sums the partials in their fixed order and decides.  The pattern
// of the reference-mode linearise_kernel (solve.hip).  One launch fewer per Gauss-Newton slot than the one-workgroup kernel
// this replaced — which measured the same frame time (C2 841 against 845 frames/s, C3 317.5 against 316.8: that launch hid
// in the stream's launch pipeline).
__device__ __forceinline__ void s6_linearise_tail(const Solve6View& s, Solve6State* st, const S6Decide& d) {
    __shared__ int is_last;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int shard   = blockIdx.x % S6_LIN_SHARDS;
        const unsigned int members = (gridDim.x - shard + S6_LIN_SHARDS - 1) / S6_LIN_SHARDS;
        const unsigned int nshards = min((unsigned int)S6_LIN_SHARDS, gridDim.x);
        int last                   = 0;
        if (__hip_atomic_fetch_add(&st->lin_ticket[shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(&st->lin_ticket[shard], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&st->lin_ticket[S6_LIN_SHARDS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nshards - 1) {
                __hip_atomic_store(&st->lin_ticket[S6_LIN_SHARDS], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    s6_cost_total(s, st, true);
    if (threadIdx.x == 0) s6_decide(st, d);
}

// blocks [0, nlin): the data term, a lane per vertex; blocks from nlin on: the regulariser, a lane per edge
template <int K>
__global__ __launch_bounds__(256) void s6_linearise_kernel(Solve6View s, Solve6State* st, Solve6Image img,
                                                           Solve6Params prm, int update_w, int nlin, float wreg2, int gate,
                                                           const S6Decide dec) {
    if (gate && st->gn_stop) return;  // (uniform) the outer iteration has ended: nothing to linearise, nothing to decide
    if ((int)blockIdx.x >= nlin) {  // (uniform)
        const int e = ((int)blockIdx.x - nlin) * 256 + (int)threadIdx.x;
        block_add_cost(e < s.D * s.k ? s6_reg_edge(s, e, wreg2, prm.psi_reg, update_w) : 0.0, 0u, s, dec.on != 0);
        if (dec.on) s6_linearise_tail(s, st, dec);
        return;
    }
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    double cost = 0.0;
    unsigned int nvalid = 0;
    if (v < s.N) {
        const int k = s.k;
        int32_t idx[K];
        float wn[K];
        if (k == K) {  // (uniform) the vertex's neighbours and weights as 16-byte loads: K / 4 each instead of K dwords
#pragma unroll
            for (int q = 0; q < K / 4; ++q) {
                const int4 iv   = reinterpret_cast<const int4*>(s.idx + (size_t)v * K)[q];
                const float4 wv = reinterpret_cast<const float4*>(s.wn + (size_t)v * K)[q];
                idx[4 * q] = iv.x, idx[4 * q + 1] = iv.y, idx[4 * q + 2] = iv.z, idx[4 * q + 3] = iv.w;
                wn[4 * q] = wv.x, wn[4 * q + 1] = wv.y, wn[4 * q + 2] = wv.z, wn[4 * q + 3] = wv.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                idx[j] = j < k ? s.idx[(size_t)v * k + j] : -1;
                wn[j]  = j < k ? s.wn[(size_t)v * k + j] : 0.f;
            }
        }
        float w_eff = 0.f, rr = 0.f;
        bool ok     = false;
        float h[K], swr = 0.f;
        float4 l0 = make_float4(0.f, 0.f, 0.f, 0.f), l1 = l0;
#pragma unroll
        for (int j = 0; j < K; ++j) h[j] = 0.f;
        Blend<K> B;
        blend<K>(s.dq, idx, wn, k, B);
        const f3 c = mk3(s.canon[3 * (size_t)v], s.canon[3 * (size_t)v + 1], s.canon[3 * (size_t)v + 2]);
        f3 p = c, nl = mk3(0.f, 0.f, 0.f);
        if (B.m > 0.f) {
            p = blend_point<K>(B, c);
            if (p.z > 0.f) {
                // nearest pixel (proj_icp.cu:80-86 uses __float2int_rn)
                const int u = (int)rintf(img.fx * (p.x / p.z) + img.cx), w = (int)rintf(img.fy * (p.y / p.z) + img.cy);
                if (u >= 0 && w >= 0 && u < img.cols && w < img.rows) {
                    const float4 L  = *reinterpret_cast<const float4*>((const char*)img.vmap + (size_t)w * img.vstep + 16 * (size_t)u);
                    const float4 Ln = *reinterpret_cast<const float4*>((const char*)img.nmap + (size_t)w * img.nstep + 16 * (size_t)u);
                    if (L.x == L.x && Ln.x == Ln.x) {
                        const f3 dl      = mk3(p.x - L.x, p.y - L.y, p.z - L.z);
                        const float dist = sqrtf(dot3(dl, dl));
                        nl               = mk3(Ln.x, Ln.y, Ln.z);
                        bool gate        = dist <= prm.dist_thresh;
                        if (gate && s.canon_n) {
                            const f3 n0 = mk3(s.canon_n[3 * (size_t)v], s.canon_n[3 * (size_t)v + 1], s.canon_n[3 * (size_t)v + 2]);
                            gate        = dot3(blend_normal<K>(B, n0), nl) >= prm.cos_thresh;
                        }
                        if (gate) ok = true, rr = dot3(nl, dl);
                    }
                }
            }
        }
        if (ok) {
            if (update_w) s.rho[v] = tukey6(fabsf(rr), prm.tukey_offset, prm.psi_data);
            w_eff = s.rho[v];
            // n . dp = lW . W + lD . Wd  for the twist's (W, Wd) = (omega^ r, omega^ d + v0^ r), times w~ s / m
            const Quat ac = qconj(B.a), nq = pureq(nl);
            const Quat T = qmul(ac, nq);                                  // a* n^
            const Quat U = qmul(qmul(pureq(c), ac), nq);                  // (c^ a*) n^
            const Quat S = qscale(qmul(qconj(B.b), nq), -1.f);            // -b* n^
            const float np = dot3(nl, p);
            const Quat lD = Quat{-T.w, T.x, T.y, T.z};
            const Quat lW = Quat{-(U.w + S.w) - np * B.a.w, (U.x + S.x) - np * B.a.x, (U.y + S.y) - np * B.a.y,
                                 (U.z + S.z) - np * B.a.z};
            const float im = 1.f / B.m;
            // row of the Jacobian for neighbour j: f_j * M_j l with l = (lW, lD); see s6_node_now.  ONE record per vertex —
            // l, the k scalars f, the robust weight — which the assembly gathers through each node's row list (round 2
            // wrote the record once per neighbour, at the row's place in that node's list: k times the bytes, and an
            // index scatter to build per frame)
            // ... with the robust weight folded in: h_j = sqrt(rho) f_j, so that rho f_a f_b = h_a h_b is ONE product in the
            // assembly, rows without weight vanish by themselves, and -J^T r needs (sqrt(rho) r) h_a
            const float sw = sqrtf(w_eff);
#pragma unroll
            for (int j = 0; j < K; ++j) h[j] = j < k ? sw * (wn[j] * B.s[j] * im) : 0.f;
            l0 = make_float4(lW.w, lW.x, lW.y, lW.z), l1 = make_float4(lD.w, lD.x, lD.y, lD.z);
            swr    = sw * rr;
            cost   = (double)w_eff * (double)rr * (double)rr;
            nvalid = w_eff > 0.f;
        }
        // no association: a record of zeros (l too: 0 x a stale NaN would be NaN)
        float4* rec = reinterpret_cast<float4*>(s.rec + (12 + K) * (size_t)v);  // 2 + K / 4 + 1 chunks of 16 bytes per vertex
        rec[0] = l0, rec[1] = l1;
#pragma unroll
        for (int q = 0; q < K / 4; ++q) rec[2 + q] = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
        rec[2 + K / 4] = make_float4(swr, 0.f, 0.f, 0.f);
    }
    block_add_cost(cost, nvalid, s, dec.on != 0);
    if (dec.on) s6_linearise_tail(s, st, dec);
}

__device__ __forceinline__ double s6_reg_edge(const Solve6View& s, int e, float wreg2, float psi_reg, int update_w) {
    double cost = 0.0;
    const int n = e / s.k, m = s.reg_idx[e];
    float* er = s.rres + 3 * (size_t)e;
    float* vc = s.rvec + 18 * (size_t)e;
    float ev[3] = {0.f, 0.f, 0.f}, vec[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) vec[i] = 0.f;
    if (m >= 0) {
        const f3 y   = dq_point(dq_load(s.dq + 8 * (size_t)n), mk3(s.node_pos[3 * m], s.node_pos[3 * m + 1], s.node_pos[3 * m + 2]));
        const f3 ghm = mk3(s.ghat[3 * m], s.ghat[3 * m + 1], s.ghat[3 * m + 2]);
        const f3 ghn = mk3(s.ghat[3 * n], s.ghat[3 * n + 1], s.ghat[3 * n + 2]);
        ev[0] = y.x - ghm.x, ev[1] = y.y - ghm.y, ev[2] = y.z - ghm.z;
        const float en = sqrtf(ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]);
        if (update_w) s.rhub[e] = en <= psi_reg ? 1.f : psi_reg / en;  // opt_solver.cpp:233-268
        const float l0 = y.x - ghn.x, l1 = y.y - ghn.y, l2 = y.z - ghn.z;
        // rows of [ -[l]x | I ]
        vec[0] = 0.f, vec[1] = l2, vec[2] = -l1, vec[3] = 1.f;
        vec[6] = -l2, vec[7] = 0.f, vec[8] = l0, vec[10] = 1.f;
        vec[12] = l1, vec[13] = -l0, vec[14] = 0.f, vec[17] = 1.f;
        cost = (double)wreg2 * (double)s.rhub[e] * (double)en * (double)en;
    }
    er[0] = ev[0], er[1] = ev[1], er[2] = ev[2];
#pragma unroll
    for (int i = 0; i < 18; ++i) vc[i] = vec[i];
    return cost;
}

// --------------------------------------------------------------------------------- assembly
constexpr int S6_HASH = 128;

__device__ __forceinline__ int hash_slot(int* keys, int b) {
    unsigned h = ((unsigned)b * 2654435761u) >> 25;  // 7 bits
    for (int probe = 0; probe < S6_HASH; ++probe) {
        // (a plain read first: thousands of pairs look up a few dozen keys, and an LDS atomic per look-up on a handful of
        // addresses serialises the wave; an entry, once written, never changes)
        int seen = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
        if (seen == -1) seen = atomicCAS(&keys[h], -1, b);
        if (seen == -1 || seen == b) return (int)h;
        h = (h + 1) & (S6_HASH - 1);
    }
    return -1;
}

// inverse of a symmetric positive definite 6x6 (Cholesky); identity-scaled fallback if it fails
__device__ void inv6(const float* M, float* out) {
    float L[36];
    bool ok = true;
    for (int i = 0; i < 36; ++i) L[i] = 0.f;
    for (int i = 0; i < 6 && ok; ++i)
        for (int j = 0; j <= i; ++j) {
            float sm = M[6 * i + j];
            for (int q = 0; q < j; ++q) sm -= L[6 * i + q] * L[6 * j + q];
            if (i == j) {
                if (!(sm > 0.f)) {
                    ok = false;
                    break;
                }
                L[6 * i + i] = sqrtf(sm);
            } else {
                L[6 * i + j] = sm / L[6 * j + j];
            }
        }
    if (!ok) {
        for (int i = 0; i < 36; ++i) out[i] = 0.f;
        return;  // z = 0 for this node (as the oracle does)
    }
    // columns of the inverse: solve L L^T x = e_c
    for (int c = 0; c < 6; ++c) {
        float y[6], x[6];
        for (int i = 0; i < 6; ++i) {
            float sm = i == c ? 1.f : 0.f;
            for (int q = 0; q < i; ++q) sm -= L[6 * i + q] * y[q];
            y[i] = sm / L[6 * i + i];
        }
        for (int i = 5; i >= 0; --i) {
            float sm = y[i];
            for (int q = i + 1; q < 6; ++q) sm -= L[6 * q + i] * x[q];
            x[i] = sm / L[6 * i + i];
        }
        for (int i = 0; i < 6; ++i) out[6 * i + c] = x[i];
    }
}

// bookkeeping of the linearisation that just finished (one thread of the assembly launch): costs, the slot of this
// Gauss-Newton iteration in the per-iteration history, the stop test of the PCG that follows
__device__ __forceinline__ void s6_bookkeeping(Solve6State* st, const S6Forcing f) {
    if (!st->have_first) st->initial_cost = st->cost, st->valid_first = st->valid, st->have_first = 1;
    st->final_cost = st->cost, st->valid_last = st->valid;
    const int h = st->gn_iters;
    if (h < S6_HIST) {
        st->cost_hist[h] = st->cost, st->pcg_it_hist[h] = 0, st->pcg_rel_hist[h] = 1.f, st->pcg_tol_hist[h] = sqrtf(f.tol2);
        st->valid_hist[h] = (unsigned int)st->valid, st->stop_hist[h] = 0, st->hist_n = h + 1;
    }
    st->gn_iters = h + 1, st->gn_solves = h + 1, st->cur = h;
    st->tol2 = f.tol2, st->pcg_last_it = 0, st->pcg_done = 0;
    st->ew_gamma = f.ew_gamma, st->ew_min2 = f.ew_min2, st->ew_max2 = f.ew_max2, st->ew_slot = f.ew_slot;
}

// column c of the inverse of a symmetric positive definite 6x6 (every caller lane factorises for itself: six lanes invert
// the block in the time of one column); zero if the factorisation fails, as inv6
__device__ __forceinline__ void inv6_column(const float* M, float* out, int c, float (&x)[6]) {
    float L[6][6];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            float sm = M[6 * i + j];
#pragma unroll
            for (int q = 0; q < j; ++q) sm -= L[i][q] * L[j][q];
            if (i == j) {
                ok      = ok && sm > 0.f;
                L[i][i] = sqrtf(sm);
            } else {
                L[i][j] = sm / L[j][j];
            }
        }
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float sm = i == c ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < i; ++q) sm -= L[i][q] * y[q];
        y[i] = sm / L[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        float sm = y[i];
#pragma unroll
        for (int q = i + 1; q < 6; ++q) sm -= L[q][i] * x[q];
        x[i] = sm / L[i][i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = ok ? x[i] : 0.f, out[6 * i + c] = x[i];
}

// Sparsity pattern of block row a (fixed by the graphs of the frame, built once per set_problem):
// the diagonal first, then every node that shares a vertex with a or is joined to it by a
// regularisation edge, ascending.
constexpr int S6_MAXSLOT_PATTERN = 48;  // = S6_MAXSLOT (declared below), the plan capacity of a block row
constexpr int S6_UNITS = 256;           // work units of a node's assembly = threads of its workgroup
constexpr int S6_DEAL  = 8;             // a node's sorted rows are dealt out in this many interleaved runs (s6_pattern_kernel)

#ifdef DFA_S6_TIMING  // development builds only: per-workgroup phase clocks of the assembly (tools/ns_assemble_phases.py)
__device__ unsigned long long s6_tbuf[16384 * 16];
#define S6_TICK(var) const unsigned long long var = clock64()
extern "C" __attribute__((visibility("default"))) int dfa_dev_s6_timing(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(s6_tbuf), sizeof(unsigned long long) * 16 * (size_t)n);
}
#else
#define S6_TICK(var)
#endif
__global__ __launch_bounds__(256) void s6_pattern_kernel(Solve6View s, Solve6State* st) {
    __shared__ int keys[S6_HASH];
    __shared__ int cnt_sh;
    __shared__ uint32_t sortbuf[S6_SORT_MAX];
    const int a = blockIdx.x, tid = threadIdx.x, k = s.k;
    S6_TICK(pk0);
    for (int i = tid; i < S6_HASH; i += 256) keys[i] = -1;
    if (tid == 0) cnt_sh = 0;
    // The transposition fills a node's list in whatever order its workgroups' atomics land; sort it so
    // that the assembly adds the rows in ascending (vertex, slot) order: run-to-run reproducible sums.
    {
        const int beg = s.node_ptr[a], len = s.node_ptr[a + 1] - beg;
        if (len > 1 && len <= S6_SORT_MAX) {
            int n2 = 1;
            while (n2 < len) n2 <<= 1;
            for (int i = tid; i < n2; i += 256) sortbuf[i] = i < len ? s.node_list[beg + i] : 0xffffffffu;
            __syncthreads();
            s6_sort_lds(sortbuf, n2, tid);
            // ... and deal the sorted rows out in S6_DEAL interleaved runs (rows 0, 8, 16, ..., then 1, 9, ...): the assembly
            // stages the list a few hundred rows at a time, and a vertex's neighbours come in clusters along the sorted list —
            // in sorted order one pass held all the records of some blocks and none of others, and the lanes (one block each)
            // waited for the fullest (3.5 us of a workgroup's 42 at C3).  Now every pass is a uniform sample.
            const int per = len / S6_DEAL, extra = len - per * S6_DEAL;
            for (int i = tid; i < len; i += 256) {
                const int g = i % S6_DEAL, t = i / S6_DEAL;
                s.node_list[beg + g * per + min(g, extra) + t] = sortbuf[i];
            }
        }
        if (tid == 0) {  // the few regularisation edges arriving at a: insertion sort
            const int rb = s.rnode_ptr[a], re = s.rnode_ptr[a + 1];
            for (int i = rb + 1; i < re; ++i) {
                const uint32_t x = s.rnode_list[i];
                int j = i - 1;
                for (; j >= rb && s.rnode_list[j] > x; --j) s.rnode_list[j + 1] = s.rnode_list[j];
                s.rnode_list[j + 1] = x;
            }
        }
    }
    __syncthreads();
    S6_TICK(pk1);
    // ---- the columns of block row a: every neighbour of every vertex of a's rows (and the regularisation partners) into a
    // 128-entry LDS hash; the position a pair's neighbour landed at is remembered per pair (a byte: 254 = a itself, 255 =
    // none), so that the pair's slot is a table look-up once the columns are ranked.  The bytes live in the (now idle) sort
    // buffer when they fit, else in global scratch.
    const int pbeg = s.node_ptr[a], plen = s.node_ptr[a + 1] - pbeg, npairs = plen * k;
    uint8_t* es_lds   = reinterpret_cast<uint8_t*>(sortbuf);
    const bool in_lds = npairs <= (int)sizeof(sortbuf);
    uint8_t* es       = in_lds ? es_lds : s.eslot + (size_t)pbeg * k;
    bool lost = false;
    // (four rows of a thread at a time: their list entries requested together, then their neighbour ids — row by row the two
    // dependent loads of every row were a chain of 2 x rows / 256 round trips)
    for (int r0 = tid; r0 < plen; r0 += 4 * 256) {
        unsigned v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = s.node_list[pbeg + min(r0 + 256 * i, plen - 1)] / (unsigned)k;
        int nb[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) nb[i][j] = j < k ? s.idx[(size_t)v[i] * k + j] : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 256 * i;
            if (r >= plen) break;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j >= k) break;
                const int b = nb[i][j];
                int hp      = 255;
                if (b == a) hp = 254;
                else if (b >= 0) {
                    hp = hash_slot(keys, b);
                    if (hp < 0) lost = true, hp = 255;
                }
                es[(size_t)r * k + j] = (uint8_t)hp;
            }
        }
    }
    if (tid < k) {
        const int m = s.reg_idx[a * k + tid];
        if (m >= 0 && m != a && hash_slot(keys, m) < 0) lost = true;
    }
    for (int e = s.rnode_ptr[a] + tid; e < s.rnode_ptr[a + 1]; e += 256) {
        const int n = (int)(s.rnode_list[e] / (unsigned)k);
        if (n != a && hash_slot(keys, n) < 0) lost = true;
    }
    __syncthreads();
    S6_TICK(pk2);
    __shared__ uint8_t hrank[S6_HASH];  // hash position -> slot of that column (255: not stored)
    if (tid < S6_HASH) {
        int r = 254;
        if (keys[tid] >= 0) {
            r = 0;
            for (int h = 0; h < S6_HASH; ++h) r += keys[h] >= 0 && keys[h] < keys[tid];
            if (r + 1 < s.cap) s.bcols[(size_t)a * s.cap + r + 1] = keys[tid];
            atomicAdd(&cnt_sh, 1);
        }
        hrank[tid] = (uint8_t)(r + 1 < s.cap ? r + 1 : 255);
    }
    __syncthreads();
    const int nblk = cnt_sh + 1, stored = nblk < s.cap ? nblk : s.cap;
    for (int i = stored + tid; i < s.cap; i += 256) s.bcols[(size_t)a * s.cap + i] = -1;  // the matvec reads no count
    if (tid == 0) {
        s.bcols[(size_t)a * s.cap] = a;
        s.bcnt[a] = stored;
        // (a running maximum: most workgroups find it already at or above theirs and skip the same-address atomic)
        if (nblk > __atomic_load_n(&st->max_row_blocks, __ATOMIC_RELAXED)) atomicMax(&st->max_row_blocks, nblk);
        if (nblk > s.cap || plen > 60000) st->overflow = 1;  // (16-bit offsets in the split below)
    }
    if (lost) st->overflow = 1;
    // The matrix is symmetric, H_ba = H_ab^T: the assembly computes a block once, in the row of the smaller node index, and
    // writes it to both rows.  Pair lists are kept for slot 0 (the row's own neighbour: one record per row) and for the "upper"
    // slots (column > a; the columns ascend, so these are the slots from `fu` on), the others stay empty.
    __shared__ int pcnt[64], pstart[64];
    __shared__ int fu_sh;
    if (tid < S6_HASH / 2) {  // first wave: slots 1 .. fu - 1 are the columns below a = the keys below a
        const uint64_t l0 = __ballot(keys[tid] >= 0 && keys[tid] < a), l1 = __ballot(keys[tid + 64] >= 0 && keys[tid + 64] < a);
        if (tid == 0) {
            const int f = min(1 + __popcll(l0) + __popcll(l1), stored);
            fu_sh = f, s.bfu[a] = f;
        }
    }
    __syncthreads();
    S6_TICK(pk3);
    const int fu = fu_sh;
    // ---- the pattern by slot (s6_assemble2_kernel): a STABLE split of the (row, neighbour) pairs by slot, as a counting sort —
    // thread t takes a contiguous run of pairs, counts its pairs per slot into its own column of hist (16-bit), the columns
    // are scanned per slot (wave w: slots w, w + 4, ...), then every thread writes its pairs, in order, from its offsets on:
    // lists in ascending pair order, run-to-run identical, O(pairs) work (ballots per slot and 64 pairs: 83 of the kernel's
    // 194 us per workgroup at C3).
    // (S6_SPLIT listed slots — slot 0 and the upper ones, in slot order — per round: all 48 at once are 24 KiB of counters, which
    // left room for three workgroups per CU instead of five)
    // (Round 4, tools/ns_pattern_phases.py with finer marks at C3: of the split's 29 us a thread spends 6 translating its bytes,
    // 4 clearing + counting, 3 in the scan and 16 WRITING — 8 300 four-byte stores per workgroup, each lane of a wave to
    // another list: the store transactions, not LDS, are what it waits for.  Giving every thread whole rows with its bytes in
    // a run padded to an odd number of words — no bank shared by the lanes of a pass, where 32-40-byte runs share four —
    // changed nothing: 29.0 us either way.)
    constexpr int S6_SPLIT = 24;
    __shared__ uint16_t hist[S6_SPLIT][256];
    __shared__ int run_sh;
    const int nlisted = stored > fu ? 1 + stored - fu : 1;  // listed slot li: slot 0 (li = 0) or slot fu + li - 1
    const int per_thread = (npairs + 255) / 256, p0 = min(tid * per_thread, npairs), p1 = min(p0 + per_thread, npairs);
    for (int p = p0; p < p1; ++p) {  // hash position -> listed slot index (255: none)
        const int hp = es[p];
        const int sl = hp == 254 ? 0 : hp == 255 ? 255 : (int)hrank[hp];
        es[p] = (uint8_t)(sl == 0 ? 0 : (sl >= fu && sl < stored) ? sl - fu + 1 : 255);
    }
    if (tid < 64) pcnt[tid] = 0;
    if (tid == 0) run_sh = pbeg * k;  // the lists of node a live in pair_list[pbeg k, (pbeg + plen) k): slot 0 (every row), then the upper slots
    for (int l0 = 0; l0 < nlisted; l0 += S6_SPLIT) {
        const int nl = min(S6_SPLIT, nlisted - l0);
        for (int i = tid; i < S6_SPLIT * 256 / 8; i += 256) reinterpret_cast<uint4*>(&hist[0][0])[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (int p = p0; p < p1; ++p) {
            const int li = (int)es[p] - l0;
            if (li >= 0 && li < nl) hist[li][tid] += 1;
        }
        __syncthreads();
        {
            const int wave = tid >> 6, lane = tid & 63;
            for (int li = wave; li < nl; li += 4) {
                uint2* hq      = reinterpret_cast<uint2*>(&hist[li][4 * lane]);  // threads 4 lane .. 4 lane + 3
                const uint2 hv = *hq;
                const uint32_t c0 = hv.x & 0xffffu, c1 = hv.x >> 16, c2 = hv.y & 0xffffu, c3 = hv.y >> 16;
                const uint32_t mine = c0 + c1 + c2 + c3;
                uint32_t incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                const uint32_t e0 = incl - mine, e1 = e0 + c0, e2 = e1 + c1, e3 = e2 + c2;
                *hq = make_uint2(e0 | (e1 << 16), e2 | (e3 << 16));
                if (lane == 63) pcnt[l0 + li == 0 ? 0 : fu + l0 + li - 1] = (int)incl;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int run = run_sh;
            for (int li = l0; li < l0 + nl; ++li) {
                const int q = li == 0 ? 0 : fu + li - 1;
                if (li == 1)
                    for (int z = 1; z < fu; ++z) pstart[z] = run;  // (the lower slots: empty lists)
                pstart[q] = run, run += pcnt[q];
            }
            run_sh = run;
            if (l0 + nl >= nlisted) {
                if (nlisted == 1)
                    for (int z = 1; z < stored; ++z) pstart[z] = run;
                pstart[stored] = run;
            }
        }
        __syncthreads();
        // the pairs of this thread, in order, each to the next free place of its slot's list
        for (int p = p0; p < p1; ++p) {
            const int li = (int)es[p] - l0;
            if (li < 0 || li >= nl) continue;
            const int r = p / k, q = l0 + li == 0 ? 0 : fu + l0 + li - 1;
            const uint32_t oj  = s.node_list[pbeg + r] % (unsigned)k;  // the row's own neighbour slot
            const uint32_t off = hist[li][tid];
            hist[li][tid]      = (uint16_t)(off + 1u);
            s.pair_list[pstart[q] + (int)off] = ((uint32_t)r << 8) | (oj << 4) | (uint32_t)(p - r * k);
        }
        if (l0 + S6_SPLIT < nlisted) __syncthreads();  // (the counters are cleared for the next round)
    }
    S6_TICK(pk4);
    __syncthreads();
    if (tid <= stored) s.pair_ptr[(size_t)a * (s.cap + 1) + tid] = pstart[tid];
    // ---- work units of the assembly: S6_UNITS per node, one per LANE of its workgroup.  A unit walks every n-th record of ONE
    // block's list (slot 0 — every row — or an upper slot), starting at its `phase`; a block with c of the node's T records
    // gets 2 (1 + floor((S6_UNITS / 2 - blocks) c / T)) units, so every unit of the workgroup walks about T / S6_UNITS records,
    // the units of a block are neighbours (neighbouring lanes read neighbouring rows) and lanes 2 i, 2 i + 1 always work
    // for the same block.
    __shared__ int ustart[64], ucount[64];
    if (tid < 64) {
        const int q    = tid == 0 ? 0 : fu + tid - 1;  // lane = block: slot 0, then the upper slots
        const int mine = (tid == 0 || q < stored) ? pcnt[tid == 0 ? 0 : q] : 0;
        const int nblk2 = __popcll(__ballot(mine > 0));
        int total = mine;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
        const int n = mine > 0 ? 2 * (1 + (int)(((long long)(S6_UNITS / 2 - nblk2) * mine) / total)) : 0;  // (even: see the assembly's closing part)
        int incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (tid >= o) incl += t;
        }
        ustart[tid] = incl - n, ucount[tid] = n;
    }
    __syncthreads();
    {
        uint32_t info = 63u;  // idle
        for (int i = 0; i < 64; ++i) {
            const int u0 = ustart[i], n = ucount[i];
            if (tid >= u0 && tid < u0 + n) info = (uint32_t)(i == 0 ? 0 : fu + i - 1) | ((uint32_t)(tid - u0) << 6) | ((uint32_t)n << 16);
        }
        s.utab[(size_t)a * S6_UNITS + tid] = info;
    }
    S6_TICK(pk5);
#ifdef DFA_S6_TIMING
    if (tid == 0 && a < 16384) {
        unsigned long long* o = s6_tbuf + 16 * (size_t)a;
        const unsigned long long pk6 = clock64();
        o[0] = pk0, o[1] = pk1 - pk0, o[2] = pk2 - pk1, o[3] = pk3 - pk2, o[4] = pk4 - pk3, o[5] = pk5 - pk4, o[6] = pk6 - pk5, o[10] = pk6 - pk0;
    }
#endif
}

// slot of node a in the block row of each of its columns (the mirror position of block (a, slot)); 255 = not there
__global__ __launch_bounds__(256) void s6_rslot_kernel(Solve6View s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.D * s.cap) return;
    const int a = i / s.cap, q = i - a * s.cap;
    int r = 255;
    if (q < s.bcnt[a]) {
        const int b = s.bcols[i];
        if (q == 0) r = 0;
        else if (b >= 0) {
            const int32_t* cb = s.bcols + (size_t)b * s.cap;
            const int nb      = s.bcnt[b];
            for (int t = 1; t < nb; ++t)
                if (cb[t] == a) r = t;
        }
    }
    s.rslot[i] = (uint8_t)r;
}

// Values of block row a.  A data row's 6-vector for neighbour j factors as f_j * M_j l (l: 8 numbers per
// vertex, M_j: 6 x 8 per node, s6_node_now), so the block H_ab = sum_v rho f_a f_b (M_a l)(M_b l)^T is
// M_a S_ab M_b^T with the 8 x 8 moment S_ab = sum_v (rho f_a f_b) l l^T: the rows that touch node a are read
// as 8 + k floats instead of 6 k, S is accumulated, and M is applied once per block at the end.
// History at 4 k nodes, k = 8 (one Gauss-Newton iteration): one lane per row with LDS atomics 3.7 ms (64-way
// same-address conflicts); every (slot, c) thread scanning every row 2.2 ms (VALU-bound); per-wave private LDS copies
// of the moments, rows streamed 1.0 ms, with factored rows 0.81 ms (LDS-bandwidth-bound: 4 KiB read + written per row
// and wave; the "first form", in the tree until round 3); registers instead of LDS copies, a wave per slot, 0.65 ms (second
// form); symmetric blocks, a quad of lanes per work unit 0.315 ms (third form); a LANE per work unit — below.
constexpr int S6_MAXSLOT = S6_MAXSLOT_PATTERN;  // = plan capacity of a block row

typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef DFA_S6_ABLATE
#define DFA_S6_ABLATE 0  // development builds only (-DDFA_S6_ABLATE=mask): 1 no record walk, 4 no off-diagonal epilogue, 8 contiguous rows
#endif

// Fourth form (round 3).  The third form (a quad of lanes owned a unit's 8 x 8 moment, 16 records per wave instruction) was
// bound by instruction ISSUE, not by memory or LDS: SQ counters at C3 gave 22 k vector instructions per workgroup of which
// 10 % were the products, the SIMDs' issue slots ~90 % taken, and neither removing exposed load waits nor an XCD-aware
// node order moved the time (tools/ns_assemble_phases.py).  Per 16 records a wave spent ~30 instructions
// on decoding the record, the validity selects, the addresses and eight DPP broadcasts, for eight packed FMAs.  Now:
//  (1) a LANE PER UNIT: every lane walks its own records (s6_pattern: S6_UNITS units per node, every n-th record of one
//      block's list) and owns that unit's whole moment — decode, addresses and LDS reads serve 64 records per wave
//      instruction instead of 16, and nothing is exchanged between lanes until the node is finished;
//  (2) SYMMETRY of the moment itself: S = sum c l l^T has 36 distinct entries, not 64 — 16 packed + 4 scalar FMAs per
//      record (and 4 packed multiplies for c l);
//  (3) slot 0 (every row's own neighbour) is a list like the others (its records are (row, own slot), its coefficient the
//      same rho f_a f_j): no separate loop, no per-pass cross-lane reduction; the gradient's 8 sums ride on its units;
//  (4) a lane's records come in batches of NG registers, loaded from clamped addresses before the staging barrier;
//  (5) rows staged by unconditional 16-byte loads from selected addresses (with a branch per kind of chunk the compiler
//      waited for every load before issuing the next: two loads of different width into one register).
// Kept from the third form: block (a, b) computed once, by the workgroup of the smaller index, written to both rows
// (H_ba = H_ab^T bit for bit); the coefficient rho f_a f_j of every (row, neighbour) formed once when the rows are staged;
// the regulariser's edges of the node staged in LDS; mirror blocks transposed through LDS; the PCG's start in the tail.
constexpr int S6_REGIN = 24;  // arriving regularisation edges staged in LDS (more: read from global memory)

// position of S[i][j] (= S[j][i]) in a unit's 36 accumulators: even rows i = 2t as pairs (j, j + 1) from j = i — 20
// numbers; odd rows i = 2t + 1 as pairs from j = i + 1 — 12 numbers; the odd rows' diagonal entries — 4 numbers
__host__ __device__ constexpr int s6_sym(int i, int j) {
    if (i > j) { const int t = i; i = j; j = t; }
    const int t = i >> 1;
    if ((i & 1) == 0) return 2 * ((t == 0 ? 0 : t == 1 ? 4 : t == 2 ? 7 : 9)) + (j - i);
    if (j == i) return 32 + t;
    return 20 + 2 * (t == 0 ? 0 : t == 1 ? 3 : 5) + (j - i - 1);
}

#ifndef DFA_S6_WAVES
#define DFA_S6_WAVES 4  // waves per SIMD the register allocation aims at (four workgroups per CU by LDS: <= 128 VGPRs, 3-4 spilled)
#endif
#ifdef DFA_S6_DEBUG
__device__ float s6_dbg[256 * (S6_MAXSLOT * 36 + 8)];
extern "C" __attribute__((visibility("default"))) int dfa_dev_s6_moments(float* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(s6_dbg), sizeof(float) * 256 * (S6_MAXSLOT * 36 + 8));
}
#endif
#ifdef DFA_S6_DEBUG
__device__ float s6_dbgm[256 * S6_MAXSLOT * 48];
extern "C" __attribute__((visibility("default"))) int dfa_dev_s6_m(float* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(s6_dbgm), sizeof(float) * 256 * S6_MAXSLOT * 48);
}
#endif
constexpr size_t s6_assemble_lds(int K, int RC) {  // dynamic segment: the staged rows, later the units' moments + the blocks' sums
    const size_t rows = (size_t)((RC * (2 + K / 4 + 1) + 255) / 256) * 4096, closing = (size_t)(S6_UNITS / 2 * 36 + S6_MAXSLOT * 36) * 4;
    return rows > closing ? rows : closing;
}
template <int K, int S6_RC, bool KEXACT>  // KEXACT: k == K (the common case: divisions by k are shifts)
__global__ __launch_bounds__(256, DFA_S6_WAVES) void s6_assemble2_kernel(Solve6View s, Solve6State* st, float wreg2, float damping,
                                                                         const S6Forcing forcing, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) char s6_dyn[];
    // a staged row = the vertex's record as s6_linearise wrote it: l = (lW, lD) (32 bytes) | h_j = sqrt(rho) f_j, j < K |
    // sqrt(rho) r, -, -, - : RS bytes, in the order of the 16-byte chunks the row is fetched in (the LDS-DMA writes a wave's 64
    // chunks to 1 KiB in lane order).  Nothing is prepared per pass: rho f_a f_j = h_a h_j is formed where it is used.
    constexpr int RS = 32 + 4 * K + 16, CFO = 32, MTO = 32 + 4 * K;
    // after the passes the dynamic segment holds, in turn: the moments of half of the units [0, 18 KiB); the blocks' sums (36
    // floats per slot) behind them; M of the column nodes (48 floats per slot) and the transposition tile over the — by then
    // dead — units.  (All 256 units at once would be 36 KiB: the fourth workgroup of a CU would not fit.)
    float* upart     = reinterpret_cast<float*>(s6_dyn);                                                   // [S6_UNITS / 2][36]
    float(*accS)[36] = reinterpret_cast<float(*)[36]>(s6_dyn + sizeof(float) * 36 * (S6_UNITS / 2));
    float* smb       = reinterpret_cast<float*>(s6_dyn);
    __shared__ float g8w[4][8];           // -J^T r in l coordinates, per wave
    __shared__ float g8s[8];
    __shared__ float diag[36];
    __shared__ float gsh[6];              // -J^T W r of the node
    __shared__ float rout[8][24];         // edges leaving a: neighbour (bits), weight, residual (3), vectors (18)
    __shared__ float rin[S6_REGIN][24];   // edges arriving at a: source node (bits), weight, residual (3), vectors (18)
    __shared__ uint32_t bfirst[S6_MAXSLOT];  // first unit | units << 16 of every block that has units
    // (workgroup -> node in launch order.  Consecutive nodes on one XCD — node (b mod 8) D / 8 + b / 8 for workgroup b, so that
    // a vertex record is fetched into one L2 instead of up to k — measured 0.048 against 0.050 ms at C2 and 0.156 against
    // 0.150 at C3: not kept.)
    int a = blockIdx.x;
    const int tid = threadIdx.x, k = s.k;
#ifdef DFA_DEV_AB  // DFA_XCD_MAP=1: the experiment above, kept for its counters (profiles/r06_xcd_map.md)
    if (xcd_map && a < s.D && (s.D & 7) == 0) {
        a = (a & 7) * (s.D >> 3) + (a >> 3);           // a contiguous eighth of the order per XCD ...
        if (xcd_map == 2 && s.xcd_perm) a = s.xcd_perm[a];  // ... of the Morton order of the node positions
    }
#endif
    if (a == s.D) {  // one workgroup more than nodes: the energy of the linearisation this launch follows, the state block's bookkeeping
        if (forcing.decided) return;  // (the linearisation's last workgroup has done both)
        s6_cost_total(s, st);
        if (tid == 0) s6_bookkeeping(st, forcing);
        return;
    }
    if (forcing.decided && st->gn_stop) return;  // (uniform) the outer iteration has ended: no normal equations
    S6_TICK(tk0);
#ifdef DFA_S6_TIMING
    const unsigned long long wk0 = wall_clock64();
    unsigned long long tk_stage = 0, tk_prep = 0, tk_up = 0;
#endif

    const int cnt = s.bcnt[a], fu = s.bfu[a];
    const int beg = s.node_ptr[a], len = s.node_ptr[a + 1] - beg;
    const int wave = tid >> 6, lane = tid & 63;
    const int32_t* pptr = s.pair_ptr + (size_t)a * (s.cap + 1);
    const int rib = s.rnode_ptr[a], nri = s.rnode_ptr[a + 1] - rib;
    if (tid < S6_MAXSLOT) bfirst[tid] = 0u;
    // positions of the moment's entries the closing part's lane z reads: S[p][2 z + h] at symt[z][2 p + h]
    __shared__ __attribute__((aligned(16))) int symt[4][16];
    if (tid >= 128 && tid < 192) symt[(tid - 128) >> 4][tid & 15] = s6_sym((tid & 15) >> 1, 2 * ((tid - 128) >> 4) + (tid & 1));
    // the row's columns and the mirror slots: read by every phase of the closing part (two dependent global loads there)
    __shared__ int32_t scol[S6_MAXSLOT];
    __shared__ uint8_t srs[S6_MAXSLOT];
    __shared__ uint32_t rent[S6_REGIN];  // the arriving edges' (source node k + slot): one round trip less in the closing part
    // (everything the prologue needs from global memory is requested first, by unconditional loads from clamped indices, and
    // parked afterwards: a load under its own `if` followed by its LDS store made six dependent round trips of it)
    constexpr int EPT = (S6_RC + 255) / 256;
    const int col_i       = (int)((size_t)a * s.cap) + min(tid, s.cap - 1);
    const int32_t scol_v  = s.bcols[col_i];
    const uint8_t srs_v   = s.rslot[col_i];
    const uint32_t rent_v = s.rnode_list[rib + min(tid & 63, max(nri, 1) - 1)];  // (an empty list reads its neighbour's first entry: unused)
    uint32_t ent0[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) ent0[q] = s.node_list[beg + min(tid + 256 * q, max(len, 1) - 1)];
    // ---- this lane's work unit (s6_pattern: utab): every un-th record of block uq's list, from its phase on
    const int nup        = cnt > fu ? cnt - fu : 0;
    const uint32_t uinfo = s.utab[(size_t)a * S6_UNITS + tid];
    if (tid < S6_MAXSLOT) scol[tid] = tid < cnt ? scol_v : -1, srs[tid] = tid < cnt ? srs_v : (uint8_t)255;
    else if (tid >= 64 && tid < 64 + S6_REGIN) rent[tid - 64] = tid - 64 < nri ? rent_v : 0u;
    const int uq         = (int)(uinfo & 63u);
    const int un         = (int)((uinfo >> 16) & 1023u);
    const bool uhas      = uq < cnt && (uq == 0 || uq >= fu) && un > 0;
    const bool uown      = uhas && uq == 0;  // a unit of slot 0: it also sums the gradient
    int ucur = 0, uend = 0;                  // this lane's next record; end of the list
    {
        const int q0 = uhas ? uq : 0, b0 = pptr[q0], b1 = pptr[q0 + 1];
        if (uhas) ucur = b0 + (int)((uinfo >> 6) & 1023u), uend = b1;
    }
    v2f me[10], mo[6], G[4];  // even rows of the moment (pairs), odd rows beyond the diagonal (pairs); the gradient
    float md[4];              // the odd rows' diagonal entries
#pragma unroll
    for (int e = 0; e < 10; ++e) me[e] = v2f{0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 6; ++e) mo[e] = v2f{0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) G[e] = v2f{0.f, 0.f}, md[e] = 0.f;
    // a lane's records are fetched a BATCH of NG at a time into registers that are indexed statically: unconditional loads
    // from clamped addresses, all in flight together.  The first batch of a pass flies during the staging; a second one in a
    // pass is rare.  (One record ahead in a rotating register made the compiler wait for the load in the same trip.)
    constexpr int NG = 8;
    uint32_t rb[NG];
    auto load_batch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int at = ucur + g * un;
            rb[g]        = s.pair_list[uhas && at < uend ? at : 0];
        }
    };
    constexpr int RS4 = 2 + K / 4 + 1, NCH = (S6_RC * RS4 + 255) / 256;  // 16-byte chunks of a record; chunks per thread and pass
    __shared__ uint32_t sent[2][S6_RC];  // (vertex k + slot) of the rows of this pass / the next one
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int r = tid + 256 * q;
        if (r < min(S6_RC, len)) sent[0][r] = ent0[q];
    }
    int pass = 0;
    S6_TICK(tk1);
    for (int r0 = 0; r0 < len; r0 += S6_RC, pass ^= 1) {
        const int nr = min(S6_RC, len - r0);
        S6_TICK(tp0);
        __syncthreads();  // the pass before is done with the staged rows (and has stored this pass's list entries)
        const size_t e0 = (size_t)(beg + r0);
        // the next pass's list entries: loaded now, parked in LDS at the end of this pass (one register meanwhile)
        uint32_t nxt_ent[EPT];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = tid + 256 * q;
            nxt_ent[q]  = r0 + S6_RC + r < len ? s.node_list[e0 + S6_RC + r] : 0u;
        }
        // The rows of the pass: one record per vertex (s6_linearise), gathered through the node's row list; RS4 chunks per row —
        // l (2 x 16 bytes), f (K / 4 x 16 bytes) from the vertex's 64-byte line, (weight, weight x residual, 0, 0) from the
        // per-vertex array — one chunk per lane and step, straight into LDS (global_load_lds_dwordx4: no staging registers, no
        // ds_write pass; rows past the end fetch row 0 again, into rows nobody reads).
        // (the chunk's row and part are recomputed in every pass: hoisted out of the pass loop, as the compiler would have it,
        // they are 2 x NCH registers that live — spilled — through the whole kernel.  The list entries are all read before the
        // first DMA is issued: the compiler orders a ds_read behind an LDS-DMA in flight with s_waitcnt vmcnt(0), one memory
        // round trip per chunk.)
        unsigned tl = (unsigned)tid;
        asm volatile("" : "+v"(tl));
        uint32_t ens[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const unsigned r = (tl + 256u * q) / (unsigned)RS4;
            ens[q]           = sent[pass][min(r, (unsigned)nr - 1u)];
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const unsigned i = tl + 256u * q, r = i / (unsigned)RS4, c = i - (unsigned)RS4 * r;
#if DFA_S6_ABLATE & 8
            const size_t v = (e0 + r) % (size_t)s.N + 0 * ens[q];  // (timing only: contiguous rows instead of the gather)
#else
            const size_t v = KEXACT ? ens[q] / (unsigned)K : ens[q] / (unsigned)k;  // (K: a shift)
#endif
            const float4* src = reinterpret_cast<const float4*>(s.rec + (12 + K) * v) + c;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(s6_dyn + 16 * (256 * q + 64 * wave)), 16, 0, 0);
        }
        // this lane's next NG records fly while the rows settle
        load_batch();
        __syncthreads();
        S6_TICK(tp1);
        S6_TICK(tp2);
        // ---- the records of this pass: a lane takes its unit's records while their rows are staged (the lists ascend)
#if !(DFA_S6_ABLATE & 1)
        {
            bool act       = uhas && ucur < uend;
            const bool gw  = __any(uown);  // (this wave has units of slot 0)
            auto walk_batch = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (!__any(act)) break;
                    // (an active lane is at the g-th record of its batch: it advanced in every trip so far)
                    const uint32_t pr = rb[g];  // row of the node's list << 8 | the row's own slot << 4 | neighbour slot
                    const unsigned rr = (pr >> 8) - (unsigned)r0;
                    const bool in     = act && rr < (unsigned)nr;
                    const unsigned ri = in ? rr : 0u;
                    const char* rp  = s6_dyn + RS * ri;
                    const float4 la = reinterpret_cast<const float4*>(rp)[0], lb = reinterpret_cast<const float4*>(rp)[1];
                    const float ho  = *reinterpret_cast<const float*>(rp + CFO + ((pr >> 2) & 60u));  // h_a: sqrt(rho) f_a
                    const float hj  = *reinterpret_cast<const float*>(rp + CFO + ((pr << 2) & 60u));  // h_j
                    const float cf  = in ? ho * hj : 0.f;                                             // rho f_a f_j
                    const v2f L[4]  = {v2f{la.x, la.y}, v2f{la.z, la.w}, v2f{lb.x, lb.y}, v2f{lb.z, lb.w}};
                    const v2f cc    = v2f{cf, cf};
                    v2f F[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) F[t] = cc * L[t];
                    constexpr int eo[4] = {0, 4, 7, 9}, oo[4] = {0, 3, 5, 6};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const v2f fe = v2f{F[t].x, F[t].x}, fo = v2f{F[t].y, F[t].y};
#pragma unroll
                        for (int q = 0; q < 4 - t; ++q) me[eo[t] + q] = __builtin_elementwise_fma(fe, L[t + q], me[eo[t] + q]);
                        md[t] = fmaf(F[t].y, L[t].y, md[t]);
#pragma unroll
                        for (int q = 0; q < 3 - t; ++q) mo[oo[t] + q] = __builtin_elementwise_fma(fo, L[t + 1 + q], mo[oo[t] + q]);
                    }
                    if (gw) {  // -J^T r: the slot-0 units' rows, each once
                        const float gr = uown && in ? -*reinterpret_cast<const float*>(rp + MTO) * ho : 0.f;  // -(sqrt(rho) r) h_a
                        const v2f gg   = v2f{gr, gr};
#pragma unroll
                        for (int t = 0; t < 4; ++t) G[t] = __builtin_elementwise_fma(gg, L[t], G[t]);
                    }
                    if (in) ucur += un;
                    act = in && ucur < uend;
                }
            };
            // (the first batch was loaded before the staging barrier; a refill — rare — is a path of its own: with the refill
            // at the bottom of ONE loop the batch registers were loop-carried loads and the common path waited for them in every
            // trip — C3 0.143 -> 0.120 ms.  Pulling the next pass's records towards L2 meanwhile, by 4-byte LDS-DMA loads into a
            // scrap area issued from an asm statement, made it 0.135: not kept.)
            walk_batch();
            while (__any(act)) {
                load_batch();  // (more than NG records of a unit in one pass)
                walk_batch();
            }
        }
#endif
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = tid + 256 * q;
            if (r < S6_RC) sent[pass ^ 1][r] = nxt_ent[q];
        }
#ifdef DFA_S6_TIMING
        {
            S6_TICK(tp4);
            tk_stage += tp1 - tp0, tk_prep += tp2 - tp1, tk_up += tp4 - tp2;
        }
#endif
    }
    S6_TICK(tk2);
    // ---- the units' moments: lanes 2 i and 2 i + 1 (the same block, s6_pattern) are added by DPP, the even lanes park the
    // pair's moment in LDS (128 x 36 floats, in the rows' area), then every block is the sum of its pairs in unit order
    __syncthreads();  // the passes are over: the rows' area is free (and bfirst is cleared, scol / rent are written)
    S6_TICK(tk2a);
    if (uhas && ((uinfo >> 6) & 1023u) == 0u) bfirst[uq] = (uint32_t)(tid >> 1) | ((uint32_t)(un >> 1) << 16);
    {   // the gradient: summed over the wave (only the slot-0 units hold anything)
        float gv[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) gv[2 * t] = G[t].x, gv[2 * t + 1] = G[t].y;
        if (__any(uown)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] = wave_sum_all(gv[e]);
        }
        if (lane < 8) {
            float x = gv[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) x = lane == e ? gv[e] : x;
            g8w[wave][lane] = x;
        }
    }
    {
        auto pair = [](float x) __attribute__((always_inline)) {  // x + the other lane of the pair's (quad_perm [1, 0, 3, 2])
            return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));
        };
        float w[36];
#pragma unroll
        for (int e = 0; e < 10; ++e) w[2 * e] = pair(me[e].x), w[2 * e + 1] = pair(me[e].y);
#pragma unroll
        for (int e = 0; e < 6; ++e) w[20 + 2 * e] = pair(mo[e].x), w[21 + 2 * e] = pair(mo[e].y);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[32 + e] = pair(md[e]);
        if ((tid & 1) == 0) {
            float4* d0 = reinterpret_cast<float4*>(upart + 36 * (tid >> 1));
#pragma unroll
            for (int e = 0; e < 9; ++e) d0[e] = make_float4(w[4 * e], w[4 * e + 1], w[4 * e + 2], w[4 * e + 3]);
        }
    }
    // ---- what the epilogue needs from global memory is requested now (the moments' registers are free), into registers — the
    // node's regularisation edges (leaving: 1 value per thread, arriving: up to 3) and M of the row's column nodes (48 floats
    // per slot) — and parked in LDS once the moments are summed: the round trip passes under the sums and their barriers
    constexpr int RINQ = (S6_REGIN * 24 + 255) / 256, MQ = (S6_MAXSLOT * 48 + 255) / 256;
    // (one unconditional 4-byte load from a selected address per value: a branch per field makes them wait for each other)
    auto edge_field = [&](int f, size_t e) __attribute__((always_inline)) {  // field f of edge e: 0 neighbour, 1 weight, 2-4 residual, 5-22 vectors
        const float* p = reinterpret_cast<const float*>(s.reg_idx + e);
        p = f == 1 ? s.rhub + e : p;
        p = f >= 2 && f < 5 ? s.rres + 3 * e + (f - 2) : p;
        p = f >= 5 && f < 23 ? s.rvec + 18 * e + (f - 5) : p;
        return *p;
    };
    float rov, riv[RINQ], mreg[MQ];
    {
        const int q = tid < k * 24 ? tid / 24 : 0, f = tid - 24 * (tid / 24);
        rov = edge_field(f, (size_t)(a * k + q));
    }
#pragma unroll
    for (int t = 0; t < RINQ; ++t) {
        const int i = tid + 256 * t, q = i / 24, f = i - 24 * q;
        riv[t]      = edge_field(f, (size_t)rent[i < min(nri, S6_REGIN) * 24 ? q : 0]);
    }
#pragma unroll
    for (int t = 0; t < MQ; ++t) {
        const int i = tid + 256 * t, slot = i / 48;
        mreg[t]     = i < cnt * 48 ? s.mnode[48 * (size_t)scol[slot] + (i - 48 * slot)] : 0.f;
    }
    S6_TICK(tk2b);
    __syncthreads();
    S6_TICK(tk2c);
    for (int i = tid; i < (1 + nup) * 36; i += 256) {
        const int bi = i / 36, e = i - 36 * bi, slot = bi == 0 ? 0 : fu + bi - 1;
        const uint32_t bf = bfirst[slot];
        const int u0 = (int)(bf & 0xffffu), n = (int)(bf >> 16);
        float sum = 0.f;
        for (int u = 0; u < n; ++u) sum += upart[36 * (u0 + u) + e];
        accS[slot][e] = sum;
    }
    if (tid < 8) g8s[tid] = (g8w[0][tid] + g8w[1][tid]) + (g8w[2][tid] + g8w[3][tid]);
    __syncthreads();
#ifdef DFA_S6_DEBUG
    if (a >= DFA_S6_DEBUG && a < DFA_S6_DEBUG + 256) {
        const int da = a - DFA_S6_DEBUG;
        for (int i = tid; i < S6_MAXSLOT * 36; i += 256) s6_dbg[(size_t)da * (S6_MAXSLOT * 36 + 8) + i] = (&accS[0][0])[i];
        if (tid < 8) s6_dbg[(size_t)da * (S6_MAXSLOT * 36 + 8) + S6_MAXSLOT * 36 + tid] = g8s[tid];
    }
#endif
    S6_TICK(tk3);
    // H_ab = M_a S_ab M_b^T, regularisation, output: thread (slot, row) finishes row `row` of the block in `slot` — of slot 0
    // and of the upper slots; an upper block also goes, transposed, to the row of its column.
    // M of the row's column nodes (48 floats per slot) and the edges: from the registers they arrived in
#pragma unroll
    for (int t = 0; t < MQ; ++t) {
        const int i = tid + 256 * t;
        if (i < cnt * 48) smb[i] = mreg[t];
    }
    if (tid < k * 24) {
        const int f = tid - 24 * (tid / 24);
        (&rout[0][0])[tid] = f == 1 ? wreg2 * rov : f == 23 ? 0.f : rov;
    }
#pragma unroll
    for (int t = 0; t < RINQ; ++t) {
        const int i = tid + 256 * t, q = i / 24, f = i - 24 * q;
        if (i < min(nri, S6_REGIN) * 24)
            (&rin[0][0])[i] = f == 0 ? __int_as_float((int)(rent[q] / (unsigned)k)) : f == 1 ? wreg2 * riv[t] : f == 23 ? 0.f : riv[t];
    }
    __syncthreads();
#ifdef DFA_S6_DEBUG
    if (a >= DFA_S6_DEBUG && a < DFA_S6_DEBUG + 256)
        for (int i = tid; i < cnt * 48; i += 256) s6_dbgm[(size_t)(a - DFA_S6_DEBUG) * S6_MAXSLOT * 48 + i] = smb[i];
#endif
    S6_TICK(tk4);
    // A QUAD of lanes per (block, row), ten blocks (240 threads) per round: lane z takes columns 2 z, 2 z + 1 of M_a S and every
    // fourth regularisation edge, the four partial rows are added by DPP.  (One thread per (block, row) — 102 of the 256 at C3 —
    // walked 112 FMAs, ~90 LDS reads and both edge loops alone: 7 of the workgroup's 42 us.)
#if DFA_S6_ABLATE & 4
    const int nfin = 1;
#else
    const int nfin = 1 + nup;
#endif
    constexpr int BPR = 10;
    float* ttile = reinterpret_cast<float*>(s6_dyn) + S6_MAXSLOT * 48;  // BPR x 36 floats behind M (the units' moments are dead)
    auto quad_sum = [](float x) __attribute__((always_inline)) {
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));  // quad_perm [1, 0, 3, 2]
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));  // quad_perm [2, 3, 0, 1]
        return x;
    };
    for (int b0 = 0; b0 < nfin; b0 += BPR) {
        const int quad = tid >> 2, z = tid & 3, lb = quad / 6, my_row = quad - 6 * lb, sidx = b0 + lb;
        const bool on  = tid < 4 * 6 * BPR && sidx < nfin;
        const int slot = on ? (sidx == 0 ? 0 : fu + sidx - 1) : 0;
        const int col  = scol[slot];
        float accr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gacc = 0.f;
        if (on) {
            // columns 2 z, 2 z + 1 of row my_row of M_a S (slot 0 of M is node a itself), then their share of (M_a S) M_b^T
            const int4* st4 = reinterpret_cast<const int4*>(&symt[z][0]);
            int sy[16];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int4 v = st4[e];
                sy[4 * e] = v.x, sy[4 * e + 1] = v.y, sy[4 * e + 2] = v.z, sy[4 * e + 3] = v.w;
            }
            const float* S0 = &accS[slot][0];
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float ma = smb[8 * my_row + p];
                v0 = fmaf(ma, S0[sy[2 * p]], v0), v1 = fmaf(ma, S0[sy[2 * p + 1]], v1);
            }
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const float2 mb = *reinterpret_cast<const float2*>(smb + 48 * slot + 8 * d + 2 * z);
                accr[d]         = fmaf(v1, mb.y, v0 * mb.x);
            }
            if (slot == 0) {
                const float2 mo2 = *reinterpret_cast<const float2*>(smb + 8 * my_row + 2 * z);
                gacc             = fmaf(mo2.y, g8s[2 * z + 1], mo2.x * g8s[2 * z]);
            }
            // regularisation edges leaving a (a -> m): rows c of e = T_a g_m - g^_m with vectors (an_c at a, -e_{3+c} at m)
            for (int q = z; q < k; q += 4) {
                const float* E = rout[q];
                const int m    = __float_as_int(E[0]);
                if (m < 0 || (slot != 0 && col != m)) continue;
                const float wt = E[1];
                for (int cc = 0; cc < 3; ++cc) {
                    const float* an = E + 5 + 6 * cc;
                    if (slot == 0) {
                        gacc -= wt * an[my_row] * E[2 + cc];
#pragma unroll
                        for (int d = 0; d < 6; ++d) accr[d] += wt * an[my_row] * an[d];
                    } else {
                        accr[3 + cc] -= wt * an[my_row];
                    }
                }
            }
            // regularisation edges arriving at a (n -> a)
            if (my_row >= 3) {
                const int cc = my_row - 3;
                for (int e = z; e < nri; e += 4) {
                    const bool staged    = e < S6_REGIN;  // (rarely not: more arriving edges than the staged ones)
                    const float* E       = rin[staged ? e : 0];
                    const unsigned entry = staged ? 0u : s.rnode_list[rib + e];
                    const int n          = staged ? __float_as_int(E[0]) : (int)(entry / (unsigned)k);
                    if (slot != 0 && col != n) continue;
                    const float wt = staged ? E[1] : wreg2 * s.rhub[entry];
                    if (slot == 0) {
                        gacc += wt * (staged ? E[2 + cc] : s.rres[3 * (size_t)entry + cc]);
                        accr[my_row] += wt;
                    } else {
#pragma unroll
                        for (int d = 0; d < 6; ++d)
                            accr[d] -= wt * (staged ? E[5 + 6 * cc + d] : s.rvec[18 * (size_t)entry + 6 * cc + d]);
                    }
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 6; ++d) accr[d] = quad_sum(accr[d]);
        gacc = quad_sum(gacc);
        if (on && z == 0) {
            if (slot == 0) {
                accr[my_row] += damping;
                s.g[6 * (size_t)a + my_row] = gacc, gsh[my_row] = gacc;
#pragma unroll
                for (int d = 0; d < 6; ++d) diag[my_row * 6 + d] = accr[d];
            }
            float2* out = reinterpret_cast<float2*>(s.bvals + ((size_t)a * s.cap + slot) * 36 + 6 * my_row);
            out[0] = make_float2(accr[0], accr[1]), out[1] = make_float2(accr[2], accr[3]), out[2] = make_float2(accr[4], accr[5]);
            // the mirror block H_ba = H_ab^T goes out row by row too (24 contiguous bytes per thread, the block's 144 from six
            // threads): transposed through LDS — written element by element it cost 4 bytes per store into somebody else's
            // cache lines (PMC: 135 MB written per launch at C3 for a 21 MB matrix)
            float* tile = ttile + 36 * lb + my_row;  // column my_row of the block's transpose
#pragma unroll
            for (int d = 0; d < 6; ++d) tile[6 * d] = accr[d];
        }
        __syncthreads();
        if (on && z == 0 && slot != 0) {
            const int rs = srs[slot];
            if (rs != 255) {
                const float* tr = ttile + 36 * lb + 6 * my_row;  // row my_row of the transposed block
                float2* out2    = reinterpret_cast<float2*>(s.bvals + ((size_t)col * s.cap + rs) * 36 + 6 * my_row);
                out2[0] = make_float2(tr[0], tr[1]), out2[1] = make_float2(tr[2], tr[3]), out2[2] = make_float2(tr[4], tr[5]);
            }
        }
        if (b0 + BPR < nfin) __syncthreads();  // (the tile is reused by the next round)
    }
    S6_TICK(tk5);
    if (tid < 6) {
        // column tid of M^-1 (= its row tid: symmetric), and with it the start of the PCG for this node:
        // x = 0, r = g, u = M^-1 g, p = s = t = 0
        float col[6];
        inv6_column(diag, s.minv + 36 * (size_t)a, tid, col);
        float u = 0.f;
#pragma unroll
        for (int d = 0; d < 6; ++d) u += col[d] * gsh[d];
        const size_t i = 6 * (size_t)a + tid;
        s.x[i] = 0.f, s.r[i] = gsh[tid], s.u[0][i] = u;
        s.p[i] = 0.f, s.s[i] = 0.f, s.t[0][i] = 0.f, s.t[1][i] = 0.f;
    }
#ifdef DFA_S6_TIMING
    if (tid == 0 && a < 16384) {
        unsigned long long* o = s6_tbuf + 16 * (size_t)a;
        const unsigned long long tk6 = clock64();
        o[0] = tk0, o[1] = tk1 - tk0, o[2] = tk_stage, o[3] = tk_prep, o[4] = tk2a - tk2, o[5] = tk_up, o[6] = tk3 - tk2, o[7] = tk4 - tk3,
        o[8] = tk5 - tk4, o[9] = tk6 - tk5, o[10] = tk6 - tk0, o[11] = wall_clock64() - wk0, o[12] = (unsigned long long)len,
        o[13] = (unsigned long long)cnt, o[14] = wk0, o[15] = ((tk2b - tk2a) << 32) | (tk2c - tk2b);
    }
#endif
}

// -------------------------------------------------------------------------------------- PCG
// Chronopoulos-Gear form of preconditioned CG: the two inner products of an iteration are taken
// together after the matrix product, so an iteration needs ONE grid-wide synchronisation — one
// kernel launch (the textbook form needs two: 22 us per iteration at 4 k nodes, launch-bound).
//   u = M^-1 r, w = A u, m = M^-1 w, t = M^-1 s (kept by the recurrence t_i = m_i + beta_i t_{i-1})
//   p_i = u_i + beta_i p_{i-1};  s_i = w_i + beta_i s_{i-1};  x += alpha_i p_i;  r -= alpha_i s_i
//   u_{i+1} = u_i - alpha_i t_i;  w_{i+1} = A u_{i+1};  gamma = (r, u), delta = (w, u)
//   beta_{i+1} = gamma_{i+1} / gamma_i;  alpha_{i+1} = gamma_{i+1} / (delta_{i+1} - beta_{i+1} gamma_{i+1} / alpha_i)
// The matrix product of launch i needs u_{i+1} of OTHER nodes, which their waves are only computing in
// the same launch: the gather rebuilds it from u_i, m_i, t_{i-1} and the two scalars (18 floats per
// neighbour block next to the block's own 36).  Same iterates as textbook PCG in exact arithmetic.
// launch `it` = -1: w_0 = A u_0, m_0, gamma_0, delta_0.   launch it >= 0: iteration `it` as above.
#ifndef DFA_S6_PCG_WAVES
#define DFA_S6_PCG_WAVES 5
#endif
__global__ __launch_bounds__(64 * S6_NODES_PER_BLOCK, DFA_S6_PCG_WAVES) void s6_pcg_step_kernel(Solve6View s, Solve6State* st, int it) {
    __shared__ float stage[S6_NODES_PER_BLOCK][3][64];
    __shared__ float gd_sh[S6_NODES_PER_BLOCK][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a    = blockIdx.x * S6_NODES_PER_BLOCK + wave;
    const int cur = it >= 0 ? (it & 1) : 0, nxt = cur ^ 1;  // u, m: read [cur], write [nxt]; t: read [nxt], write [cur]
    const float* ucur  = s.u[cur];
    const float* mcur  = s.m[cur];
    const float* tprev = s.t[nxt];
    // The kernel is a chain of dependent memory round trips (~2 us each) and little else, so it is
    // written as two rounds of loads: (1) everything addressable from the launch arguments — the
    // stop flag, the partial inner products and scalars of the previous launch, this row's columns,
    // its M^-1 and its own vector entries — and (2) the gathers through the columns.  The three
    // products A u, A m, A t are gathered separately and w = A u - alpha (A m + beta A t) is formed at
    // the end, so round (2) does not wait for the scalars either.
    constexpr int MAXIT = 5;  // 10 slots per pass, plan capacity 48
    // The partial inner products of the previous launch (one pair per workgroup) are read ONCE per workgroup — thread t takes
    // partials t, t + 512, ... —, added over the wave, and the eight waves' sums go through LDS: the same value, in the same
    // order, in every wave of every workgroup.  (Every wave used to read all of them: 2 x 16 registers per lane, which kept
    // the kernel at 92 VGPRs — two of these 512-thread workgroups per CU.)
    constexpr int MAXP  = 4;  // partials per thread held in registers: 512 x 4 = 2048 workgroups = 16 384 nodes (more: a loop)
    constexpr int NT    = 64 * S6_NODES_PER_BLOCK;
    const int nb   = s6_matvec_blocks(s.D);
    const int done = st->pcg_done;
    float gp[MAXP], dp[MAXP];
    float gamma_prev = 1.f, alpha_prev = 1.f, rz0 = 0.f;
    float tol2 = st->tol2;
    if (it >= 0) {
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            const int i = (int)threadIdx.x + NT * q;
            gp[q] = i < nb ? s.g_part[it & 1][i] : 0.f;
            dp[q] = i < nb ? s.d_part[it & 1][i] : 0.f;
        }
        for (int i = (int)threadIdx.x + NT * MAXP; i < nb; i += NT) gp[MAXP - 1] += s.g_part[it & 1][i], dp[MAXP - 1] += s.d_part[it & 1][i];
        if (it > 0) gamma_prev = st->gamma_prev[(it + 1) & 1], alpha_prev = st->alpha_prev[(it + 1) & 1], rz0 = st->rz0;
    }
    const int ss = lane / 6, c = lane - 6 * ss;
    int cols[MAXIT];
#pragma unroll
    for (int q = 0; q < MAXIT; ++q) {
        const int sl = ss + 10 * q;
        cols[q]      = (a < s.D && lane < 60 && sl < s.cap) ? s.bcols[(size_t)a * s.cap + sl] : -1;
    }
    float minv_row[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, own_u = 0.f, own_m = 0.f, own_t = 0.f, own_p = 0.f, own_s = 0.f,
          own_w = 0.f, own_r = 0.f, own_x = 0.f;
    if (a < s.D && lane < 6) {
        const size_t i = 6 * (size_t)a + lane;
#pragma unroll
        for (int d = 0; d < 6; ++d) minv_row[d] = s.minv[36 * (size_t)a + 6 * lane + d];
        own_u = ucur[i], own_r = s.r[i];
        if (it >= 0) own_m = mcur[i], own_t = tprev[i], own_p = s.p[i], own_s = s.s[i], own_w = s.w[i], own_x = s.x[i];
    }
    if (done) return;  // (uniform)
    // (the partials arrive with the first round of loads, before the columns the gathers need anyway)
    __shared__ float gd_wave[S6_NODES_PER_BLOCK][2];
    if (it >= 0) {
        float gsum = 0.f, dsum = 0.f;
#pragma unroll
        for (int q = 0; q < MAXP; ++q) gsum += gp[q], dsum += dp[q];
        gsum = wave_sum_all(gsum), dsum = wave_sum_all(dsum);
        if (lane == 0) gd_wave[wave][0] = gsum, gd_wave[wave][1] = dsum;
    }
    float au = 0.f, am = 0.f, at = 0.f;
#pragma unroll
    for (int q = 0; q < MAXIT; ++q) {
        if (cols[q] < 0) continue;
        const float2* H  = reinterpret_cast<const float2*>(s.bvals + ((size_t)a * s.cap + ss + 10 * q) * 36 + 6 * c);
        const size_t cb  = 6 * (size_t)cols[q];
        const float2* pu = reinterpret_cast<const float2*>(ucur + cb);
        const float2* pm = reinterpret_cast<const float2*>(mcur + cb);
        const float2* pt = reinterpret_cast<const float2*>(tprev + cb);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float2 h = H[d], uu = pu[d];
            au += h.x * uu.x, au += h.y * uu.y;
            if (it >= 0) {
                const float2 mm = pm[d], tt = pt[d];
                am += h.x * mm.x, am += h.y * mm.y;
                at += h.x * tt.x, at += h.y * tt.y;
            }
        }
    }
    // scalars of this iteration
    float alpha = 0.f, beta = 0.f;
    if (it >= 0) {
        __syncthreads();  // (uniform: `it` and `done` are the same everywhere)
        float gamma = 0.f, delta = 0.f;
#pragma unroll
        for (int w = 0; w < S6_NODES_PER_BLOCK; ++w) gamma += gd_wave[w][0], delta += gd_wave[w][1];
        float denom = delta;
        if (it > 0) {
            beta = gamma / gamma_prev;
            denom -= beta * gamma / alpha_prev;
        } else {
            rz0 = gamma;
            // Eisenstat-Walker forcing term (choice 2, alpha = 2): the tolerance of this PCG from the gradients of this and
            // of the previous linearisation — the same value in every workgroup; one thread records it for the launches
            // that follow
            const int slot   = st->ew_slot & 1;
            const float prev = st->rz0_gn[slot ^ 1];
            const float eg   = st->ew_gamma;
            if (eg > 0.f && prev > 0.f) {
                const float eta = eg * gamma / prev;
                tol2            = fminf(fmaxf(eta * eta, st->ew_min2), st->ew_max2);
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                st->rz0_gn[slot] = gamma;
                if (eg > 0.f && prev > 0.f) {
                    st->tol2 = tol2;
                    const int h = st->cur;
                    if (h >= 0 && h < S6_HIST) st->pcg_tol_hist[h] = sqrtf(tol2);
                }
            }
        }
        // converged, or breakdown: the same decision in every workgroup
        if (!(gamma > 0.f) || gamma <= tol2 * rz0 || !(denom > 0.f)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                st->pcg_done = 1;
                const int h = st->cur;
                if (h >= 0 && h < S6_HIST) st->pcg_rel_hist[h] = gamma > 0.f && rz0 > 0.f ? sqrtf(gamma / rz0) : 0.f;
            }
            return;
        }
        alpha = gamma / denom;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->gamma_prev[it & 1] = gamma, st->alpha_prev[it & 1] = alpha;
            if (it == 0) st->rz0 = gamma;
            st->pcg_iters += 1;
            st->pcg_last_it = it + 1;
            const int h = st->cur;
            if (h >= 0 && h < S6_HIST) st->pcg_it_hist[h] = it + 1;
        }
    }
    stage[wave][0][lane] = au, stage[wave][1][lane] = am, stage[wave][2][lane] = at;
    __syncthreads();
    float gpart = 0.f, dpart = 0.f;
    if (a < s.D) {
        float wn = 0.f;
        if (lane < 6) {
            float su = 0.f, sm = 0.f, stt = 0.f;
#pragma unroll
            for (int q = 0; q < 10; ++q)
                su += stage[wave][0][q * 6 + lane], sm += stage[wave][1][q * 6 + lane], stt += stage[wave][2][q * 6 + lane];
            wn = it >= 0 ? su - alpha * (sm + beta * stt) : su;
        }
        // m_new = M^-1 w_new needs the six components held by lanes 0..5
        float mn = 0.f;
#pragma unroll
        for (int d = 0; d < 6; ++d) mn += minv_row[d] * __shfl(wn, d, 64);
        if (lane < 6) {
            const size_t i = 6 * (size_t)a + lane;
            float un = own_u, rn = own_r;
            if (it >= 0) {
                const float tn = own_m + beta * own_t;
                const float pn = own_u + beta * own_p;
                const float sn = own_w + beta * own_s;
                un = own_u - alpha * tn;
                rn = own_r - alpha * sn;
                s.t[cur][i] = tn, s.p[i] = pn, s.s[i] = sn;
                s.x[i] = own_x + alpha * pn;
                s.r[i] = rn;
                s.u[nxt][i] = un;
            }
            s.w[i] = wn;
            s.m[it >= 0 ? nxt : 0][i] = mn;
            gpart = rn * un, dpart = wn * un;
        }
    }
    gpart = wave_sum_all(gpart), dpart = wave_sum_all(dpart);
    if (lane == 0) gd_sh[wave][0] = gpart, gd_sh[wave][1] = dpart;
    __syncthreads();
    if (threadIdx.x == 0) {
        float g = 0.f, d = 0.f;
        for (int w = 0; w < S6_NODES_PER_BLOCK; ++w) g += gd_sh[w][0], d += gd_sh[w][1];
        const int slot = it >= 0 ? ((it + 1) & 1) : 0;
        s.g_part[slot][blockIdx.x] = g, s.d_part[slot][blockIdx.x] = d;
    }
}

// ------------------------------------------------------------------------------------ update
__global__ __launch_bounds__(256) void s6_update_kernel(Solve6View s, Solve6State* st, int launched, int linear_iter,
                                                        int* __restrict__ mirror, int h, int apply) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    // Gauss-Newton control (written by the linearisation's last workgroup, by no thread of this launch): 0 = apply the step; 1 = the outer iteration
    // has ended, nothing moves; 2 = ended by a rejected step: the transforms before that step come back (every update
    // launch until the next outer iteration does this again: the same values)
    const int stop = st->gn_stop;
    if (blockIdx.x == 0 && threadIdx.x < 64 && apply) {
        // a PCG that used every launch it was given: its last (r, u) is still in the per-workgroup partials of the last launch
        const bool ran = !st->pcg_done && launched > 0 && st->pcg_last_it == launched;  // (uniform)
        bool reached   = !ran;
        if (ran) {
            const int nb = s6_matvec_blocks(s.D);
            float g = 0.f;
            for (int i = threadIdx.x; i < nb; i += 64) g += s.g_part[launched & 1][i];
            g = wave_sum_all(g);
            const float rz0 = st->rz0;
            reached         = !(g > st->tol2 * rz0);
            if (threadIdx.x == 0 && h >= 0 && h < S6_HIST) st->pcg_rel_hist[h] = g > 0.f && rz0 > 0.f ? sqrtf(g / rz0) : 0.f;
        }
        if (threadIdx.x == 0) {
            st->cost = 0.0, st->valid = 0ull;  // accumulators of the next linearisation
            const bool cut = !reached && launched < linear_iter;  // stopped by the plan's prediction, not by the caller's cap
            if (cut) st->pcg_short += 1;
            // (an iteration without a PCG — the outer iteration had ended — tells the launch budget so: S6_MIRROR_SKIPPED)
            if (mirror && h >= 0 && h < S6_HIST) mirror[h] = stop ? S6_MIRROR_SKIPPED : cut ? -st->pcg_last_it : st->pcg_last_it;
        }
    }
    if (n >= s.D) return;
    if (stop == 2) {
        const DQ q = dq_load(s.dq_prev + 8 * (size_t)n);
        dq_store(s.dq + 8 * (size_t)n, q);
        s6_node_now(s, n, q);
        return;
    }
    if (stop || !apply) return;
    const float* tw = s.x + 6 * (size_t)n;
    const DQ q      = dq_load(s.dq + 8 * (size_t)n);
    dq_store(s.dq_prev + 8 * (size_t)n, q);
    const f3 gh     = mk3(s.ghat[3 * n], s.ghat[3 * n + 1], s.ghat[3 * n + 2]);
    const float th  = sqrtf(tw[0] * tw[0] + tw[1] * tw[1] + tw[2] * tw[2]);
    const float sc  = th > 1e-6f ? sinf(0.5f * th) / th : 0.5f;
    const Quat qo   = Quat{cosf(0.5f * th), sc * tw[0], sc * tw[1], sc * tw[2]};
    Quat rn         = qmul(qo, q.r);
    const f3 t0     = qvec(qmul(q.d, qconj(q.r)));
    const f3 tc     = mk3(2.f * t0.x - gh.x, 2.f * t0.y - gh.y, 2.f * t0.z - gh.z);
    const f3 tr     = qvec(qmul(qmul(qo, pureq(tc)), qconj(qo)));
    const f3 t      = mk3(tr.x + gh.x + tw[3], tr.y + gh.y + tw[4], tr.z + gh.z + tw[5]);
    rn              = qscale(rn, 1.f / sqrtf(qdot(rn, rn)));
    const Quat dn   = qscale(qmul(pureq(t), rn), 0.5f);
    dq_store(s.dq + 8 * (size_t)n, DQ{rn, dn});
    s6_node_now(s, n, DQ{rn, dn});  // for the next linearisation
}

template <int K>
__global__ __launch_bounds__(256) void s6_warp_kernel(Solve6View s, const float* __restrict__ dq,
                                                      float* __restrict__ out_v, float* __restrict__ out_n) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= s.N) return;
    int32_t idx[K];
    float wn[K];
    if (s.k == K) {  // (uniform) 16-byte loads, as the linearisation
#pragma unroll
        for (int q = 0; q < K / 4; ++q) {
            const int4 iv   = reinterpret_cast<const int4*>(s.idx + (size_t)v * K)[q];
            const float4 wv = reinterpret_cast<const float4*>(s.wn + (size_t)v * K)[q];
            idx[4 * q] = iv.x, idx[4 * q + 1] = iv.y, idx[4 * q + 2] = iv.z, idx[4 * q + 3] = iv.w;
            wn[4 * q] = wv.x, wn[4 * q + 1] = wv.y, wn[4 * q + 2] = wv.z, wn[4 * q + 3] = wv.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            idx[j] = j < s.k ? s.idx[(size_t)v * s.k + j] : -1;
            wn[j]  = j < s.k ? s.wn[(size_t)v * s.k + j] : 0.f;
        }
    }
    Blend<K> B;
    blend<K>(dq, idx, wn, s.k, B);
    const f3 c = mk3(s.canon[3 * (size_t)v], s.canon[3 * (size_t)v + 1], s.canon[3 * (size_t)v + 2]);
    const f3 p = B.m > 0.f ? blend_point<K>(B, c) : c;
    const size_t o = s.vperm[v];  // the caller's index of the solver's vertex v
    out_v[3 * o] = p.x, out_v[3 * o + 1] = p.y, out_v[3 * o + 2] = p.z;
    if (out_n && s.canon_n) {
        const f3 n0 = mk3(s.canon_n[3 * (size_t)v], s.canon_n[3 * (size_t)v + 1], s.canon_n[3 * (size_t)v + 2]);
        const f3 n  = B.m > 0.f ? blend_normal<K>(B, n0) : n0;
        out_n[3 * o] = n.x, out_n[3 * o + 1] = n.y, out_n[3 * o + 2] = n.z;
    }
}

__global__ __launch_bounds__(256) void s6_begin_kernel(Solve6View s, Solve6State* st, const float* __restrict__ node_dq, int slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S6_HIST) {  // gn_tol > 0: every slot the solve enqueues starts as "skipped"; the ones that run overwrite theirs
        st->cost_hist[i] = 0.0, st->valid_hist[i] = 0u, st->pcg_it_hist[i] = 0, st->pcg_rel_hist[i] = 0.f, st->pcg_tol_hist[i] = 0.f;
        st->stop_hist[i] = i < slots ? 3 : 0;
    }
    if (i <= S6_LIN_SHARDS) st->lin_ticket[i] = 0u;
    if (i == 0) {
        st->cost = 0.0, st->initial_cost = 0.0, st->final_cost = 0.0;
        st->valid = st->valid_first = st->valid_last = 0ull;
        st->have_first = 0, st->gn_iters = 0, st->pcg_iters = 0;  // overflow / max_row_blocks belong to the pattern
        st->pcg_done = 0, st->rz0 = 0.f, st->pcg_short = 0, st->rz0_gn[0] = st->rz0_gn[1] = 0.f;
        st->cur = 0, st->hist_n = min(slots, S6_HIST), st->gn_stop = 0, st->gn_solves = 0, st->gn_rejected = 0, st->gn_converged = 0;
        st->cost_ref = 0.0, st->valid_ref = 0ull;
    }
    if (i < 8 * s.D) s.dq[i] = node_dq[i], s.dq_prev[i] = node_dq[i];
    if (i < s.N) s.rho[i] = 0.f;
    if (i < s.D * s.k) s.rhub[i] = 1.f;
    if (i < s.D) s6_node_now(s, i, dq_load(node_dq + 8 * (size_t)i));
}

// kfusion::device::computePointNormals (src/kfusion/cuda/imgproc.cu:187-215)
__global__ __launch_bounds__(256) void points_normals_kernel(const uint16_t* __restrict__ depth, int depth_step, int cols,
                                                             int rows, float finvx, float finvy, float cx, float cy,
                                                             float* __restrict__ points, int points_step,
                                                             float* __restrict__ normals, int normals_step) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= cols || y >= rows) return;
    const float qnan = __builtin_nanf("");
    float4 P = make_float4(qnan, qnan, qnan, qnan), Nn = P;  // :195-196
    if (x < cols - 1 && y < rows - 1) {                      // :198
        const uint16_t* d0 = (const uint16_t*)((const char*)depth + (size_t)y * depth_step);
        const uint16_t* d1 = (const uint16_t*)((const char*)depth + (size_t)(y + 1) * depth_step);
        const float z00 = d0[x] * 0.001f, z01 = d0[x + 1] * 0.001f, z10 = d1[x] * 0.001f;
        if (z00 * z01 * z10 != 0.f) {  // :206
            // Reprojector device.hpp:50-54: x = z * (u - cx) * finv.x
            const f3 v00 = mk3(z00 * ((float)x - cx) * finvx, z00 * ((float)y - cy) * finvy, z00);
            const f3 v01 = mk3(z01 * ((float)(x + 1) - cx) * finvx, z01 * ((float)y - cy) * finvy, z01);
            const f3 v10 = mk3(z10 * ((float)x - cx) * finvx, z10 * ((float)(y + 1) - cy) * finvy, z10);
            const f3 n   = normalized(cross(v01 - v00, v10 - v00));
            Nn = make_float4(-n.x, -n.y, -n.z, 0.f);  // :212
            P  = make_float4(v00.x, v00.y, v00.z, 0.f);
        }
    }
    *reinterpret_cast<float4*>((char*)points + (size_t)y * points_step + 16 * (size_t)x)   = P;
    *reinterpret_cast<float4*>((char*)normals + (size_t)y * normals_step + 16 * (size_t)x) = Nn;
}

}  // namespace

#define K6DISPATCH(kernel, k, ...)            \
    do {                                      \
        if ((k) <= 4) kernel<4> __VA_ARGS__;  \
        else kernel<8> __VA_ARGS__;           \
    } while (0)

namespace {
__global__ void s6_pattern_reset_kernel(Solve6State* st) { st->overflow = 0, st->max_row_blocks = 0; }
}  // namespace

hipError_t s6_build_graph(const Solve6View& s, Solve6State* state, const float* canon_user, const float* canon_n_user,
                          const float* raw_w, const int32_t* raw_reg, int kreg, hipStream_t st) {
    s6_pattern_reset_kernel<<<1, 1, 0, st>>>(state);
    hipError_t e = hipSuccess;
    if (s.N > 0) {
        // the solver's vertex order: by nearest node (counting sort), by index inside a node
        s6_near_kernel<<<(s.N + 255) / 256, 256, 0, st>>>(s.idx_nat, s.N, s.k, s.D, s.near);
        e = solve_transpose_graph(s.near, (size_t)s.N, s.D, s.blk_hist, s.vptr, s.vlist, st);
        if (e != hipSuccess) return e;
        K6DISPATCH(s6_permute_kernel, s.k, <<<s.D, 256, 0, st>>>(s, canon_user, canon_n_user, raw_w));
    }
    s6_reg_graph_kernel<<<(s.D + 255) / 256, 256, 0, st>>>(raw_reg, s.D, kreg, s.k, s.reg_idx);
    e = solve_transpose_graph(s.idx, (size_t)s.N * s.k, s.D, s.blk_hist, s.node_ptr, s.node_list, st);
    if (e != hipSuccess) return e;
    e = solve_transpose_graph(s.reg_idx, (size_t)s.D * s.k, s.D, s.blk_hist, s.rnode_ptr, s.rnode_list, st);
    if (e != hipSuccess) return e;
    s6_pattern_kernel<<<s.D, 256, 0, st>>>(s, state);
    s6_rslot_kernel<<<(s.D * s.cap + 255) / 256, 256, 0, st>>>(s);
    return hipGetLastError();
}

hipError_t s6_begin(const Solve6View& s, Solve6State* state, const float* node_dq, int slots, hipStream_t st) {
    const int n = std::max(std::max(std::max(8 * s.D, s.N), s.D * s.k), S6_HIST + S6_LIN_SHARDS);
    s6_begin_kernel<<<(n + 255) / 256, 256, 0, st>>>(s, state, node_dq, slots);
    return hipGetLastError();
}

hipError_t s6_linearise(const Solve6View& s, Solve6State* state, const Solve6Image& img, const Solve6Params& p,
                        int update_weights, int gi, int gn_in_outer, int closing, hipStream_t st) {
    // (g^, M of the nodes and the cleared cost accumulators come from the launch that produced the transforms:
    // s6_begin / s6_update)
    const float wreg2 = p.lambda / ((float)s.D * (float)s.k);
    const int nlin = (s.N + 255) / 256, nreg = (s.D * s.k + 255) / 256;
    // gn_tol > 0: a linearisation that does not open an outer iteration is skipped once that iteration has ended
    const int gate = p.gn_tol > 0.f && !update_weights;
    const S6Decide dec{p.gn_tol > 0.f ? 1 : 0, gi, gn_in_outer, closing, p.gn_tol, s6_forcing(p, gn_in_outer)};
    K6DISPATCH(s6_linearise_kernel, s.k, <<<nlin + nreg, 256, 0, st>>>(s, state, img, p, update_weights, nlin, wreg2, gate, dec));
    return hipGetLastError();
}

S6Forcing s6_forcing(const Solve6Params& p, int gn_in_outer) {
    S6Forcing f{p.pcg_tol * p.pcg_tol, 0.f, 0.f, 0.f, gn_in_outer & 1, p.gn_tol > 0.f ? 1 : 0};
    if (p.pcg_tol_first > 0.f && p.pcg_tol_adapt > 0.f) {
        // adaptive: the first iteration of an outer iteration at pcg_tol_first (no previous gradient under these weights),
        // the others decided on the device; ew_gamma = 0 keeps the tolerance given here
        const float hi = std::max(p.pcg_tol_first, p.pcg_tol);
        f.tol2 = hi * hi, f.ew_min2 = p.pcg_tol * p.pcg_tol, f.ew_max2 = hi * hi;
        f.ew_gamma = gn_in_outer > 0 ? p.pcg_tol_adapt : 0.f;
    } else if (p.pcg_tol_first > 0.f) {
        float e = p.pcg_tol_first;
        for (int i = 0; i < gn_in_outer; ++i) e *= p.pcg_tol_decay;
        const float eta = std::max(p.pcg_tol, e);
        f.tol2 = eta * eta;
    }
    return f;
}

hipError_t s6_assemble(const Solve6View& s, Solve6State* state, const Solve6Params& p, int gn_in_outer, hipStream_t st) {
    const float wreg2 = p.lambda / ((float)s.D * (float)s.k);
    const S6Forcing f = s6_forcing(p, gn_in_outer);
    {
        // rows staged per pass (development builds: DFA_S6_RC for A/B): what fits in 28 KiB — with the static arrays 36 KiB,
        // four workgroups per CU
        static const int rc_env = dev_env_int("DFA_S6_RC", 0);
        const int rc = rc_env ? rc_env : (s.k <= 4 ? 448 : 352);
#define S6A2(KK, RC)                                                                                              \
    do {                                                                                                          \
        if (s.k == (KK)) S6A2X(KK, RC, true);                                                                     \
        else S6A2X(KK, RC, false);                                                                                \
    } while (0)
#define S6A2X(KK, RC, EX)                                                                                         \
    do {                                                                                                          \
        constexpr size_t sh = s6_assemble_lds(KK, RC);                                                            \
        if (sh > 48 * 1024) {                                                                                     \
            const hipError_t ae = allow_dynamic_lds((const void*)s6_assemble2_kernel<KK, RC, EX>, (int)sh);         \
            if (ae != hipSuccess) return ae;                                                                      \
        }                                                                                                         \
        s6_assemble2_kernel<KK, RC, EX><<<s.D + 1, 256, sh, st>>>(s, state, wreg2, p.damping, f, dev_env_int("DFA_XCD_MAP", 0)); \
    } while (0)
        if (s.k <= 4) {
            if (rc <= 320) S6A2(4, 320);
            else S6A2(4, 448);
        } else {
            if (rc <= 256) S6A2(8, 256);
            else S6A2(8, 352);
        }
#undef S6A2
#undef S6A2X
    }
    return hipGetLastError();
}

hipError_t s6_pcg(const Solve6View& s, Solve6State* state, const Solve6Params& p, hipStream_t st) {
    const int mb = s6_matvec_blocks(s.D);
    // launch -1 forms w_0 = A u_0; launch it >= 0 is iteration it (x_{it+1} is complete when it returns); the stop test
    // reads the tolerance of this Gauss-Newton iteration from the state block (set by the assembly launch)
    for (int it = -1; it < p.linear_iter; ++it) s6_pcg_step_kernel<<<mb, 64 * S6_NODES_PER_BLOCK, 0, st>>>(s, state, it);
    return hipGetLastError();
}

hipError_t s6_pcg_n(const Solve6View& s, Solve6State* state, int launches, hipStream_t st) {
    Solve6Params p{};
    p.linear_iter = launches;
    return s6_pcg(s, state, p, st);
}

hipError_t s6_update(const Solve6View& s, Solve6State* state, int launched, int linear_iter, int* mirror, int gi, int apply,
                     hipStream_t st) {
    s6_update_kernel<<<(s.D + 255) / 256, 256, 0, st>>>(s, state, launched, linear_iter, mirror, gi, apply);
    return hipGetLastError();
}

hipError_t s6_warp(const Solve6View& s, const float* dq, float* out_v, float* out_n, hipStream_t st) {
    if (s.N == 0) return hipSuccess;
    K6DISPATCH(s6_warp_kernel, s.k, <<<(s.N + 255) / 256, 256, 0, st>>>(s, dq, out_v, out_n));
    return hipGetLastError();
}

hipError_t launch_points_normals(const uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                                 float cy, float* points, int points_step, float* normals, int normals_step,
                                 hipStream_t st) {
    dim3 grid((cols + 31) / 32, (rows + 7) / 8);
    points_normals_kernel<<<grid, 256, 0, st>>>(depth, depth_step, cols, rows, 1.f / fx, 1.f / fy, cx, cy, points,
                                                points_step, normals, normals_step);
    return hipGetLastError();
}

}  // namespace dfa
