// warp.hip — gfx950 kernels for the warp-field seam: exact k-NN of the deformation nodes,
// RBF transformation weights, the reference's ordered dual-quaternion "blend" and warpToLive.
//
// Reference semantics: src/dynfu/warp_field.cpp:99-171 (nanoflann KD-tree k-NN per vertex on
// the CPU, three std::vector allocations per query) and src/dynfu/utils/node.cpp:29-36.
//
// MI355X design.  The result contract is "the k nodes with the smallest (squared distance,
// node index) pairs, ascending" — what nanoflann returns up to the order of exact ties.  Two
// exact searches produce it:
//   * brute force (small problems): one lane per query, node positions staged through LDS in
//     1024-node tiles (a wave reads one node per step -> LDS broadcast), the k best kept in
//     registers as a sorted list updated by an unrolled compare-exchange chain;
//   * uniform grid (n_query * D large): nodes are bucketed by a counting sort into at most 32^3
//     cells sized ~2 node spacings (the grid geometry is computed ON the device from the node
//     bounding box, no host round trip); a query walks Chebyshev shells of cells around its own
//     cell and stops once its k-th distance is below the distance to the next shell.  At
//     262 k vertices x 2 k nodes that is ~50 candidates per query instead of 2048.
// Both use the same distance expression (nanoflann L2_Simple_Adaptor order of operations) so
// the neighbour lists are bit-identical to the CPU oracle's.
#include <hip/hip_runtime.h>

#include "dq_device.hpp"
#include "dev_switch.hpp"
#include "kernels.hpp"

namespace dfa {

constexpr int KNN_TILE = 1024;

// sorted (ascending by (distance, index)) list of the K nearest candidates.  An entry is ONE 64-bit key — the bits of the
// (non-negative) squared distance above the node index — so that "(d, i) before (d', i')" is one unsigned compare and an
// exchange is a min / max pair: the 8-NN search of 1.08 M vertices is bound by vector-ALU issue, and the insertion is most of
// what it issues (profiles/r05_sq_hostseq_ref.md).  The order of non-negative floats is the order of their bits; +inf
// (empty) sorts behind every finite distance, a NaN distance (a vertex with NaN coordinates) behind +inf: never inserted.
template <int K>
struct KnnList {
    unsigned long long key[K];
    static constexpr unsigned long long EMPTY = 0x7f8000007fffffffull;  // (+inf, index 0x7fffffff)
    __device__ __forceinline__ static unsigned long long pack(float dist, int idx) {
        return ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned int)idx;
    }
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) key[j] = EMPTY;
    }
    __device__ __forceinline__ void push(float dist, int idx) {
        const unsigned long long x = pack(dist, idx);
        if (x < key[K - 1]) {
            key[K - 1] = x;
#pragma unroll
            for (int j = K - 1; j > 0; --j) {
                const unsigned long long a = key[j - 1], b = key[j];
                key[j - 1] = a < b ? a : b, key[j] = a < b ? b : a;
            }
        }
    }
    __device__ __forceinline__ float dist(int j) const { return __uint_as_float((unsigned int)(key[j] >> 32)); }
    __device__ __forceinline__ int raw_index(int j) const { return (int)(unsigned int)key[j]; }  // 0x7fffffff: empty
    __device__ __forceinline__ int index(int j) const { return raw_index(j) == 0x7fffffff ? -1 : raw_index(j); }
    __device__ __forceinline__ void set_index(int j, int n) { key[j] = pack(__builtin_inff(), n < 0 ? 0x7fffffff : n); }
};

// L2_Simple_Adaptor::evalMetric (nanoflann.hpp:338-345): ((0 + d0^2) + d1^2) + d2^2
__device__ __forceinline__ float dist2(f3 q, float gx, float gy, float gz) {
    const float d0 = q.x - gx, d1 = q.y - gy, d2 = q.z - gz;
    return (d0 * d0 + d1 * d1) + d2 * d2;
}

// ---------------------------------------------------------------------------- brute force
// scans all D nodes (block-cooperative LDS staging); every thread of the block must call it
template <int K>
__device__ __forceinline__ void knn_scan(const float* __restrict__ node_pos, int D, f3 q, KnnList<K>& best,
                                         float4* tile) {
    best.init();
    for (int base = 0; base < D; base += KNN_TILE) {
        const int n = min(KNN_TILE, D - base);
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const float* p = node_pos + 3 * (size_t)(base + j);
            tile[j]        = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const float4 g = tile[j];
            best.push(dist2(q, g.x, g.y, g.z), base + j);
        }
    }
}

// ------------------------------------------------------------------------------ uniform grid
__global__ __launch_bounds__(1024) void grid_setup_kernel(const float* __restrict__ node_pos, int D,
                                                          KnnGridDesc* __restrict__ desc,
                                                          int32_t* __restrict__ cell_count) {
    __shared__ float smin[3][16], smax[3][16];
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = threadIdx.x; i < D; i += blockDim.x)
        for (int c = 0; c < 3; ++c) {
            const float v = node_pos[3 * (size_t)i + c];
            mn[c] = fminf(mn[c], v), mx[c] = fmaxf(mx[c], v);
        }
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
        if ((threadIdx.x & 63) == 0) smin[c][threadIdx.x >> 6] = mn[c], smax[c][threadIdx.x >> 6] = mx[c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KNN_GRID_MAX_CELLS; i += blockDim.x) cell_count[i] = 0;
    if (threadIdx.x == 0) {
        float ext[3];
        for (int c = 0; c < 3; ++c) {
            float a = smin[c][0], b = smax[c][0];
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) a = fminf(a, smin[c][w]), b = fmaxf(b, smax[c][w]);
            desc->bmin[c] = a;
            ext[c]        = fmaxf(b - a, 0.f);
        }
        const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
        // ~2 node spacings for nodes on a surface; never more than 32 cells per axis
        float cs = cbrtf((ext[0] * ext[1] * ext[2]) / (float)D);
        cs       = fmaxf(cs, emax / (float)KNN_GRID_MAX_DIM);
        if (!(cs > 0.f)) cs = 1.f;  // all nodes coincide
        desc->cs     = cs;
        desc->inv_cs = 1.f / cs;
        for (int c = 0; c < 3; ++c) {
            int n = (int)(ext[c] * desc->inv_cs) + 1;
            desc->dim[c] = min(max(n, 1), KNN_GRID_MAX_DIM);
        }
    }
}

__device__ __forceinline__ void cell_of(const KnnGridDesc& g, f3 p, int& cx, int& cy, int& cz) {
    // clamped: queries outside the node bounding box are projected onto it (the projection is
    // never farther from any node than the query itself, so shell bounds stay valid)
    cx = min(max((int)floorf((p.x - g.bmin[0]) * g.inv_cs), 0), g.dim[0] - 1);
    cy = min(max((int)floorf((p.y - g.bmin[1]) * g.inv_cs), 0), g.dim[1] - 1);
    cz = min(max((int)floorf((p.z - g.bmin[2]) * g.inv_cs), 0), g.dim[2] - 1);
}

__global__ __launch_bounds__(256) void grid_count_kernel(const float* __restrict__ node_pos, int D,
                                                         const KnnGridDesc* __restrict__ desc,
                                                         int32_t* __restrict__ cell_count,
                                                         int32_t* __restrict__ node_cell) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const KnnGridDesc g = *desc;
    int cx, cy, cz;
    cell_of(g, mk3(node_pos[3 * (size_t)i], node_pos[3 * (size_t)i + 1], node_pos[3 * (size_t)i + 2]), cx, cy, cz);
    const int c  = cx + g.dim[0] * (cy + g.dim[1] * cz);
    node_cell[i] = c;
    atomicAdd(&cell_count[c], 1);
}

// exclusive scan of KNN_GRID_MAX_CELLS counts by one workgroup (32 consecutive cells per thread)
__global__ __launch_bounds__(1024) void grid_scan_kernel(int32_t* __restrict__ cell_count /* in: counts, out: cursors */,
                                                         int32_t* __restrict__ cell_start) {
    __shared__ int32_t wave_tot[16];
    constexpr int PER = KNN_GRID_MAX_CELLS / 1024;
    const int base    = threadIdx.x * PER;
    int loc[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) loc[j] = cell_count[base + j], sum += loc[j];
    int incl        = sum;
    const int lane  = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = incl - sum;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        cell_start[base + j] = off;
        cell_count[base + j] = off;  // running cursor for the fill pass
        off += loc[j];
    }
    if (threadIdx.x == 1023) cell_start[KNN_GRID_MAX_CELLS] = off;
}

__global__ __launch_bounds__(256) void grid_fill_kernel(const float* __restrict__ node_pos, int D,
                                                        const int32_t* __restrict__ node_cell,
                                                        int32_t* __restrict__ cursor,
                                                        float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const int slot = atomicAdd(&cursor[node_cell[i]], 1);
    sorted[slot]   = make_float4(node_pos[3 * (size_t)i], node_pos[3 * (size_t)i + 1], node_pos[3 * (size_t)i + 2],
                                 __int_as_float(i));
}

// The same grid for a deformation-node set (D <= GRID_ONE_MAX nodes) built by ONE workgroup: bounding box, geometry,
// cell histogram, exclusive scan and the counting-sort fill all in LDS (the 32^3 counters are 128 KiB) — one launch of
// ~8 us instead of four (setup 6 + count 5 + scan 13 + fill 5 us and three kernel boundaries at C2, twice per frame in
// the pipelined schedule).  Same desc / cell_start / sorted as the four-kernel path (order inside a cell is that of the
// atomics in both: the searches order candidates by (distance, index) themselves).
constexpr int GRID_ONE_MAX = 8192;
__global__ __launch_bounds__(1024) void grid_build_one_kernel(const float* __restrict__ node_pos, int D,
                                                              KnnGridDesc* __restrict__ desc,
                                                              int32_t* __restrict__ cell_start,
                                                              float4* __restrict__ sorted) {
    extern __shared__ int32_t cnt[];  // KNN_GRID_MAX_CELLS counters, then cursors
    __shared__ float smin[3][16], smax[3][16];
    __shared__ int32_t wave_tot[16];
    __shared__ KnnGridDesc gsh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = tid; i < D; i += 1024)
        for (int c = 0; c < 3; ++c) {
            const float v = node_pos[3 * (size_t)i + c];
            mn[c] = fminf(mn[c], v), mx[c] = fmaxf(mx[c], v);
        }
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
        if (lane == 0) smin[c][wave] = mn[c], smax[c][wave] = mx[c];
    }
    __syncthreads();
    if (tid == 0) {  // geometry: the arithmetic of grid_setup_kernel
        float ext[3];
        for (int c = 0; c < 3; ++c) {
            float a = smin[c][0], b = smax[c][0];
            for (int w = 1; w < 16; ++w) a = fminf(a, smin[c][w]), b = fmaxf(b, smax[c][w]);
            gsh.bmin[c] = a;
            ext[c]      = fmaxf(b - a, 0.f);
        }
        const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
        float cs = cbrtf((ext[0] * ext[1] * ext[2]) / (float)D);
        cs       = fmaxf(cs, emax / (float)KNN_GRID_MAX_DIM);
        if (!(cs > 0.f)) cs = 1.f;
        gsh.cs     = cs;
        gsh.inv_cs = 1.f / cs;
        for (int c = 0; c < 3; ++c) {
            const int n = (int)(ext[c] * gsh.inv_cs) + 1;
            gsh.dim[c]  = min(max(n, 1), KNN_GRID_MAX_DIM);
        }
        *desc = gsh;
    }
    __syncthreads();
    const KnnGridDesc g = gsh;
    // only the cells of this grid are cleared, scanned and published (the searches never index beyond dim x dim x dim):
    // wave w owns PER consecutive cells, 64 per step
    const int ncells = g.dim[0] * g.dim[1] * g.dim[2];
    const int PER    = ((ncells + 16 * 64 - 1) / (16 * 64)) * 64;  // <= KNN_GRID_MAX_CELLS / 16
    for (int j = 0; j < PER; j += 64) cnt[wave * PER + j + lane] = 0;
    __syncthreads();
    for (int i = tid; i < D; i += 1024) {
        int cx, cy, cz;
        cell_of(g, mk3(node_pos[3 * (size_t)i], node_pos[3 * (size_t)i + 1], node_pos[3 * (size_t)i + 2]), cx, cy, cz);
        atomicAdd(&cnt[cx + g.dim[0] * (cy + g.dim[1] * cz)], 1);
    }
    __syncthreads();
    // exclusive scan in cell order
    int carry = 0;
    for (int j = 0; j < PER; j += 64) {
        const int idx  = wave * PER + j + lane;
        const int v    = cnt[idx];
        const int incl = wave_inclusive_scan(v);
        cnt[idx]       = carry + incl - v;
        carry += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) wave_tot[wave] = carry;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    for (int j = 0; j < PER; j += 64) {
        const int idx   = wave * PER + j + lane;
        const int start = cnt[idx] + off;
        cnt[idx]        = start;  // cursor of the fill
        if (idx <= ncells) cell_start[idx] = start;
    }
    if (tid == 1023 && 16 * PER <= ncells) cell_start[ncells] = off + carry;  // (16 PER == ncells: the end marker)
    __syncthreads();
    for (int i = tid; i < D; i += 1024) {
        const float px = node_pos[3 * (size_t)i], py = node_pos[3 * (size_t)i + 1], pz = node_pos[3 * (size_t)i + 2];
        int cx, cy, cz;
        cell_of(g, mk3(px, py, pz), cx, cy, cz);
        const int slot = atomicAdd(&cnt[cx + g.dim[0] * (cy + g.dim[1] * cz)], 1);
        sorted[slot]   = make_float4(px, py, pz, __int_as_float(i));
    }
}

// ---- large point sets (the canonical cloud of dfa_correspond): up to 128^3 cells, 256^3 above 500 k points ----
// Same data structure, built by multi-workgroup kernels: bounding box by per-block partials, the
// exclusive scan of the cell counts in chunks of PGRID_CHUNK cells (chunk sums, then a scan kernel
// that first adds up the sums of the chunks before it).
__global__ __launch_bounds__(256) void pgrid_bbox_kernel(const float* __restrict__ pts, int n,
                                                         float* __restrict__ partials /* gridDim.x x 6 */) {
    __shared__ float sh[6][4];
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int c = 0; c < 3; ++c) {
            const float v = pts[3 * (size_t)i + c];
            mn[c] = fminf(mn[c], v), mx[c] = fmaxf(mx[c], v);
        }
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
        if ((threadIdx.x & 63) == 0) sh[c][threadIdx.x >> 6] = mn[c], sh[3 + c][threadIdx.x >> 6] = mx[c];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        float a     = sh[c][0];
        for (int w = 1; w < 4; ++w) a = c < 3 ? fminf(a, sh[c][w]) : fmaxf(a, sh[c][w]);
        partials[6 * blockIdx.x + c] = a;
    }
}

__global__ __launch_bounds__(64) void pgrid_finalize_kernel(const float* __restrict__ partials, int nblocks, int n,
                                                            KnnGridDesc* __restrict__ desc) {
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int b = threadIdx.x; b < nblocks; b += 64)
        for (int c = 0; c < 3; ++c) mn[c] = fminf(mn[c], partials[6 * b + c]), mx[c] = fmaxf(mx[c], partials[6 * b + 3 + c]);
    for (int c = 0; c < 3; ++c)
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
    if (threadIdx.x == 0) {
        float ext[3];
        for (int c = 0; c < 3; ++c) desc->bmin[c] = mn[c], ext[c] = fmaxf(mx[c] - mn[c], 0.f);
        const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
        // points of a surface: the volume rule over-estimates the spacing, so aim below it and let the
        // cells-per-axis cap decide for thin clouds.  Up to half a million points: 128 cells per axis (2 M cells, 8 MiB of
        // counters to clear and scan per build).  Above (a 512^3 volume's surface: a million vertices) the search
        // walks ~70 points per cell at that cap, so the cap doubles and the aim halves: ~20 points per cell, a search
        // 2.5x shorter for ~40 us more of clearing and scanning.
        const bool large = n > 500000;
        float cs = (large ? 0.5f : 0.7f) * cbrtf((ext[0] * ext[1] * ext[2]) / (float)n);
        cs       = fmaxf(cs, emax / (float)(large ? PGRID_MAX_DIM : 128));
        if (!(cs > 0.f)) cs = 1.f;
        desc->cs = cs, desc->inv_cs = 1.f / cs;
        // clamped with the SAME cap the cell size was derived from (and the host's chunk count assumes, point_grid_build):
        // dim[0] dim[1] dim[2] <= cap^3 whatever the rounding of ext / cs does
        const int cap = large ? PGRID_MAX_DIM : 128;
        for (int c = 0; c < 3; ++c) desc->dim[c] = min(max((int)(ext[c] * desc->inv_cs) + 1, 1), cap);
    }
}

__device__ __forceinline__ int pgrid_cells(const KnnGridDesc& g) { return g.dim[0] * g.dim[1] * g.dim[2]; }

__global__ __launch_bounds__(256) void pgrid_clear_kernel(const KnnGridDesc* __restrict__ desc,
                                                          int32_t* __restrict__ cell_count) {
    const int nc = pgrid_cells(*desc);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x) cell_count[i] = 0;
}

__global__ __launch_bounds__(256) void pgrid_sum_kernel(const KnnGridDesc* __restrict__ desc,
                                                        const int32_t* __restrict__ cell_count,
                                                        int32_t* __restrict__ chunk_sums) {
    __shared__ int sh[4];
    const int nc = pgrid_cells(*desc), base = blockIdx.x * PGRID_CHUNK;
    int sum = 0;
    for (int i = base + threadIdx.x; i < min(base + PGRID_CHUNK, nc); i += 256) sum += cell_count[i];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// exclusive scan of the (at most PGRID_MAX_CELLS / PGRID_CHUNK = 2048) chunk totals, in place, by one workgroup
__global__ __launch_bounds__(256) void pgrid_chunkscan_kernel(int32_t* __restrict__ chunk_sums, int chunks) {
    __shared__ int wsum[4];
    constexpr int PER = (PGRID_MAX_CELLS / PGRID_CHUNK + 255) / 256;
    int loc[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = (int)threadIdx.x * PER + j;
        loc[j]      = i < chunks ? chunk_sums[i] : 0;
        sum += loc[j];
    }
    int incl       = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = incl - sum;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = (int)threadIdx.x * PER + j;
        if (i < chunks) chunk_sums[i] = off;
        off += loc[j];
    }
}

__global__ __launch_bounds__(256) void pgrid_scan_kernel(const KnnGridDesc* __restrict__ desc,
                                                         int32_t* __restrict__ cell_count /* in: counts, out: cursors */,
                                                         int32_t* __restrict__ cell_start,
                                                         const int32_t* __restrict__ chunk_sums) {
    __shared__ int sh[4], sh2[4];
    const int nc = pgrid_cells(*desc), base = blockIdx.x * PGRID_CHUNK;
    if (base >= nc) return;
    // offset of this chunk: chunk_sums holds the exclusive scan of the chunk totals (pgrid_chunkscan_kernel)
    if (threadIdx.x < 4) sh[threadIdx.x] = threadIdx.x == 0 ? chunk_sums[blockIdx.x] : 0;
    constexpr int PER = PGRID_CHUNK / 256;
    const int first   = base + threadIdx.x * PER;
    // a thread's PER = 32 cells are 128 contiguous bytes: sixteen-byte accesses (dword accesses at a 128-byte lane stride were
    // 32 instructions of 64 cache lines each way: 84 us for the 16.7 M cells of a 256^3 grid)
    static_assert(PER % 4 == 0, "cells per thread in groups of four");
    const bool whole = first + PER <= nc;  // (all of the thread's cells exist: every thread but the grid's last few)
    int loc[PER], sum = 0;
    if (whole) {
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) {
            const int4 c4 = reinterpret_cast<const int4*>(cell_count + first)[q];
            loc[4 * q] = c4.x, loc[4 * q + 1] = c4.y, loc[4 * q + 2] = c4.z, loc[4 * q + 3] = c4.w;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) sum += loc[j];
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) loc[j] = first + j < nc ? cell_count[first + j] : 0, sum += loc[j];
    }
    int incl       = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) sh2[wave] = incl;
    __syncthreads();
    int off = sh[0] + sh[1] + sh[2] + sh[3] + incl - sum;
    for (int w = 0; w < wave; ++w) off += sh2[w];
    if (whole) {
        int st[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) st[j] = off, off += loc[j];
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) {
            const int4 v = make_int4(st[4 * q], st[4 * q + 1], st[4 * q + 2], st[4 * q + 3]);
            reinterpret_cast<int4*>(cell_start + first)[q] = v;
            reinterpret_cast<int4*>(cell_count + first)[q] = v;
        }
        if (first + PER == nc) cell_start[nc] = off;
        return;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (first + j < nc) {
            cell_start[first + j] = off;
            cell_count[first + j] = off;
            off += loc[j];
            if (first + j == nc - 1) cell_start[nc] = off;
        }
    }
}

// exact k-NN through the grid: Chebyshev shells r = 0, 1, 2, ... around the query's cell
// TIGHT: the stop bound also counts the query's distance to the nearest wall of its own cell (the
// visited block of cells extends r cells beyond that wall), which lets a 1-NN search stop inside
// shell 0 / 1 of a fine grid.  Same result either way — the bound only decides when to stop.
template <int K, bool TIGHT = false>
__device__ __forceinline__ void knn_grid_query(const KnnGridDesc& g, const int32_t* __restrict__ cell_start,
                                               const float4* __restrict__ sorted, f3 q, KnnList<K>& best, bool rescan = false) {
    // TIGHT scans the query's own cell twice (alone, then inside the 3 x 3 x 3 block): harmless for K = 1, where a repeated
    // candidate cannot displace anything, but a K > 1 list would hold the same node twice
    static_assert(!TIGHT || K == 1, "the tight stop bound re-scans the own cell: 1-NN only");
    best.init();
    int cx, cy, cz;
    cell_of(g, q, cx, cy, cz);
    float margin = 0.f;  // in cells; 0 for queries outside the grid (clamped above)
    float wall[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};  // TIGHT: distance to the low / high wall of the own cell per axis (m)
    bool inside      = false;
    if (TIGHT) {
        const float ux = (q.x - g.bmin[0]) * g.inv_cs - (float)cx, uy = (q.y - g.bmin[1]) * g.inv_cs - (float)cy,
                    uz = (q.z - g.bmin[2]) * g.inv_cs - (float)cz;
        const float m = fminf(fminf(fminf(ux, 1.f - ux), fminf(uy, 1.f - uy)), fminf(uz, 1.f - uz));
        margin        = m > 0.f ? m : 0.f;  // negative (outside) or NaN -> 0
        inside        = m >= 0.f;           // (false for NaN)
        const float u[3] = {ux, uy, uz};
        // (1e-3 cells off every wall: the rounding of the cell assignment, as in the stop bounds below)
#pragma unroll
        for (int c = 0; c < 3; ++c) wall[c][0] = fmaxf(u[c] - 1e-3f, 0.f) * g.cs, wall[c][1] = fmaxf(1.f - u[c] - 1e-3f, 0.f) * g.cs;
    }
    const int rmax = max(g.dim[0], max(g.dim[1], g.dim[2]));
    int r_first    = 0;
    // candidates [beg, end) of the sorted node array, four at a time from clamped indices
    auto scan_range = [&](int beg, int end) __attribute__((always_inline)) {
        constexpr int KNN_BATCH = 4;  // (8: the same 48 us at C2, 16: 64; one by one: 54)
        for (int j = beg; j < end; j += KNN_BATCH) {
            float4 n[KNN_BATCH];
#pragma unroll
            for (int t = 0; t < KNN_BATCH; ++t) n[t] = sorted[min(j + t, end - 1)];
#pragma unroll
            for (int t = 0; t < KNN_BATCH; ++t)
                if (j + t < end) best.push(dist2(q, n[t].x, n[t].y, n[t].z), __float_as_int(n[t].w));
        }
    };
    // the 3 x 3 x 3 block around the query's cell as nine x-rows of cells (the cells of an x-row are consecutive in the sorted
    // node array): the rows' ranges requested together — unconditional loads from clamped cells, a row outside the grid is
    // empty —, then every row's candidates
    auto scan_block3 = [&]() __attribute__((always_inline)) {
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
        int rbeg[9], rend[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int z = cz - 1 + i / 3, y = cy - 1 + i % 3;
            const bool in = z >= 0 && z < g.dim[2] && y >= 0 && y < g.dim[1];
            const int c   = in ? g.dim[0] * (y + g.dim[1] * z) : 0;
            const int b = cell_start[c + x0], e = cell_start[c + x1 + 1];
            rbeg[i] = in ? b : 0, rend[i] = in ? e : 0;
        }
        // (the row of the query's own cell first, then the four rows that share a face with it, then the corners: the list
        // fills with near candidates early, and a late candidate that no lane of the wave accepts skips the insertion)
        constexpr int order[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
#pragma unroll
        for (int i = 0; i < 9; ++i) scan_range(rbeg[order[i]], rend[order[i]]);
    };
    if (TIGHT) {
        // shell 0 (the query's own cell) — a 1-NN search on a fine grid usually ends here —, then, if it does not, shells 0
        // and 1 together as the nine rows (the own cell's candidates a second time: a set, the order and repeats do not
        // matter), each followed by the stop rule of its shell; further shells in the loop below
        const int c0 = cx + g.dim[0] * (cy + g.dim[1] * cz);
        scan_range(cell_start[c0], cell_start[c0 + 1]);
        const float b0 = fmaxf(margin - 1e-3f, 0.f) * g.cs;
        if (best.dist(K - 1) < b0 * b0 * 0.9999f) return;
        // Shell 1, without the cells that cannot hold anything nearer than what the own cell gave (d0): a query that fails
        // the test above sits near ONE wall or edge of its cell, and of the 26 neighbours only the few across that wall are
        // within d0 — a cell whose nearest point is farther than d0 is skipped by the rule that ends the search (strictly
        // farther, with the same margins), so the result, ties included, is that of the full block.  The own cell is not
        // scanned again unless both x neighbours of its row are.  On a million-point surface (~20 points per cell at the
        // 256-cell cap) the full block is ~180 candidates; this is the part of it across the near walls.
        {
            const float d0 = best.dist(K - 1);
            int rbeg[9], rend[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int dz = i / 3 - 1, dy = i % 3 - 1, z = cz + dz, y = cy + dy;
                const float ey = dy < 0 ? wall[1][0] : dy > 0 ? wall[1][1] : 0.f, ez = dz < 0 ? wall[2][0] : dz > 0 ? wall[2][1] : 0.f;
                const float rb2 = ey * ey + ez * ez;
                const bool keep = !inside || !(d0 < rb2 * 0.9999f);
                const bool xl   = keep && (!inside || !(d0 < (rb2 + wall[0][0] * wall[0][0]) * 0.9999f));
                const bool xr   = keep && (!inside || !(d0 < (rb2 + wall[0][1] * wall[0][1]) * 0.9999f));
                const bool own  = dy == 0 && dz == 0;  // the row of the own cell: that cell is done
                const bool mid  = keep && (!own || (xl && xr));
                const bool in   = z >= 0 && z < g.dim[2] && y >= 0 && y < g.dim[1] && (xl || xr || mid);
                const int x0 = max(xl ? cx - 1 : (mid ? cx : cx + 1), 0), x1 = min(xr ? cx + 1 : (mid ? cx : cx - 1), g.dim[0] - 1);
                const int c  = in ? g.dim[0] * (y + g.dim[1] * z) : 0;
                const bool some = in && x0 <= x1;
                const int b = cell_start[c + (some ? x0 : 0)], e = cell_start[c + (some ? x1 + 1 : 0)];
                rbeg[i] = some ? b : 0, rend[i] = some ? e : 0;
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) scan_range(rbeg[i], rend[i]);  // (one key per lane: the order does not matter here)
        }
        const float b1 = fmaxf(1.f + margin - 1e-3f, 0.f) * g.cs;
        if (best.dist(K - 1) < b1 * b1 * 0.9999f) return;
        r_first = 2;
        // Not settled by shells 0 and 1 (the surface has moved by more than a cell): whatever they found is an upper bound d
        // of the answer, and everything at most that far away lies in the cells that meet the ball of radius sqrt(d) around
        // the query — per (dy, dz) row one contiguous x range of cells, the rows of a z layer requested together.  That is
        // the exact answer (ties included: the ball is closed, with the margins of the stop rules), for the price of the
        // ball's cells instead of whole Chebyshev shells walked cell by cell with two dependent table loads each (2.2 ms
        // against 0.13 ms per 262 k queries when the cloud had moved by a centimetre).  Balls wider than RB cells, queries
        // outside the grid and empty neighbourhoods take the shell loop below.
        constexpr int RB = 4;
        auto reach = [&](float r, float w) { return r > w ? (int)fminf((r - w) * g.inv_cs, 1e6f) + 1 : 0; };
        // (nothing within the block: the ball is tried at two cells and grown by one until it holds a point — then that point,
        // or one nearer inside the same ball, is the answer.  A ball that holds a known point settles at once, so only the
        // balls that were EMPTY grow — and a grown ball scans only the cells the larger radius adds: per row the two ends of
        // its x range beyond the range of the ball before, nothing of which held a point.  The first form scanned the whole
        // ball again on every growth; `rescan` — development builds — keeps that form for the comparison.)
        float rad   = best.dist(K - 1) < 3.0e38f ? sqrtf(best.dist(K - 1)) * 1.0001f : 2.f * g.cs;
        bool settled = false;
        for (int attempt = 0; attempt < RB; ++attempt) {
            const float rad_p = attempt > 0 && !rescan ? rad - g.cs : -1.f;  // the (empty) ball this lane scanned before
            const int nzl = reach(rad, wall[2][0]), nzh = reach(rad, wall[2][1]), nyl = reach(rad, wall[1][0]), nyh = reach(rad, wall[1][1]);
            const bool ball = inside && !settled && max(max(nzl, nzh), max(nyl, nyh)) <= RB &&
                              max(reach(rad, wall[0][0]), reach(rad, wall[0][1])) <= RB;
            if (__ballot(ball) == 0ull) break;
            int zl = 0, zh = 0;  // the wave's reach in z (ballots: lanes that left the search earlier take no part)
#pragma unroll
            for (int v = 1; v <= RB; ++v) {
                if (__ballot(ball && nzl >= v) != 0ull) zl = v;
                if (__ballot(ball && nzh >= v) != 0ull) zh = v;
            }
            for (int sz = 0; sz <= 2 * max(zl, zh); ++sz) {  // z layers nearest first: 0, +1, -1, +2, ...
                const int dz = (sz & 1) ? (sz + 1) / 2 : -(sz / 2);
                if (dz > zh || -dz > zl) continue;  // (wave-uniform)
                const float ez = dz == 0 ? 0.f : (dz < 0 ? wall[2][0] : wall[2][1]) + (float)(abs(dz) - 1) * g.cs;
                int rb[2 * RB + 1], re[2 * RB + 1], pb[2 * RB + 1], pe[2 * RB + 1];
#pragma unroll
                for (int i = 0; i <= 2 * RB; ++i) {
                    const int dy   = i - RB;
                    const float ey = dy == 0 ? 0.f : (dy < 0 ? wall[1][0] : wall[1][1]) + (float)(abs(dy) - 1) * g.cs;
                    const float rem = rad * rad - ey * ey - ez * ez;
                    const int z = cz + dz, y = cy + dy;
                    const bool row = ball && rem >= 0.f && dz <= nzh && -dz <= nzl && dy <= nyh && -dy <= nyl && z >= 0 &&
                                     z < g.dim[2] && y >= 0 && y < g.dim[1];
                    const float sx = sqrtf(fmaxf(rem, 0.f));
                    const int x0 = max(cx - reach(sx, wall[0][0]), 0), x1 = min(cx + reach(sx, wall[0][1]), g.dim[0] - 1);
                    const int c  = row ? g.dim[0] * (y + g.dim[1] * z) : 0;
                    const int b = cell_start[c + (row ? x0 : 0)], e = cell_start[c + (row ? x1 + 1 : 0)];
                    // the part of this row the ball before covered (same formulas at the radius before: a sub-range)
                    const float remp = rad_p * rad_p - ey * ey - ez * ez;
                    const bool prow  = row && rad_p > 0.f && remp >= 0.f;
                    const float sxp  = sqrtf(fmaxf(remp, 0.f));
                    const int x0p = max(cx - reach(sxp, wall[0][0]), x0), x1p = min(cx + reach(sxp, wall[0][1]), x1);
                    const int bp = cell_start[c + (prow ? x0p : 0)], ep = cell_start[c + (prow ? x1p + 1 : 0)];
                    rb[i] = row ? b : 0, re[i] = row ? e : 0;
                    pb[i] = prow ? bp : re[i], pe[i] = prow ? ep : re[i];  // (no ball before: [rb, re) and an empty second part)
                }
#pragma unroll
                for (int i = 0; i <= 2 * RB; ++i) scan_range(rb[i], pb[i]), scan_range(pe[i], re[i]);
            }
            // every point within `rad` of the query has been looked at: a best inside the ball is the nearest point
            if (ball && best.dist(K - 1) <= rad * rad * 0.9999f) settled = true;
            else rad += g.cs;  // (only balls that were empty so far get here: one more cell)
        }
        if (settled) return;
    }
    if (!TIGHT) {
        // Shells 0 and 1 together: the 3 x 3 x 3 block around the query's cell is nine x-rows of cells, and the cells of
        // an x-row are consecutive in the sorted node array — nine contiguous candidate ranges (18 cell_start loads)
        // instead of 27 cells (54) walked one by one.  Nearly every query ends here: the stop rule below is that of r = 1.
        // The query is a chain of dependent loads and little else (a wave of 64 queries is resident from launch to end:
        // 4 waves per SIMD at C2), so the loads are issued for memory-level parallelism (scan_block3 above).
        scan_block3();
        if (best.dist(K - 1) < g.cs * g.cs * 0.9999f) return;  // (r = 1: every node not visited is at least one cell away)
        r_first = 2;  // (a grid of at most 2 cells per axis has been visited completely: the loop below does not run)
    }
    for (int r = r_first; r < rmax; ++r) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.dim[2] - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, g.dim[1] - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.dim[0] - 1);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const bool face = (abs(z - cz) == r) || (abs(y - cy) == r);
                // on a z/y face of the shell every x belongs to it; otherwise only the two x ends
                const int xstep = face ? 1 : max(x1 - x0, 1);
                for (int x = x0; x <= x1; x += xstep) {
                    if (!face && abs(x - cx) != r) continue;
                    const int c   = x + g.dim[0] * (y + g.dim[1] * z);
                    const int beg = cell_start[c], end = cell_start[c + 1];
                    for (int j = beg; j < end; ++j) {
                        const float4 n = sorted[j];
                        best.push(dist2(q, n.x, n.y, n.z), __float_as_int(n.w));
                    }
                }
            }
        // every node not visited yet is at least r*cs away (from the query's projection onto
        // the grid, hence from the query); stop when the k-th candidate is strictly closer,
        // with a relative margin that absorbs the rounding of the cell assignment
        // (TIGHT: the cell coordinate of a point is rounded with an error ~1e-5 cells at 128 cells per
        // axis, twice that at 256; 1e-3 cells are taken off the bound before the relative margin)
        const float bound = TIGHT ? fmaxf((float)r + margin - 1e-3f, 0.f) * g.cs : (float)r * g.cs;
        if (best.dist(K - 1) < bound * bound * 0.9999f) break;
    }
}

// ---- wave-cooperative grid search: ONE WAVE PER QUERY -----------------------------------------
// For a few thousand queries (the node -> node regularisation graph) one lane per query leaves
// most of the chip idle and every lane walks ~100 cells serially.  Here the 64 lanes of a wave
// split the cells of each shell, keep private sorted lists, stop when at least K candidates lie
// strictly inside the shell bound, and merge their lists with K rounds of a 64-bit wave minimum
// on (distance bits << 32 | index) keys — the same (distance, index) order as the other paths.
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o, 64);
        v                          = t < v ? t : v;
    }
    return v;
}

template <int K>
__global__ __launch_bounds__(256) void knn_wave_kernel(const float* __restrict__ node_pos,
                                                       const float* __restrict__ node_w, int D,
                                                       const float* __restrict__ query, int n_query, int k,
                                                       int32_t* __restrict__ idx, float* __restrict__ weights,
                                                       KnnGridView grid) {
    const int v    = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (v >= n_query) return;
    const KnnGridDesc g = *grid.desc;
    const f3 q = mk3(query[3 * (size_t)v], query[3 * (size_t)v + 1], query[3 * (size_t)v + 2]);
    KnnList<K> best;
    best.init();
    int cx, cy, cz;
    cell_of(g, q, cx, cy, cz);
    const int rmax = max(g.dim[0], max(g.dim[1], g.dim[2]));
    for (int r = 0; r < rmax; ++r) {
        const int side = 2 * r + 1, ncell = side * side * side;
        for (int c = lane; c < ncell; c += 64) {
            const int dx = c % side - r, dy = (c / side) % side - r, dz = c / (side * side) - r;
            if (max(abs(dx), max(abs(dy), abs(dz))) != r) continue;  // interior: earlier shells
            const int x = cx + dx, y = cy + dy, z = cz + dz;
            if (x < 0 || y < 0 || z < 0 || x >= g.dim[0] || y >= g.dim[1] || z >= g.dim[2]) continue;
            const int cell = x + g.dim[0] * (y + g.dim[1] * z);
            const int beg = grid.cell_start[cell], end = grid.cell_start[cell + 1];
            for (int j = beg; j < end; ++j) {
                const float4 n = grid.sorted[j];
                best.push(dist2(q, n.x, n.y, n.z), __float_as_int(n.w));
            }
        }
        // unvisited nodes are >= r*cs away: done once K candidates are strictly closer (margin as in
        // knn_grid_query)
        const float bound = (float)r * g.cs, b2 = bound * bound * 0.9999f;
        int inside = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) inside += best.dist(j) < b2 ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) inside += __shfl_xor(inside, o, 64);
        if (inside >= K) break;
    }
    // merge: K rounds of wave-min over each lane's current head
    int pos = 0;
    for (int j = 0; j < K; ++j) {
        unsigned long long key = KnnList<K>::EMPTY;
#pragma unroll
        for (int t = 0; t < K; ++t)
            if (t == pos) key = best.key[t];
        const int hi = (int)(unsigned int)key;
        const unsigned long long win = wave_min_u64(key);
        if (key == win && hi != 0x7fffffff) ++pos;
        if (lane == 0 && j < k) {
            const int n            = (int)(win & 0xffffffffu);
            const int node         = n == 0x7fffffff ? -1 : n;
            idx[(size_t)v * k + j] = node;
            if (weights) {
                float w = 0.f;
                if (node >= 0)
                    w = transformation_weight(mk3(node_pos[3 * node], node_pos[3 * node + 1], node_pos[3 * node + 2]),
                                              node_w[node], q);
                weights[(size_t)v * k + j] = w;
            }
        }
    }
}

template <int K, bool GRID>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ node_pos,
                                                  const float* __restrict__ node_w, int D,
                                                  const float* __restrict__ query, int n_query, int k,
                                                  int32_t* __restrict__ idx, float* __restrict__ weights,
                                                  KnnGridView grid) {
    __shared__ float4 tile[GRID ? 1 : KNN_TILE];
    const int v       = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = v < n_query;
    f3 q              = mk3(0.f, 0.f, 0.f);
    if (active) q = mk3(query[3 * (size_t)v], query[3 * (size_t)v + 1], query[3 * (size_t)v + 2]);
    KnnList<K> best;
    if (GRID) {
        if (!active) return;
        knn_grid_query<K>(*grid.desc, grid.cell_start, grid.sorted, q, best);
    } else {
        knn_scan<K>(node_pos, D, q, best, tile);
        if (!active) return;
    }
    int out_i[K];
    float out_w[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        out_i[j] = -1, out_w[j] = 0.f;
        if (j < k) {
            const int n = best.index(j);
            out_i[j]    = n;
            if (weights && n >= 0)
                out_w[j] = transformation_weight(mk3(node_pos[3 * n], node_pos[3 * n + 1], node_pos[3 * n + 2]), node_w[n], q);
        }
    }
    // (uniform) a query's k ids / weights are 16 or 32 contiguous bytes: 16-byte stores where the caller's arrays allow them
    if (k == K && K % 4 == 0 && ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(weights)) & 15u) == 0) {
#pragma unroll
        for (int h = 0; h < K / 4; ++h) {
            reinterpret_cast<int4*>(idx + (size_t)v * K)[h] = make_int4(out_i[4 * h], out_i[4 * h + 1], out_i[4 * h + 2], out_i[4 * h + 3]);
            if (weights)
                reinterpret_cast<float4*>(weights + (size_t)v * K)[h] = make_float4(out_w[4 * h], out_w[4 * h + 1], out_w[4 * h + 2], out_w[4 * h + 3]);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < k) {
            idx[(size_t)v * k + j] = out_i[j];
            if (weights) weights[(size_t)v * k + j] = out_w[j];
        }
}

// Warpfield::calcDQB (warp_field.cpp:127-148) given the neighbour list
template <int K>
__device__ __forceinline__ DQ calc_dqb(const KnnList<K>& nb, int k, const float* __restrict__ node_pos,
                                       const float* __restrict__ node_dq, const float* __restrict__ node_w, f3 p) {
    DQ sum = dq_identity();  // :133
    // the neighbours' positions, radii and transforms four at a time by unconditional loads (an absent neighbour reads node 0
    // and is skipped): with the loads inside the `if` they were k dependent round trips.  Same products in the same order.
    constexpr int G = K < 4 ? K : 4;
#pragma unroll
    for (int h = 0; h < K; h += G) {
        f3 g[G];
        float r[G];
        DQ q[G];
        bool on[G];
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const int j = h + jj;
            on[jj]      = j < k && nb.index(j) >= 0;
            const int n = on[jj] ? nb.index(j) : 0;
            g[jj] = mk3(node_pos[3 * n], node_pos[3 * n + 1], node_pos[3 * n + 2]), r[jj] = node_w[n];
            q[jj] = dq_load(node_dq + 8 * (size_t)n);
        }
#pragma unroll
        for (int jj = 0; jj < G; ++jj)
            if (on[jj]) sum = dq_mul(sum, dq_scale(q[jj], transformation_weight(g[jj], r[jj], p)));  // :139-141
    }
    return dq_normalize(sum);  // :145
}

// Warpfield::warpToLive (warp_field.cpp:150-171)
template <int K, bool GRID>
__global__ __launch_bounds__(256) void warp_to_live_kernel(const float* __restrict__ node_pos,
                                                           const float* __restrict__ node_dq,
                                                           const float* __restrict__ node_w, int D, int k,
                                                           const float* __restrict__ verts,
                                                           const float* __restrict__ normals, int N,
                                                           float* __restrict__ out_verts,
                                                           float* __restrict__ out_normals, KnnGridView grid) {
    __shared__ float4 tile[GRID ? 1 : KNN_TILE];
    const int v       = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = v < N;
    f3 p              = mk3(0.f, 0.f, 0.f);
    if (active) p = mk3(verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2]);
    KnnList<K> best;
    if (GRID) {
        if (!active) return;
        knn_grid_query<K>(*grid.desc, grid.cell_start, grid.sorted, p, best);
    } else {
        knn_scan<K>(node_pos, D, p, best, tile);
        if (!active) return;
    }
    const DQ dq = calc_dqb<K>(best, k, node_pos, node_dq, node_w, p);
    const f3 o  = dq_transform(dq, p);
    out_verts[3 * (size_t)v] = o.x, out_verts[3 * (size_t)v + 1] = o.y, out_verts[3 * (size_t)v + 2] = o.z;
    if (normals && out_normals) {
        // transformNormal == transformVertex formula (dual_quaternion.hpp:217-228)
        const f3 nn = dq_transform(dq, mk3(normals[3 * (size_t)v], normals[3 * (size_t)v + 1],
                                           normals[3 * (size_t)v + 2]));
        out_normals[3 * (size_t)v] = nn.x, out_normals[3 * (size_t)v + 1] = nn.y,
                                out_normals[3 * (size_t)v + 2] = nn.z;
    }
}

// warpToLive with a neighbour list that is already known (the solver plan's data graph of the same vertices and
// nodes): the same calcDQB and transform as warp_to_live_kernel, without searching again
template <int K>
__global__ __launch_bounds__(256) void warp_graph_kernel(const float* __restrict__ node_pos,
                                                         const float* __restrict__ node_dq,
                                                         const float* __restrict__ node_w, int k,
                                                         const int32_t* __restrict__ idx, const float* __restrict__ verts,
                                                         const float* __restrict__ normals, int N,
                                                         float* __restrict__ out_verts, float* __restrict__ out_normals) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    KnnList<K> nb;
    nb.init();
    if (k == K && K % 4 == 0 && (reinterpret_cast<uintptr_t>(idx) & 15u) == 0) {  // (uniform) the vertex's k neighbours by 16-byte loads
#pragma unroll
        for (int h = 0; h < K / 4; ++h) {
            const int4 n4 = reinterpret_cast<const int4*>(idx + (size_t)v * K)[h];
            nb.set_index(4 * h, n4.x), nb.set_index(4 * h + 1, n4.y), nb.set_index(4 * h + 2, n4.z), nb.set_index(4 * h + 3, n4.w);
        }
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < k) {
                const int n = idx[(size_t)v * k + j];
                nb.set_index(j, n);
            }
    }
    const f3 p  = mk3(verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2]);
    const DQ dq = calc_dqb<K>(nb, k, node_pos, node_dq, node_w, p);
    const f3 o  = dq_transform(dq, p);
    out_verts[3 * (size_t)v] = o.x, out_verts[3 * (size_t)v + 1] = o.y, out_verts[3 * (size_t)v + 2] = o.z;
    if (normals && out_normals) {
        const f3 nn = dq_transform(dq, mk3(normals[3 * (size_t)v], normals[3 * (size_t)v + 1], normals[3 * (size_t)v + 2]));
        out_normals[3 * (size_t)v] = nn.x, out_normals[3 * (size_t)v + 1] = nn.y, out_normals[3 * (size_t)v + 2] = nn.z;
    }
}

// DynFusion::findCorrespondingFrame (dyn_fusion.cpp:212-242): for every LIVE vertex the nearest
// vertex of the (warped) canonical cloud, by the same exact (distance, index) order as the k-NN
// above with k = 1; the canonical vertex and normal at that index are gathered into a cloud that
// is index-aligned with the live one.  The reference rebuilds a nanoflann KD-tree over the N
// canonical points every frame on the host (:221-224) and queries it once per live vertex.
template <bool GRID>
__global__ __launch_bounds__(256) void correspond_kernel(const float* __restrict__ canon_v,
                                                         const float* __restrict__ canon_n, int n_canon,
                                                         const float* __restrict__ live_v, int n_live,
                                                         float* __restrict__ out_v, float* __restrict__ out_n,
                                                         int32_t* __restrict__ out_idx, KnnGridView grid, bool rescan) {
    __shared__ float4 tile[GRID ? 1 : KNN_TILE];
    const int v       = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = v < n_live;
    f3 q              = mk3(0.f, 0.f, 0.f);
    if (active) q = mk3(live_v[3 * (size_t)v], live_v[3 * (size_t)v + 1], live_v[3 * (size_t)v + 2]);
    KnnList<1> best;
    if (GRID) {
        if (!active) return;
        knn_grid_query<1, true>(*grid.desc, grid.cell_start, grid.sorted, q, best, rescan);
    } else {
        knn_scan<1>(canon_v, n_canon, q, best, tile);
        if (!active) return;
    }
    const int n = best.index(0);  // -1 only if every distance is NaN
    if (out_idx) out_idx[v] = n;
    const size_t src = 3 * (size_t)max(n, 0), dst = 3 * (size_t)v;
    if (out_v) out_v[dst] = canon_v[src], out_v[dst + 1] = canon_v[src + 1], out_v[dst + 2] = canon_v[src + 2];
    if (out_n && canon_n)
        out_n[dst] = canon_n[src], out_n[dst + 1] = canon_n[src + 1], out_n[dst + 2] = canon_n[src + 2];
}

// Warpfield::calcDQB (warp_field.cpp:127-148) at arbitrary points: the blended transform itself (new nodes
// are seeded with it, warp_field.cpp:78) — and Warpfield::getUnsupportedVertices (warp_field.cpp:34-62):
// a vertex is unsupported when min_j |v - g_j| / dg_w_j >= 1 over its k nearest nodes.
// FLAGS_ONLY: the instantiation of Warpfield::getUnsupportedVertices (no blended transform: 117 -> ~60 VGPRs at K = 8, twice
// the waves per SIMD for a kernel that is a chain of dependent loads)
template <int K, bool GRID, bool FLAGS_ONLY>
__global__ __launch_bounds__(256) void dqb_support_kernel(const float* __restrict__ node_pos,
                                                          const float* __restrict__ node_dq,
                                                          const float* __restrict__ node_w, int D, int k,
                                                          const float* __restrict__ pts, int n,
                                                          float* __restrict__ out_dq, uint8_t* __restrict__ out_flag,
                                                          KnnGridView grid) {
    __shared__ float4 tile[GRID ? 1 : KNN_TILE];
    const int v       = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = v < n;
    f3 p              = mk3(0.f, 0.f, 0.f);
    if (active) p = mk3(pts[3 * (size_t)v], pts[3 * (size_t)v + 1], pts[3 * (size_t)v + 2]);
    // the support quotient of one node (:45-46 — pow(float, int) is double arithmetic, the root is rounded to float on assignment)
    auto quotient = [&](int m) __attribute__((always_inline)) {
        const double dx = (double)(p.x - node_pos[3 * m]), dy = (double)(p.y - node_pos[3 * m + 1]),
                     dz = (double)(p.z - node_pos[3 * m + 2]);
        return (float)sqrt(dx * dx + dy * dy + dz * dz) / node_w[m];
    };
    KnnList<K> best;
    if (GRID) {
        if (!active) return;
        if (FLAGS_ONLY && k >= 1) {
            // Flags only (Warpfield::getUnsupportedVertices): the NEAREST node is one of the k nearest whatever k is, so a
            // vertex inside that node's radius is supported (min <= its quotient < 1) without the other k - 1 — the 1-NN
            // search stops in shell 0 / 1 and keeps no sorted list.  Almost every vertex of a tracked surface ends here; the
            // others (a farther node with a wider radius may still support them) take the full search below.
            KnnList<1> near;
            knn_grid_query<1, true>(*grid.desc, grid.cell_start, grid.sorted, p, near);
            const int m = near.index(0);
            if (m >= 0 && quotient(m) < 1.f) {
                out_flag[v] = 0;
                return;
            }
        }
        knn_grid_query<K>(*grid.desc, grid.cell_start, grid.sorted, p, best);
    } else {
        knn_scan<K>(node_pos, D, p, best, tile);
        if (!active) return;
    }
    if (!FLAGS_ONLY && out_dq) dq_store(out_dq + 8 * (size_t)v, calc_dqb<K>(best, k, node_pos, node_dq, node_w, p));
    if (out_flag) {
        float mn = __builtin_huge_valf();  // :40
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j < k && best.index(j) >= 0) {
                const float q = quotient(best.index(j));
                if (q <= mn) mn = q;  // :48-50
            }
        }
        out_flag[v] = mn >= 1.f ? 1 : 0;  // :53
    }
}

// ------------------------------------------------------------------------------ launchers

hipError_t point_grid_build(const PointGridView& pg, const float* pts, int n, hipStream_t s) {
    const KnnGridView& g = pg.g;
    const int nb         = (n + 255) / 256;
    const int bbox_blocks = nb < PGRID_BBOX_BLOCKS ? nb : PGRID_BBOX_BLOCKS;
    pgrid_bbox_kernel<<<bbox_blocks, 256, 0, s>>>(pts, n, pg.bbox_partials);
    pgrid_finalize_kernel<<<1, 64, 0, s>>>(pg.bbox_partials, bbox_blocks, n, g.desc);
    pgrid_clear_kernel<<<1024, 256, 0, s>>>(g.desc, g.cell_count);
    grid_count_kernel<<<nb, 256, 0, s>>>(pts, n, g.desc, g.cell_count, g.node_cell);
    const int chunks = (n > 500000 ? PGRID_MAX_CELLS : 128 * 128 * 128) / PGRID_CHUNK;  // (the cap pgrid_finalize_kernel applies)
    pgrid_sum_kernel<<<chunks, 256, 0, s>>>(g.desc, g.cell_count, pg.chunk_sums);
    pgrid_chunkscan_kernel<<<1, 256, 0, s>>>(pg.chunk_sums, chunks);
    pgrid_scan_kernel<<<chunks, 256, 0, s>>>(g.desc, g.cell_count, g.cell_start, pg.chunk_sums);
    grid_fill_kernel<<<nb, 256, 0, s>>>(pts, n, g.node_cell, g.cell_count, g.sorted);
    return hipGetLastError();
}

hipError_t knn_grid_build(const KnnGridView& g, const float* node_pos, int D, hipStream_t s) {
    const bool four = dev_env("DFA_GRID_FOUR_KERNELS") != nullptr;  // A/B (development builds): the four-kernel build below 2 048 nodes
    if (D <= GRID_ONE_MAX && !four) {
        constexpr size_t lds = sizeof(int32_t) * KNN_GRID_MAX_CELLS;
        // 128 KiB of dynamic LDS needs the opt-in, once per device
        hipError_t e = allow_dynamic_lds((const void*)grid_build_one_kernel, (int)lds);
        if (e != hipSuccess) return e;
        grid_build_one_kernel<<<1, 1024, lds, s>>>(node_pos, D, g.desc, g.cell_start, g.sorted);
        return hipGetLastError();
    }
    grid_setup_kernel<<<1, 1024, 0, s>>>(node_pos, D, g.desc, g.cell_count);
    grid_count_kernel<<<(D + 255) / 256, 256, 0, s>>>(node_pos, D, g.desc, g.cell_count, g.node_cell);
    grid_scan_kernel<<<1, 1024, 0, s>>>(g.cell_count, g.cell_start);
    grid_fill_kernel<<<(D + 255) / 256, 256, 0, s>>>(node_pos, D, g.node_cell, g.cell_count, g.sorted);
    return hipGetLastError();
}

#define KGDISPATCH(kernel, k, use_grid, ...)                         \
    do {                                                             \
        if (use_grid) {                                              \
            if ((k) <= 4) kernel<4, true> __VA_ARGS__;               \
            else if ((k) <= 8) kernel<8, true> __VA_ARGS__;          \
            else kernel<16, true> __VA_ARGS__;                       \
        } else {                                                     \
            if ((k) <= 4) kernel<4, false> __VA_ARGS__;              \
            else if ((k) <= 8) kernel<8, false> __VA_ARGS__;         \
            else kernel<16, false> __VA_ARGS__;                      \
        }                                                            \
    } while (0)

hipError_t launch_knn(const float* node_pos, const float* node_w, int D, const float* query, int n_query, int k,
                      int32_t* idx, float* weights, const KnnGridView* grid, hipStream_t s) {
    if (n_query == 0) return hipSuccess;
    dim3 block(256), gridDim((n_query + 255) / 256);
    const bool use_grid = grid != nullptr;
    KnnGridView g       = use_grid ? *grid : KnnGridView{};
    if (use_grid && n_query <= 32768) {  // few queries: one wave per query
        dim3 wgrid((n_query + 3) / 4);
        if (k <= 4) knn_wave_kernel<4><<<wgrid, block, 0, s>>>(node_pos, node_w, D, query, n_query, k, idx, weights, g);
        else if (k <= 8) knn_wave_kernel<8><<<wgrid, block, 0, s>>>(node_pos, node_w, D, query, n_query, k, idx, weights, g);
        else knn_wave_kernel<16><<<wgrid, block, 0, s>>>(node_pos, node_w, D, query, n_query, k, idx, weights, g);
        return hipGetLastError();
    }
    KGDISPATCH(knn_kernel, k, use_grid, <<<gridDim, block, 0, s>>>(node_pos, node_w, D, query, n_query, k, idx, weights, g));
    return hipGetLastError();
}

hipError_t launch_warp_to_live(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                               const float* verts, const float* normals, int N, float* out_verts, float* out_normals,
                               const KnnGridView* grid, hipStream_t s) {
    if (N == 0) return hipSuccess;
    dim3 block(256), gridDim((N + 255) / 256);
    const bool use_grid = grid != nullptr;
    KnnGridView g       = use_grid ? *grid : KnnGridView{};
    KGDISPATCH(warp_to_live_kernel, k, use_grid,
               <<<gridDim, block, 0, s>>>(node_pos, node_dq, node_w, D, k, verts, normals, N, out_verts, out_normals, g));
    return hipGetLastError();
}

// Projective association: the O(N) alternative to the nearest-neighbour search above.  Every
// (warped canonical) vertex is projected into the live frame's vertex map and takes the vertex / normal of the pixel it
// lands on, under the gates of ComputeIcpHelper::find_coresp (proj_icp.cu:72-98: point-sampled fetch, squared-distance
// threshold, |n . n'| >= min_cosine).  NaN / -1 where there is no association.
__global__ __launch_bounds__(256) void correspond_projective_kernel(const float* __restrict__ verts,
                                                                    const float* __restrict__ normals, int n,
                                                                    const float* __restrict__ vmap, int vmap_step,
                                                                    const float* __restrict__ nmap, int nmap_step, int cols,
                                                                    int rows, float fx, float fy, float cx, float cy,
                                                                    float dist2_thres, float min_cosine,
                                                                    float* __restrict__ out_v, float* __restrict__ out_n,
                                                                    int32_t* __restrict__ out_pixel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float qn = __uint_as_float(0x7fc00000u);
    const f3 s     = mk3(verts[3 * (size_t)i], verts[3 * (size_t)i + 1], verts[3 * (size_t)i + 2]);
    f3 d = mk3(qn, qn, qn), nd = mk3(qn, qn, qn);
    int pix = -1;
    if (s.z > 0.f) {
        const float u = fmaf(fx, s.x / s.z, cx), w = fmaf(fy, s.y / s.z, cy);  // proj, proj_icp.cu:28-33
        if (u >= 0.f && w >= 0.f && u < (float)cols && w < (float)rows) {
            const int iu = (int)floorf(u), iw = (int)floorf(w);  // point-sampled
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(vmap) + (size_t)iw * vmap_step + 16 * (size_t)iu);
            bool ok = v.x == v.x;
            const f3 dd = mk3(v.x, v.y, v.z), sd = s - dd;
            ok = ok && !(dot(sd, sd) > dist2_thres);
            f3 nn = mk3(qn, qn, qn);
            if (ok && nmap) {
                const float4 nv = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nmap) + (size_t)iw * nmap_step + 16 * (size_t)iu);
                nn = mk3(nv.x, nv.y, nv.z);
                ok = nv.x == nv.x;
                if (ok && normals) {
                    const f3 ns = mk3(normals[3 * (size_t)i], normals[3 * (size_t)i + 1], normals[3 * (size_t)i + 2]);
                    ok          = !(fabsf(dot(ns, nn)) < min_cosine);
                }
            }
            if (ok) d = dd, nd = nn, pix = iw * cols + iu;
        }
    }
    if (out_v) out_v[3 * (size_t)i] = d.x, out_v[3 * (size_t)i + 1] = d.y, out_v[3 * (size_t)i + 2] = d.z;
    if (out_n) out_n[3 * (size_t)i] = nd.x, out_n[3 * (size_t)i + 1] = nd.y, out_n[3 * (size_t)i + 2] = nd.z;
    if (out_pixel) out_pixel[i] = pix;
}

hipError_t launch_correspond_projective(const float* verts, const float* normals, int n, const float* vmap, int vmap_step,
                                        const float* nmap, int nmap_step, int cols, int rows, float fx, float fy, float cx,
                                        float cy, float dist_thres, float min_cosine, float* out_v, float* out_n,
                                        int32_t* out_pixel, hipStream_t s) {
    if (n == 0) return hipSuccess;
    correspond_projective_kernel<<<(n + 255) / 256, 256, 0, s>>>(verts, normals, n, vmap, vmap_step, nmap, nmap_step, cols,
                                                                 rows, fx, fy, cx, cy, dist_thres * dist_thres, min_cosine,
                                                                 out_v, out_n, out_pixel);
    return hipGetLastError();
}

hipError_t launch_correspond(const float* canon_v, const float* canon_n, int n_canon, const float* live_v,
                             int n_live, float* out_v, float* out_n, int32_t* out_idx, const KnnGridView* grid,
                             hipStream_t s) {
    if (n_live == 0) return hipSuccess;
    dim3 block(256), gridDim((n_live + 255) / 256);
    const bool rescan = dev_env("DFA_BALL_RESCAN") != nullptr;  // (development builds: a grown ball scans all of its cells again)
    if (grid) correspond_kernel<true><<<gridDim, block, 0, s>>>(canon_v, canon_n, n_canon, live_v, n_live, out_v, out_n, out_idx, *grid, rescan);
    else correspond_kernel<false><<<gridDim, block, 0, s>>>(canon_v, canon_n, n_canon, live_v, n_live, out_v, out_n, out_idx, KnnGridView{}, false);
    return hipGetLastError();
}

hipError_t launch_dqb_support(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                              const float* pts, int n, float* out_dq, uint8_t* out_flag, const KnnGridView* grid,
                              hipStream_t s) {
    if (n == 0) return hipSuccess;
    dim3 block(256), gridDim((n + 255) / 256);
    const bool use_grid = grid != nullptr;
    KnnGridView g       = use_grid ? *grid : KnnGridView{};
#define DQBK(KK, GG)                                                                                                         \
    do {                                                                                                                    \
        if (out_flag && !out_dq) dqb_support_kernel<KK, GG, true><<<gridDim, block, 0, s>>>(node_pos, node_dq, node_w, D, k, pts, n, out_dq, out_flag, g);  \
        else dqb_support_kernel<KK, GG, false><<<gridDim, block, 0, s>>>(node_pos, node_dq, node_w, D, k, pts, n, out_dq, out_flag, g);                     \
    } while (0)
    if (use_grid) {
        if (k <= 4) DQBK(4, true);
        else if (k <= 8) DQBK(8, true);
        else DQBK(16, true);
    } else {
        if (k <= 4) DQBK(4, false);
        else if (k <= 8) DQBK(8, false);
        else DQBK(16, false);
    }
#undef DQBK
    return hipGetLastError();
}

hipError_t launch_warp_graph(const float* node_pos, const float* node_dq, const float* node_w, int k, const int32_t* idx,
                             const float* verts, const float* normals, int N, float* out_verts, float* out_normals,
                             hipStream_t s) {
    if (N == 0) return hipSuccess;
    dim3 block(256), gridDim((N + 255) / 256);
    if (k <= 4) warp_graph_kernel<4><<<gridDim, block, 0, s>>>(node_pos, node_dq, node_w, k, idx, verts, normals, N, out_verts, out_normals);
    else if (k <= 8) warp_graph_kernel<8><<<gridDim, block, 0, s>>>(node_pos, node_dq, node_w, k, idx, verts, normals, N, out_verts, out_normals);
    else warp_graph_kernel<16><<<gridDim, block, 0, s>>>(node_pos, node_dq, node_w, k, idx, verts, normals, N, out_verts, out_normals);
    return hipGetLastError();
}

}  // namespace dfa
