// warp.hip — gfx950 kernels for the warp-field seam: exact k-NN of the deformation nodes,
// RBF transformation weights, the reference's ordered dual-quaternion "blend" and warpToLive.
//
// Reference semantics: src/dynfu/warp_field.cpp:99-171 (nanoflann k-NN per vertex on the
// CPU, three std::vector allocations per query) and src/dynfu/utils/node.cpp:29-36.
// MI355X layout: one lane per query vertex, the node positions are staged through LDS in
// 1024-node tiles (16 KiB, every lane of a wave reads the same node -> LDS broadcast, no bank
// conflicts), the k best candidates live in registers as a sorted list updated by a fully
// unrolled compare-exchange chain (no dynamic register indexing -> no scratch).  With at
// most a few thousand nodes the exhaustive scan (N*D distance evaluations, 0.5 G at
// 262 k vertices x 2 k nodes) is cheaper than any tree walk and is exact by construction.
#include <hip/hip_runtime.h>

#include "dq_device.hpp"
#include "kernels.hpp"

namespace dfa {

constexpr int KNN_TILE = 1024;

// sorted (ascending) list of the K nearest candidates; equal distances keep scan order
// (== nanoflann KNNResultSet::addPoint without NANOFLANN_FIRST_MATCH, nanoflann.hpp:100-122,
// when candidates arrive in ascending node index)
template <int K>
struct KnnList {
    float d[K];
    int i[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) d[j] = __builtin_inff(), i[j] = -1;
    }
    __device__ __forceinline__ void push(float dist, int idx) {
        if (dist < d[K - 1]) {
            d[K - 1] = dist, i[K - 1] = idx;
#pragma unroll
            for (int j = K - 1; j > 0; --j) {
                if (d[j] < d[j - 1]) {
                    const float td = d[j];
                    d[j] = d[j - 1], d[j - 1] = td;
                    const int ti = i[j];
                    i[j] = i[j - 1], i[j - 1] = ti;
                }
            }
        }
    }
};

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
            // L2_Simple_Adaptor::evalMetric (nanoflann.hpp:338-345): ((0 + d0^2) + d1^2) + d2^2
            const float d0 = q.x - g.x, d1 = q.y - g.y, d2 = q.z - g.z;
            const float dist = (d0 * d0 + d1 * d1) + d2 * d2;
            best.push(dist, base + j);
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ node_pos,
                                                  const float* __restrict__ node_w, int D,
                                                  const float* __restrict__ query, int n_query, int k,
                                                  int32_t* __restrict__ idx, float* __restrict__ weights) {
    __shared__ float4 tile[KNN_TILE];
    const int v       = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = v < n_query;
    f3 q              = mk3(0.f, 0.f, 0.f);
    if (active) q = mk3(query[3 * (size_t)v], query[3 * (size_t)v + 1], query[3 * (size_t)v + 2]);
    KnnList<K> best;
    knn_scan<K>(node_pos, D, q, best, tile);
    if (!active) return;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < k) {
            const int n                = best.i[j];
            idx[(size_t)v * k + j] = n;
            if (weights) {
                float w = 0.f;
                if (n >= 0)
                    w = transformation_weight(mk3(node_pos[3 * n], node_pos[3 * n + 1], node_pos[3 * n + 2]),
                                              node_w[n], q);
                weights[(size_t)v * k + j] = w;
            }
        }
    }
}

// Warpfield::calcDQB (warp_field.cpp:127-148) given the neighbour list
template <int K>
__device__ __forceinline__ DQ calc_dqb(const KnnList<K>& nb, int k, const float* __restrict__ node_pos,
                                       const float* __restrict__ node_dq, const float* __restrict__ node_w, f3 p) {
    DQ sum = dq_identity();  // :133
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < k && nb.i[j] >= 0) {
            const int n   = nb.i[j];
            const float w = transformation_weight(mk3(node_pos[3 * n], node_pos[3 * n + 1], node_pos[3 * n + 2]),
                                                  node_w[n], p);
            sum = dq_mul(sum, dq_scale(dq_load(node_dq + 8 * (size_t)n), w));  // :139-141
        }
    }
    return dq_normalize(sum);  // :145
}

// Warpfield::warpToLive (warp_field.cpp:150-171)
template <int K>
__global__ __launch_bounds__(256) void warp_to_live_kernel(const float* __restrict__ node_pos,
                                                           const float* __restrict__ node_dq,
                                                           const float* __restrict__ node_w, int D, int k,
                                                           const float* __restrict__ verts,
                                                           const float* __restrict__ normals, int N,
                                                           float* __restrict__ out_verts,
                                                           float* __restrict__ out_normals) {
    __shared__ float4 tile[KNN_TILE];
    const int v       = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = v < N;
    f3 p              = mk3(0.f, 0.f, 0.f);
    if (active) p = mk3(verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2]);
    KnnList<K> best;
    knn_scan<K>(node_pos, D, p, best, tile);
    if (!active) return;
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

hipError_t launch_knn(const float* node_pos, const float* node_w, int D, const float* query, int n_query, int k,
                      int32_t* idx, float* weights, hipStream_t s) {
    if (n_query == 0) return hipSuccess;
    dim3 block(256), grid((n_query + 255) / 256);
    if (k <= 4) knn_kernel<4><<<grid, block, 0, s>>>(node_pos, node_w, D, query, n_query, k, idx, weights);
    else if (k <= 8) knn_kernel<8><<<grid, block, 0, s>>>(node_pos, node_w, D, query, n_query, k, idx, weights);
    else knn_kernel<16><<<grid, block, 0, s>>>(node_pos, node_w, D, query, n_query, k, idx, weights);
    return hipGetLastError();
}

hipError_t launch_warp_to_live(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                               const float* verts, const float* normals, int N, float* out_verts, float* out_normals,
                               hipStream_t s) {
    if (N == 0) return hipSuccess;
    dim3 block(256), grid((N + 255) / 256);
    if (k <= 4)
        warp_to_live_kernel<4><<<grid, block, 0, s>>>(node_pos, node_dq, node_w, D, k, verts, normals, N, out_verts,
                                                       out_normals);
    else if (k <= 8)
        warp_to_live_kernel<8><<<grid, block, 0, s>>>(node_pos, node_dq, node_w, D, k, verts, normals, N, out_verts,
                                                       out_normals);
    else
        warp_to_live_kernel<16><<<grid, block, 0, s>>>(node_pos, node_dq, node_w, D, k, verts, normals, N, out_verts,
                                                        out_normals);
    return hipGetLastError();
}

}  // namespace dfa
