// points.hip — point-cloud plumbing between the seams: strided re-packing of xyz triples, order-preserving stream
// compaction and the affine map of a cloud.  The reference does these with host loops over pcl::PointCloud
// (pcl::copyPointCloud, dyn_fusion.cpp:80-88; the push_back loop of Warpfield::getUnsupportedVertices,
// warp_field.cpp:42-59); with the clouds resident in HBM they are small HBM-bound kernels.  Bit copies, except the affine
// map (three multiply-adds per coordinate in a fixed order).
#include "kernels.hpp"

namespace dfa {

// one lane per point; float4-strided sides are moved as one 16-byte access
__global__ void __launch_bounds__(256) repack_points_kernel(const float* __restrict__ src, int sstride,
                                                            float* __restrict__ dst, int dstride, int n, float pad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x, y, z;
    if (sstride == 4) {
        const float4 p = reinterpret_cast<const float4*>(src)[i];
        x = p.x, y = p.y, z = p.z;
    } else {
        const float* p = src + (size_t)i * sstride;
        x = p[0], y = p[1], z = p[2];
    }
    if (dstride == 4) {
        reinterpret_cast<float4*>(dst)[i] = make_float4(x, y, z, pad);
    } else {
        float* q = dst + (size_t)i * dstride;
        q[0] = x, q[1] = y, q[2] = z;
        for (int c = 3; c < dstride; ++c) q[c] = pad;
    }
}

hipError_t launch_repack_points(const float* src, int sstride, float* dst, int dstride, int n, float pad, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    repack_points_kernel<<<(n + 255) / 256, 256, 0, s>>>(src, sstride, dst, dstride, n, pad);
    return hipGetLastError();
}

// out = R p (+ t): one lane per point, packed n x 3 in and out (in == out allowed).  Plain multiply-adds in a fixed order
// ((R0 x + R1 y) + R2 z) + t, no contraction: the same bits as the host loop it replaces.
__global__ void __launch_bounds__(256) transform_points_kernel(const float* __restrict__ in, int n, float r0, float r1, float r2,
                                                               float r3, float r4, float r5, float r6, float r7, float r8,
                                                               float tx, float ty, float tz, int with_t, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = in[3 * (size_t)i], y = in[3 * (size_t)i + 1], z = in[3 * (size_t)i + 2];
    float ox = (r0 * x + r1 * y) + r2 * z, oy = (r3 * x + r4 * y) + r5 * z, oz = (r6 * x + r7 * y) + r8 * z;
    if (with_t) ox += tx, oy += ty, oz += tz;
    out[3 * (size_t)i] = ox, out[3 * (size_t)i + 1] = oy, out[3 * (size_t)i + 2] = oz;
}

hipError_t launch_transform_points(const float* in, int n, const float aff[12], bool with_translation, float* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    transform_points_kernel<<<(n + 255) / 256, 256, 0, s>>>(in, n, aff[0], aff[1], aff[2], aff[3], aff[4], aff[5], aff[6], aff[7],
                                                         aff[8], aff[9], aff[10], aff[11], with_translation ? 1 : 0, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------- compaction in index order
// chunk = COMPACT_CHUNK consecutive points per workgroup: count -> exclusive scan of the chunk totals (one workgroup)
// -> every chunk ranks its own flags again and writes its survivors behind its offset.
constexpr int COMPACT_THREADS = 256;
constexpr int COMPACT_ITEMS   = 16;  // consecutive flags per thread: one 16-byte load
constexpr int COMPACT_CHUNK   = COMPACT_THREADS * COMPACT_ITEMS;

int compact_chunks(int n) { return (n + COMPACT_CHUNK - 1) / COMPACT_CHUNK; }

__device__ __forceinline__ unsigned load_flag_bits(const uint8_t* __restrict__ flags, int base, int n) {
    unsigned bits = 0;
    if (base + COMPACT_ITEMS <= n && ((reinterpret_cast<uintptr_t>(flags + base) & 15) == 0)) {
        const uint4 w = *reinterpret_cast<const uint4*>(flags + base);
        const unsigned v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if ((v[j] >> (8 * b)) & 0xffu) bits |= 1u << (4 * j + b);
    } else {
        for (int j = 0; j < COMPACT_ITEMS; ++j)
            if (base + j < n && flags[base + j]) bits |= 1u << j;
    }
    return bits;
}

// exclusive prefix of `v` over the workgroup's threads (256 = 4 waves), total in *total
__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_sums, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < COMPACT_THREADS / 64; ++w) {
        const int sw = wave_sums[w];
        if (w < wave) before += sw;
        all += sw;
    }
    *total = all;
    return before + incl - v;
}

__global__ void __launch_bounds__(COMPACT_THREADS) compact_count_kernel(const uint8_t* __restrict__ flags, int n,
                                                                        int32_t* __restrict__ chunk_count) {
    __shared__ int wave_sums[COMPACT_THREADS / 64];
    const int base = (blockIdx.x * COMPACT_THREADS + threadIdx.x) * COMPACT_ITEMS;
    const int c    = base < n ? __popc(load_flag_bits(flags, base, n)) : 0;
    int total;
    (void)block_exclusive_scan(c, wave_sums, &total);
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = total;
}

// one workgroup: chunk_count -> exclusive offsets in place, grand total to *count
__global__ void __launch_bounds__(COMPACT_THREADS) compact_scan_kernel(int32_t* __restrict__ chunk_count, int chunks,
                                                                       int32_t* __restrict__ count) {
    __shared__ int wave_sums[COMPACT_THREADS / 64];
    int carry = 0;
    for (int b = 0; b < chunks; b += COMPACT_THREADS) {
        const int i = b + threadIdx.x;
        const int v = i < chunks ? chunk_count[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, wave_sums, &total);
        if (i < chunks) chunk_count[i] = carry + ex;
        carry += total;
        __syncthreads();  // wave_sums is re-used by the next round
    }
    if (threadIdx.x == 0) *count = carry;
}

__global__ void __launch_bounds__(COMPACT_THREADS) compact_write_kernel(const float* __restrict__ pts,
                                                                        const uint8_t* __restrict__ flags, int n,
                                                                        const int32_t* __restrict__ chunk_off,
                                                                        float* __restrict__ out_pts,
                                                                        int32_t* __restrict__ out_idx) {
    __shared__ int wave_sums[COMPACT_THREADS / 64];
    const int base = (blockIdx.x * COMPACT_THREADS + threadIdx.x) * COMPACT_ITEMS;
    unsigned bits  = base < n ? load_flag_bits(flags, base, n) : 0u;
    int total;
    int pos = chunk_off[blockIdx.x] + block_exclusive_scan(__popc(bits), wave_sums, &total);
    while (bits) {
        const int j = __ffs(bits) - 1;
        bits &= bits - 1;
        const int i = base + j;
        if (out_pts) {
            out_pts[3 * (size_t)pos]     = pts[3 * (size_t)i];
            out_pts[3 * (size_t)pos + 1] = pts[3 * (size_t)i + 1];
            out_pts[3 * (size_t)pos + 2] = pts[3 * (size_t)i + 2];
        }
        if (out_idx) out_idx[pos] = i;
        ++pos;
    }
}

hipError_t launch_compact_points(const float* pts, const uint8_t* flags, int n, float* out_pts, int32_t* out_idx,
                                 int32_t* count, int32_t* chunk_scratch, hipStream_t s) {
    if (n <= 0) return hipMemsetAsync(count, 0, sizeof(int32_t), s);
    const int chunks = compact_chunks(n);
    compact_count_kernel<<<chunks, COMPACT_THREADS, 0, s>>>(flags, n, chunk_scratch);
    compact_scan_kernel<<<1, COMPACT_THREADS, 0, s>>>(chunk_scratch, chunks, count);
    if (out_pts || out_idx) compact_write_kernel<<<chunks, COMPACT_THREADS, 0, s>>>(pts, flags, n, chunk_scratch, out_pts, out_idx);
    return hipGetLastError();
}

}  // namespace dfa
