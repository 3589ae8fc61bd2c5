// icp.hip — one linearisation of the reference's rigid projective ICP for gfx950:
// ComputeIcpHelper::find_coresp + the row [s x n, n | n.(d - s)] + its 27 products summed over the image
// (src/kfusion/cuda/proj_icp.cu:41-103 depth / points variants, :326-375; textures point-sampled :377-379).
// The 6x6 solve and the pose update stay on the host as in the reference (src/kfusion/projective_icp.cpp:118-150).
//
// One lane per pixel of the current frame; the 27 products are reduced per wave with DPP adds and per
// workgroup through LDS (the reference tree-reduces 27 times through shared memory with 5 barriers each), one
// partial per workgroup and product; a second launch adds the partials in index order in double precision.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "kernels.hpp"

namespace dfa {

namespace {

struct IcpArgs {
    const void* curr;   // u16 depth or float4 vertex map of the current frame
    const float* ncurr;
    const void* prev;
    const float* nprev;
    int curr_step, ncurr_step, prev_step, nprev_step, cols, rows;
    Aff3 aff;
    float fx, fy, cx, cy, finvx, finvy, min_cosine, dist2_thres;
};

template <class T>
__device__ __forceinline__ const T& at(const void* base, int step, int y, int x) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)y * step + sizeof(T) * (size_t)x);
}

template <bool DEPTH>
__device__ __forceinline__ bool find_coresp(const IcpArgs& a, int x, int y, f3& n, f3& d, f3& s) {
    f3 p;
    if (DEPTH) {
        const int src_z = at<uint16_t>(a.curr, a.curr_step, y, x);
        if (src_z == 0) return false;  // :44-46
        const float z = src_z * 0.001f;
        p = mk3(z * ((float)x - a.cx) * a.finvx, z * ((float)y - a.cy) * a.finvy, z);  // reproj :35-39
    } else {
        const float4 v = at<float4>(a.curr, a.curr_step, y, x);
        if (v.x != v.x) return false;  // :75-77
        p = mk3(v.x, v.y, v.z);
    }
    s = mulR(a.aff, p) + mk3(a.aff.t[0], a.aff.t[1], a.aff.t[2]);
    const float u = fmaf(a.fx, s.x / s.z, a.cx), w = fmaf(a.fy, s.y / s.z, a.cy);  // proj :28-33
    if (s.z <= 0.f || u < 0.f || w < 0.f || u >= (float)a.cols || w >= (float)a.rows) return false;
    const int iu = (int)floorf(u), iw = (int)floorf(w);  // point-sampled texture fetch
    if (DEPTH) {
        const int dst_z = at<uint16_t>(a.prev, a.prev_step, iw, iu);
        if (dst_z == 0) return false;
        const float z = dst_z * 0.001f;
        d = mk3(z * (u - a.cx) * a.finvx, z * (w - a.cy) * a.finvy, z);  // :57: reproj at the float coordinates
    } else {
        const float4 v = at<float4>(a.prev, a.prev_step, iw, iu);
        if (v.x != v.x) return false;
        d = mk3(v.x, v.y, v.z);
    }
    const f3 sd = s - d;
    if (dot(sd, sd) > a.dist2_thres) return false;  // :59-61
    const float4 nc = at<float4>(a.ncurr, a.ncurr_step, y, x);
    const f3 ns     = mulR(a.aff, mk3(nc.x, nc.y, nc.z));
    const float4 np = at<float4>(a.nprev, a.nprev_step, iw, iu);
    n               = mk3(np.x, np.y, np.z);
    return !(fabsf(dot(ns, n)) < a.min_cosine);  // :66-68
}

template <bool DEPTH>
__global__ __launch_bounds__(256) void icp_rows_kernel(const IcpArgs a, float* __restrict__ partial, int nblocks,
                                                       unsigned int* __restrict__ matched) {
    __shared__ float sh[4][28];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    float row[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f3 n, d, s;
    const bool ok = x < a.cols && y < a.rows && find_coresp<DEPTH>(a, x, y, n, d, s);
    if (ok) {  // :333-337
        row[0] = s.y * n.z - s.z * n.y, row[1] = s.z * n.x - s.x * n.z, row[2] = s.x * n.y - s.y * n.x;
        row[3] = n.x, row[4] = n.y, row[5] = n.z;
        row[6] = dot(n, d - s);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int q = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 7; ++j) {
            const float t = wave_total(row[i] * row[j]);
            if (lane == 0) sh[wave][q] = t;
            ++q;
        }
    const float cnt = wave_total(ok ? 1.f : 0.f);
    if (lane == 0) sh[wave][27] = cnt;
    __syncthreads();
    const int b = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x < 27) partial[(size_t)threadIdx.x * nblocks + b] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    if (threadIdx.x == 27 && matched) {
        const float c = (sh[0][27] + sh[1][27]) + (sh[2][27] + sh[3][27]);
        if (c > 0.f) atomicAdd(matched, (unsigned int)c);
    }
}

__global__ __launch_bounds__(64) void icp_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
    double acc = 0.0;
    const float* p = partial + (size_t)blockIdx.x * nblocks;
    for (int i = threadIdx.x; i < nblocks; i += 64) acc += (double)p[i];
    acc = wave_sum_all(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)acc;
}

}  // namespace

size_t icp_partial_floats(int cols, int rows) { return (size_t)27 * ((cols + 31) / 32) * ((rows + 7) / 8); }

hipError_t launch_icp_sums(bool depth_variant, const void* curr, int curr_step, const float* ncurr, int ncurr_step,
                           const void* prev, int prev_step, const float* nprev, int nprev_step, int cols, int rows,
                           const float aff[12], float fx, float fy, float cx, float cy, float dist_thres, float angle_thres,
                           float* partial, float* sums27, unsigned int* matched, hipStream_t s) {
    IcpArgs a;
    a.curr = curr, a.ncurr = ncurr, a.prev = prev, a.nprev = nprev;
    a.curr_step = curr_step, a.ncurr_step = ncurr_step, a.prev_step = prev_step, a.nprev_step = nprev_step;
    a.cols = cols, a.rows = rows;
    for (int i = 0; i < 9; ++i) a.aff.m[i] = aff[i];
    for (int i = 0; i < 3; ++i) a.aff.t[i] = aff[9 + i];
    a.fx = fx, a.fy = fy, a.cx = cx, a.cy = cy, a.finvx = 1.f / fx, a.finvy = 1.f / fy;  // setLevelIntr, projective_icp.cpp:15-20
    a.min_cosine  = cosf(angle_thres);                                                     // :10-13
    a.dist2_thres = dist_thres * dist_thres;
    dim3 grid((cols + 31) / 32, (rows + 7) / 8);
    const int nblocks = (int)(grid.x * grid.y);
    if (matched) {
        hipError_t e = hipMemsetAsync(matched, 0, sizeof(unsigned int), s);
        if (e != hipSuccess) return e;
    }
    if (depth_variant) icp_rows_kernel<true><<<grid, 256, 0, s>>>(a, partial, nblocks, matched);
    else icp_rows_kernel<false><<<grid, 256, 0, s>>>(a, partial, nblocks, matched);
    icp_final_kernel<<<27, 64, 0, s>>>(partial, nblocks, sums27);
    return hipGetLastError();
}

}  // namespace dfa
