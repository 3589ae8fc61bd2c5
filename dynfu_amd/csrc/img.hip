// img.hip — depth pre-processing kernels of the reference's src/kfusion/cuda/imgproc.cu for gfx950:
// bilateral filter (:8-52), depth truncation (:60-79), depth pyramid (:84-124), normals + depth mask
// (:129-183), 2x2 down-samplers (:258-311, :314-358).  (compute_dists lives in tsdf.hip,
// computePointNormals in solve6.hip.)
//
// All of them are small image passes (VGA: 0.3 M pixels; a launch is a few microseconds), one lane per
// output pixel with the reference's arithmetic: integer window sums, IEEE fp32 without contraction.  The
// bilateral weight uses exp_neg() below instead of CUDA's __expf — a fixed sequence of IEEE operations
// that a CPU can repeat exactly, so the filtered depth is bit-identical to the tests' CPU restatement.
// The window of the bilateral filter is staged through LDS (a 32 x 8 tile + halo): every input pixel is
// fetched once per tile instead of ksz^2 times.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "kernels.hpp"

namespace dfa {

namespace {

// exp(x), x <= 0: 2^(x log2 e) = 2^n 2^f with n = rint, degree-6 Taylor of 2^f in Horner form with fused
// multiply-adds.  The CPU restatement used by the tests repeats these operations one for one.
__device__ __forceinline__ float exp_neg(float x) {
    const float t = x * 1.44269504088896341f;
    if (!(t >= -126.0f)) return 0.0f;
    const float n = rintf(t), f = t - n;
    float p = 0.00015403530393381608f;
    p = fmaf(p, f, 0.0013333558146428443f);
    p = fmaf(p, f, 0.009618129107628477f);
    p = fmaf(p, f, 0.05550410866482158f);
    p = fmaf(p, f, 0.2402265069591007f);
    p = fmaf(p, f, 0.6931471805599453f);
    p = fmaf(p, f, 1.0f);
    return p * __int_as_float(((int)n + 127) << 23);
}

template <class T>
__device__ __forceinline__ T& pix(T* base, int step, int y, int x) {
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)y * step + sizeof(T) * (size_t)x);
}
template <class T>
__device__ __forceinline__ const T& cpix(const T* base, int step, int y, int x) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)y * step + sizeof(T) * (size_t)x);
}

constexpr int BIL_MAX_K = 15;  // window sizes up to 15 use the LDS tile (the reference's default is 7)

__global__ __launch_bounds__(256) void bilateral_kernel(const uint16_t* __restrict__ src, int src_step,
                                                        uint16_t* __restrict__ dst, int dst_step, int cols, int rows,
                                                        int ksz, float ss, float sd) {
    __shared__ uint16_t tile[8 + BIL_MAX_K][32 + BIL_MAX_K + 1];
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x0 = blockIdx.x * 32 - ksz / 2, y0 = blockIdx.y * 8 - ksz / 2;  // tile origin in the image
    const bool staged = ksz <= BIL_MAX_K;
    if (staged) {
        const int tw = 32 + ksz, th = 8 + ksz;
        for (int i = threadIdx.x; i < tw * th; i += 256) {
            const int ty = i / tw, tx = i - ty * tw, gx = x0 + tx, gy = y0 + ty;
            tile[ty][tx] = (gx >= 0 && gy >= 0 && gx < cols && gy < rows) ? cpix(src, src_step, gy, gx) : (uint16_t)0;
        }
        __syncthreads();
    }
    const int x = blockIdx.x * 32 + lx, y = blockIdx.y * 8 + ly;
    if (x >= cols || y >= rows) return;
    const int value = cpix(src, src_step, y, x);
    const int tx = min(x - ksz / 2 + ksz, cols - 1), ty = min(y - ksz / 2 + ksz, rows - 1);  // :17-18
    float sum1 = 0.f, sum2 = 0.f;
    for (int cy = max(y - ksz / 2, 0); cy < ty; ++cy)
        for (int cx = max(x - ksz / 2, 0); cx < tx; ++cx) {
            const int depth    = staged ? (int)tile[cy - y0][cx - x0] : (int)cpix(src, src_step, cy, cx);
            const float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
            // :28 squares in int; the float product is the same number while the int one does not overflow
            // (|d| <= 46 340 mm) and stays defined beyond, where the reference's wraps around
            const float color2 = (float)(value - depth) * (float)(value - depth);
            const float weight = exp_neg(-(space2 * ss + color2 * sd));  // :30
            sum1 += (float)depth * weight;
            sum2 += weight;
        }
    const float q = sum1 / sum2;
    pix(dst, dst_step, y, x) = (uint16_t)(q != q ? 0 : (int)rintf(q));  // __float2int_rn; NaN -> 0
}

__global__ __launch_bounds__(256) void truncate_depth_kernel(uint16_t* __restrict__ depth, int step, int cols, int rows,
                                                             uint16_t max_dist) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x < cols && y < rows && pix(depth, step, y, x) > max_dist) pix(depth, step, y, x) = 0;  // :64-66
}

__global__ __launch_bounds__(256) void pyramid_kernel(const uint16_t* __restrict__ src, int src_step, int cols, int rows,
                                                      uint16_t* __restrict__ dst, int dst_step, float s3) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= cols / 2 || y >= rows / 2) return;
    constexpr int D  = 5;
    const int center = cpix(src, src_step, 2 * y, 2 * x);
    const int tx = min(2 * x - D / 2 + D, cols - 1), ty = min(2 * y - D / 2 + D, rows - 1);
    int sum = 0, count = 0;
    for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
            const int val = cpix(src, src_step, cy, cx);
            if ((float)abs(val - center) < s3) sum += val, ++count;  // :103
        }
    pix(dst, dst_step, y, x) = (uint16_t)(count == 0 ? 0 : sum / count);
}

__global__ __launch_bounds__(256) void normals_kernel(const uint16_t* __restrict__ depth, int depth_step, int cols, int rows,
                                                      float finvx, float finvy, float cx, float cy,
                                                      float* __restrict__ normals, int normals_step) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= cols || y >= rows) return;
    const float qnan = __builtin_nanf("");
    float4 n_out = make_float4(qnan, qnan, qnan, 0.f);  // :139
    if (x < cols - 1 && y < rows - 1) {
        const float z00 = cpix(depth, depth_step, y, x) * 0.001f, z01 = cpix(depth, depth_step, y, x + 1) * 0.001f,
                    z10 = cpix(depth, depth_step, y + 1, x) * 0.001f;
        if (z00 * z01 * z10 != 0.f) {
            const f3 v00 = mk3(z00 * ((float)x - cx) * finvx, z00 * ((float)y - cy) * finvy, z00);
            const f3 v01 = mk3(z01 * ((float)(x + 1) - cx) * finvx, z01 * ((float)y - cy) * finvy, z01);
            const f3 v10 = mk3(z10 * ((float)x - cx) * finvx, z10 * ((float)(y + 1) - cy) * finvy, z10);
            const f3 a = v01 - v00, b = v10 - v00;
            const f3 n = normalized(mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x));
            n_out      = make_float4(-n.x, -n.y, -n.z, 0.f);
        }
    }
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(normals) + (size_t)y * normals_step + 16 * (size_t)x) = n_out;
}

// second launch: the mask of pixel (x, y) needs only its own normal, but the normals of its left / upper
// neighbours read this pixel's depth — masking inside the first kernel would race
__global__ __launch_bounds__(256) void mask_depth_kernel(const float* __restrict__ normals, int normals_step,
                                                         uint16_t* __restrict__ depth, int depth_step, int cols, int rows) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= cols || y >= rows) return;
    const float nx = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(normals) + (size_t)y * normals_step + 16 * (size_t)x);
    if (nx != nx) pix(depth, depth_step, y, x) = 0;  // :165-166
}

__device__ __forceinline__ float4 ld4(const float* base, int step, int y, int x) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (size_t)y * step + 16 * (size_t)x);
}
__device__ __forceinline__ void st4(float* base, int step, int y, int x, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + (size_t)y * step + 16 * (size_t)x) = v;
}

__global__ __launch_bounds__(256) void resize_depth_normals_kernel(const uint16_t* __restrict__ dsrc, int dsrc_step,
                                                                   const float* __restrict__ nsrc, int nsrc_step,
                                                                   int out_cols, int out_rows, uint16_t* __restrict__ ddst,
                                                                   int ddst_step, float* __restrict__ ndst, int ndst_step) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= out_cols || y >= out_rows) return;
    const float qnan = __builtin_nanf("");
    uint16_t d = 0;
    float4 n   = make_float4(qnan, qnan, qnan, qnan);
    const int xs = 2 * x, ys = 2 * y;
    const int d00 = cpix(dsrc, dsrc_step, ys, xs), d01 = cpix(dsrc, dsrc_step, ys, xs + 1),
              d10 = cpix(dsrc, dsrc_step, ys + 1, xs), d11 = cpix(dsrc, dsrc_step, ys + 1, xs + 1);
    if (d00 * d01 != 0 && d10 * d11 != 0) {  // :279
        d = (uint16_t)((d00 + d01 + d10 + d11) / 4);
        const float4 n00 = ld4(nsrc, nsrc_step, ys, xs), n01 = ld4(nsrc, nsrc_step, ys, xs + 1),
                     n10 = ld4(nsrc, nsrc_step, ys + 1, xs), n11 = ld4(nsrc, nsrc_step, ys + 1, xs + 1);
        n.x = (((n00.x + n01.x) + n10.x) + n11.x) * 0.25f;
        n.y = (((n00.y + n01.y) + n10.y) + n11.y) * 0.25f;
        n.z = (((n00.z + n01.z) + n10.z) + n11.z) * 0.25f;
    }
    pix(ddst, ddst_step, y, x) = d;
    st4(ndst, ndst_step, y, x, n);
}

__global__ __launch_bounds__(256) void resize_points_normals_kernel(const float* __restrict__ vsrc, int vsrc_step,
                                                                    const float* __restrict__ nsrc, int nsrc_step,
                                                                    int out_cols, int out_rows, float* __restrict__ vdst,
                                                                    int vdst_step, float* __restrict__ ndst, int ndst_step) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= out_cols || y >= out_rows) return;
    const float qnan = __builtin_nanf("");
    float4 v = make_float4(qnan, qnan, qnan, 0.f), n = v;  // :323
    const int xs = 2 * x, ys = 2 * y;
    const float4 p00 = ld4(vsrc, vsrc_step, ys, xs), p01 = ld4(vsrc, vsrc_step, ys, xs + 1),
                 p10 = ld4(vsrc, vsrc_step, ys + 1, xs), p11 = ld4(vsrc, vsrc_step, ys + 1, xs + 1);
    const float prod = p00.x * p01.x * p10.x * p11.x;
    if (prod == prod) {  // :333
        const float4 q00 = ld4(nsrc, nsrc_step, ys, xs), q01 = ld4(nsrc, nsrc_step, ys, xs + 1),
                     q10 = ld4(nsrc, nsrc_step, ys + 1, xs), q11 = ld4(nsrc, nsrc_step, ys + 1, xs + 1);
        v = make_float4((((p00.x + p01.x) + p10.x) + p11.x) * 0.25f, (((p00.y + p01.y) + p10.y) + p11.y) * 0.25f,
                        (((p00.z + p01.z) + p10.z) + p11.z) * 0.25f, 0.f);
        n = make_float4((((q00.x + q01.x) + q10.x) + q11.x) * 0.25f, (((q00.y + q01.y) + q10.y) + q11.y) * 0.25f,
                        (((q00.z + q01.z) + q10.z) + q11.z) * 0.25f, 0.f);
    }
    st4(vdst, vdst_step, y, x, v);
    st4(ndst, ndst_step, y, x, n);
}

dim3 img_grid(int cols, int rows) { return dim3((cols + 31) / 32, (rows + 7) / 8); }

}  // namespace

hipError_t launch_bilateral(const uint16_t* src, int src_step, uint16_t* dst, int dst_step, int cols, int rows, int ksz,
                            float sigma_spatial, float sigma_depth, hipStream_t s) {
    sigma_depth *= 1000;  // metres -> mm (:43)
    bilateral_kernel<<<img_grid(cols, rows), 256, 0, s>>>(src, src_step, dst, dst_step, cols, rows, ksz,
                                                          0.5f / (sigma_spatial * sigma_spatial),
                                                          0.5f / (sigma_depth * sigma_depth));
    return hipGetLastError();
}

hipError_t launch_truncate_depth(uint16_t* depth, int step, int cols, int rows, float max_dist, hipStream_t s) {
    truncate_depth_kernel<<<img_grid(cols, rows), 256, 0, s>>>(depth, step, cols, rows, (uint16_t)(max_dist * 1000.f));
    return hipGetLastError();
}

hipError_t launch_depth_pyr(const uint16_t* src, int src_step, int cols, int rows, uint16_t* dst, int dst_step,
                            float sigma_depth, hipStream_t s) {
    if (cols / 2 == 0 || rows / 2 == 0) return hipSuccess;
    pyramid_kernel<<<img_grid(cols / 2, rows / 2), 256, 0, s>>>(src, src_step, cols, rows, dst, dst_step,
                                                                sigma_depth * 1000 * 3);
    return hipGetLastError();
}

hipError_t launch_normals_mask_depth(uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                                     float cy, float* normals, int normals_step, hipStream_t s) {
    normals_kernel<<<img_grid(cols, rows), 256, 0, s>>>(depth, depth_step, cols, rows, 1.f / fx, 1.f / fy, cx, cy, normals,
                                                        normals_step);
    mask_depth_kernel<<<img_grid(cols, rows), 256, 0, s>>>(normals, normals_step, depth, depth_step, cols, rows);
    return hipGetLastError();
}

hipError_t launch_resize_depth_normals(const uint16_t* dsrc, int dsrc_step, const float* nsrc, int nsrc_step, int cols,
                                       int rows, uint16_t* ddst, int ddst_step, float* ndst, int ndst_step, hipStream_t s) {
    if (cols / 2 == 0 || rows / 2 == 0) return hipSuccess;
    resize_depth_normals_kernel<<<img_grid(cols / 2, rows / 2), 256, 0, s>>>(dsrc, dsrc_step, nsrc, nsrc_step, cols / 2,
                                                                             rows / 2, ddst, ddst_step, ndst, ndst_step);
    return hipGetLastError();
}

hipError_t launch_resize_points_normals(const float* vsrc, int vsrc_step, const float* nsrc, int nsrc_step, int cols,
                                        int rows, float* vdst, int vdst_step, float* ndst, int ndst_step, hipStream_t s) {
    if (cols / 2 == 0 || rows / 2 == 0) return hipSuccess;
    resize_points_normals_kernel<<<img_grid(cols / 2, rows / 2), 256, 0, s>>>(vsrc, vsrc_step, nsrc, nsrc_step, cols / 2,
                                                                              rows / 2, vdst, vdst_step, ndst, ndst_step);
    return hipGetLastError();
}

}  // namespace dfa
