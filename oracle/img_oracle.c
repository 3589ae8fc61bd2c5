/* img_oracle.c — CPU restatement of the depth pre-processing kernels of src/kfusion/cuda/imgproc.cu
 * (bilateral filter :8-38, depth truncation :60-68, depth pyramid :84-111, normals + depth mask
 * :129-183, the two 2x2 down-samplers :258-293 and :314-347).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: the reference has no test or golden vector
 * for any of them; this file restates the kernels line by line and is itself the pin of the HIP kernels.
 *
 * Numerics contract (same as oracle.h): IEEE fp32, no contraction; correctly rounded `/` where the CUDA
 * build divides approximately; and for the bilateral weight the reference's `__expf` (a hardware
 * approximation with no portable definition) is replaced by orc_exp_neg below — a fixed sequence of
 * IEEE operations that the HIP kernel repeats verbatim, so the filtered depth is bit-identical. */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* exp(x) for x <= 0: 2^(x log2 e) = 2^n 2^f, n = rint, |f| <= 1/2, degree-6 Taylor of 2^f in Horner form with
 * fused multiply-adds; ~1e-7 relative.  Below 2^-126 the result is 0. */
float orc_exp_neg(float x) {
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
    union {
        uint32_t u;
        float f;
    } s;
    s.u = (uint32_t)((int)n + 127) << 23;
    return p * s.f;
}

#define PIX(type, base, step, y, x) (((type*)((char*)(base) + (size_t)(y) * (size_t)(step)))[x])
#define CPIX(type, base, step, y, x) (((const type*)((const char*)(base) + (size_t)(y) * (size_t)(step)))[x])


/* bilateral_kernel :8-38 + host :41-52 (sigma_depth metres -> mm, 0.5 / sigma^2) */
void orc_bilateral(const uint16_t* src, int src_step, uint16_t* dst, int dst_step, int cols, int rows, int ksz,
                   float sigma_spatial, float sigma_depth) {
    sigma_depth *= 1000;
    const float ss = 0.5f / (sigma_spatial * sigma_spatial), sd = 0.5f / (sigma_depth * sigma_depth);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const int value = CPIX(uint16_t, src, src_step, y, x);
            const int tx = (x - ksz / 2 + ksz) < cols - 1 ? (x - ksz / 2 + ksz) : cols - 1; /* :17-18 */
            const int ty = (y - ksz / 2 + ksz) < rows - 1 ? (y - ksz / 2 + ksz) : rows - 1;
            float sum1 = 0, sum2 = 0;
            for (int cy = (y - ksz / 2) > 0 ? (y - ksz / 2) : 0; cy < ty; ++cy)
                for (int cx = (x - ksz / 2) > 0 ? (x - ksz / 2) : 0; cx < tx; ++cx) {
                    const int depth    = CPIX(uint16_t, src, src_step, cy, cx);
                    const float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
                    /* :28 squares in int; (float)d * (float)d is the same number while the int product does not
                     * overflow (|d| <= 46 340 mm) and stays defined beyond, where the reference's wraps around */
                    const float color2 = (float)(value - depth) * (float)(value - depth);
                    const float weight = orc_exp_neg(-(space2 * ss + color2 * sd)); /* :30 */
                    sum1 += (float)depth * weight;
                    sum2 += weight;
                }
            const float q = sum1 / sum2;
            const int r   = isnan(q) ? 0 : (int)rintf(q); /* __float2int_rn; NaN -> 0 */
            PIX(uint16_t, dst, dst_step, y, x) = (uint16_t)r;
        }
}

/* truncate_depth_kernel :60-68, host :73-79 */
void orc_truncate_depth(uint16_t* depth, int step, int cols, int rows, float max_dist) {
    const uint16_t md = (uint16_t)(max_dist * 1000.f);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            if (PIX(uint16_t, depth, step, y, x) > md) PIX(uint16_t, depth, step, y, x) = 0;
}

/* pyramid_kernel :84-111, host :116-124; dst is (rows/2) x (cols/2) */
void orc_depth_pyr(const uint16_t* src, int src_step, int cols, int rows, uint16_t* dst, int dst_step, float sigma_depth) {
    const float s3 = sigma_depth * 1000 * 3;
    const int dc = cols / 2, dr = rows / 2, D = 5;
    for (int y = 0; y < dr; ++y)
        for (int x = 0; x < dc; ++x) {
            const int center = CPIX(uint16_t, src, src_step, 2 * y, 2 * x);
            const int tx = (2 * x - D / 2 + D) < cols - 1 ? (2 * x - D / 2 + D) : cols - 1;
            const int ty = (2 * y - D / 2 + D) < rows - 1 ? (2 * y - D / 2 + D) : rows - 1;
            int sum = 0, count = 0;
            for (int cy = (2 * y - D / 2) > 0 ? (2 * y - D / 2) : 0; cy < ty; ++cy)
                for (int cx = (2 * x - D / 2) > 0 ? (2 * x - D / 2) : 0; cx < tx; ++cx) {
                    const int val = CPIX(uint16_t, src, src_step, cy, cx);
                    if ((float)abs(val - center) < s3) sum += val, ++count;
                }
            PIX(uint16_t, dst, dst_step, y, x) = (uint16_t)(count == 0 ? 0 : sum / count);
        }
}

/* compute_normals_kernel :129-157 + mask_depth_kernel :159-168 (the mask's `x < cols || y < rows` is read as
 * the in-bounds test it stands for) */
void orc_normals_mask_depth(uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx, float cy,
                            float* normals, int normals_step) {
    const float finvx = 1.f / fx, finvy = 1.f / fy;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float n_out[4] = {NAN, NAN, NAN, 0.f};
            if (x < cols - 1 && y < rows - 1) {
                const float z00 = PIX(uint16_t, depth, depth_step, y, x) * 0.001f;
                const float z01 = PIX(uint16_t, depth, depth_step, y, x + 1) * 0.001f;
                const float z10 = PIX(uint16_t, depth, depth_step, y + 1, x) * 0.001f;
                if (z00 * z01 * z10 != 0) {
                    const float v00[3] = {z00 * ((float)x - cx) * finvx, z00 * ((float)y - cy) * finvy, z00};
                    const float v01[3] = {z01 * ((float)(x + 1) - cx) * finvx, z01 * ((float)y - cy) * finvy, z01};
                    const float v10[3] = {z10 * ((float)x - cx) * finvx, z10 * ((float)(y + 1) - cy) * finvy, z10};
                    const float a[3] = {v01[0] - v00[0], v01[1] - v00[1], v01[2] - v00[2]};
                    const float b[3] = {v10[0] - v00[0], v10[1] - v00[1], v10[2] - v00[2]};
                    float n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
                    const float dd  = fmaf(n[2], n[2], fmaf(n[1], n[1], n[0] * n[0]));
                    const float inv = 1.0f / sqrtf(dd);
                    n_out[0] = -(n[0] * inv), n_out[1] = -(n[1] * inv), n_out[2] = -(n[2] * inv);
                }
            }
            memcpy(&PIX(float, normals, normals_step, y, 4 * x), n_out, sizeof(n_out));
        }
    for (int y = 0; y < rows; ++y) /* the mask runs after ALL normals exist (second kernel) */
        for (int x = 0; x < cols; ++x)
            if (isnan(PIX(float, normals, normals_step, y, 4 * x))) PIX(uint16_t, depth, depth_step, y, x) = 0;
}

/* resize_depth_normals_kernel :258-293; outputs (rows/2) x (cols/2) */
void orc_resize_depth_normals(const uint16_t* dsrc, int dsrc_step, const float* nsrc, int nsrc_step, int cols, int rows,
                              uint16_t* ddst, int ddst_step, float* ndst, int ndst_step) {
    for (int y = 0; y < rows / 2; ++y)
        for (int x = 0; x < cols / 2; ++x) {
            const int xs = 2 * x, ys = 2 * y;
            uint16_t d = 0;
            float n[4] = {NAN, NAN, NAN, NAN};
            const int d00 = CPIX(uint16_t, dsrc, dsrc_step, ys, xs), d01 = CPIX(uint16_t, dsrc, dsrc_step, ys, xs + 1);
            const int d10 = CPIX(uint16_t, dsrc, dsrc_step, ys + 1, xs), d11 = CPIX(uint16_t, dsrc, dsrc_step, ys + 1, xs + 1);
            if (d00 * d01 != 0 && d10 * d11 != 0) {
                d = (uint16_t)((d00 + d01 + d10 + d11) / 4);
                for (int c = 0; c < 3; ++c)
                    n[c] = (CPIX(float, nsrc, nsrc_step, ys, 4 * xs + c) + CPIX(float, nsrc, nsrc_step, ys, 4 * (xs + 1) + c) +
                            CPIX(float, nsrc, nsrc_step, ys + 1, 4 * xs + c) + CPIX(float, nsrc, nsrc_step, ys + 1, 4 * (xs + 1) + c)) *
                           0.25f;
            }
            PIX(uint16_t, ddst, ddst_step, y, x) = d;
            memcpy(&PIX(float, ndst, ndst_step, y, 4 * x), n, sizeof(n));
        }
}

/* resize_points_normals_kernel :314-347 */
void orc_resize_points_normals(const float* vsrc, int vsrc_step, const float* nsrc, int nsrc_step, int cols, int rows,
                               float* vdst, int vdst_step, float* ndst, int ndst_step) {
    for (int y = 0; y < rows / 2; ++y)
        for (int x = 0; x < cols / 2; ++x) {
            const int xs = 2 * x, ys = 2 * y;
            float v[4] = {NAN, NAN, NAN, 0.f}, n[4] = {NAN, NAN, NAN, 0.f};
            const float* p00 = &CPIX(float, vsrc, vsrc_step, ys, 4 * xs);
            const float* p01 = &CPIX(float, vsrc, vsrc_step, ys, 4 * (xs + 1));
            const float* p10 = &CPIX(float, vsrc, vsrc_step, ys + 1, 4 * xs);
            const float* p11 = &CPIX(float, vsrc, vsrc_step, ys + 1, 4 * (xs + 1));
            if (!isnan(p00[0] * p01[0] * p10[0] * p11[0])) {
                const float* q00 = &CPIX(float, nsrc, nsrc_step, ys, 4 * xs);
                const float* q01 = &CPIX(float, nsrc, nsrc_step, ys, 4 * (xs + 1));
                const float* q10 = &CPIX(float, nsrc, nsrc_step, ys + 1, 4 * xs);
                const float* q11 = &CPIX(float, nsrc, nsrc_step, ys + 1, 4 * (xs + 1));
                for (int c = 0; c < 3; ++c) {
                    v[c] = (((p00[c] + p01[c]) + p10[c]) + p11[c]) * 0.25f;
                    n[c] = (((q00[c] + q01[c]) + q10[c]) + q11[c]) * 0.25f;
                }
            }
            memcpy(&PIX(float, vdst, vdst_step, y, 4 * x), v, sizeof(v));
            memcpy(&PIX(float, ndst, ndst_step, y, 4 * x), n, sizeof(n));
        }
}
