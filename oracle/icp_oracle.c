/* icp_oracle.c — CPU restatement of one linearisation of the reference's rigid projective ICP:
 * find_coresp + the [s x n, n | n.(d - s)] row + the 27 sums of src/kfusion/cuda/proj_icp.cu:41-103,326-375
 * (depth variant :41-71, points variant :73-101; textures are point-sampled, :377-379).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: the reference has no test for the ICP.  The sums
 * are accumulated in double in pixel order (the reference tree-reduces floats per 32x8 tile, then per row of
 * partials: its own result depends on that order); the HIP kernel is compared with a relative tolerance. */
#include <math.h>
#include <stddef.h>
#include <string.h>

#include "oracle.h"

#define CPIX(type, base, step, y, x) (((const type*)((const char*)(base) + (size_t)(y) * (size_t)(step)))[x])

static float dot3(const float a[3], const float b[3]) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
static void aff_mul(const float A[12], const float v[3], int with_t, float o[3]) { /* R rows A[0..8], t A[9..11] */
    for (int r = 0; r < 3; ++r) o[r] = dot3(A + 3 * r, v) + (with_t ? A[9 + r] : 0.f);
}

/* returns 0 when pixel (x, y) of the current frame has a correspondence; fills n (prev normal), d (prev point),
 * s (transformed current point) */
static int find_coresp(int depth_variant, const void* curr, int curr_step, const float* ncurr, int ncurr_step,
                       const void* prev, int prev_step, const float* nprev, int nprev_step, int cols, int rows,
                       const float aff[12], float fx, float fy, float cx, float cy, float min_cosine, float dist2_thres,
                       int x, int y, float n[3], float d[3], float s[3]) {
    const float finvx = 1.f / fx, finvy = 1.f / fy;
    float p[3];
    if (depth_variant) {
        const int src_z = CPIX(uint16_t, curr, curr_step, y, x);
        if (src_z == 0) return 40;
        const float z = src_z * 0.001f;
        p[0] = z * ((float)x - cx) * finvx, p[1] = z * ((float)y - cy) * finvy, p[2] = z; /* reproj :35-39 */
    } else {
        const float* v = &CPIX(float, curr, curr_step, y, 4 * x);
        if (isnan(v[0])) return 40;
        p[0] = v[0], p[1] = v[1], p[2] = v[2];
    }
    aff_mul(aff, p, 1, s);
    const float u = fmaf(fx, s[0] / s[2], cx), w = fmaf(fy, s[1] / s[2], cy); /* proj :28-33 */
    if (s[2] <= 0 || u < 0 || w < 0 || u >= (float)cols || w >= (float)rows) return 80;
    const int iu = (int)floorf(u), iw = (int)floorf(w); /* point-sampled texture */
    if (depth_variant) {
        const int dst_z = CPIX(uint16_t, prev, prev_step, iw, iu);
        if (dst_z == 0) return 120;
        const float z = dst_z * 0.001f;
        d[0] = z * (u - cx) * finvx, d[1] = z * (w - cy) * finvy, d[2] = z; /* :57 reproj(coo.x, coo.y, .) */
    } else {
        const float* v = &CPIX(float, prev, prev_step, iw, 4 * iu);
        if (isnan(v[0])) return 120;
        d[0] = v[0], d[1] = v[1], d[2] = v[2];
    }
    const float sd[3] = {s[0] - d[0], s[1] - d[1], s[2] - d[2]};
    if (dot3(sd, sd) > dist2_thres) return 160;
    float ns[3];
    const float* nc = &CPIX(float, ncurr, ncurr_step, y, 4 * x);
    aff_mul(aff, nc, 0, ns);
    const float* np = &CPIX(float, nprev, nprev_step, iw, 4 * iu);
    n[0] = np[0], n[1] = np[1], n[2] = np[2];
    if (fabsf(dot3(ns, n)) < min_cosine) return 200;
    return 0;
}

/* sums[27]: for i in 0..5, for j in i..6: sum over pixels of row[i] * row[j], row = (s x n, n, n.(d - s))
 * (the layout ProjectiveICP::StreamHelper::get reads, projective_icp.cpp:39-57); *matched = pixels used */
void orc_icp_sums(int depth_variant, const void* curr, int curr_step, const float* ncurr, int ncurr_step, const void* prev,
                  int prev_step, const float* nprev, int nprev_step, int cols, int rows, const float aff[12], float fx,
                  float fy, float cx, float cy, float dist_thres, float angle_thres, double sums[27], long* matched) {
    const float min_cosine = (float)cos(angle_thres), dist2_thres = dist_thres * dist_thres; /* projective_icp.cpp:10-13 */
    memset(sums, 0, sizeof(double) * 27);
    long m = 0;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float n[3], d[3], s[3];
            if (find_coresp(depth_variant, curr, curr_step, ncurr, ncurr_step, prev, prev_step, nprev, nprev_step, cols, rows,
                            aff, fx, fy, cx, cy, min_cosine, dist2_thres, x, y, n, d, s))
                continue;
            ++m;
            float row[7];
            row[0] = s[1] * n[2] - s[2] * n[1], row[1] = s[2] * n[0] - s[0] * n[2], row[2] = s[0] * n[1] - s[1] * n[0];
            row[3] = n[0], row[4] = n[1], row[5] = n[2];
            const float ds[3] = {d[0] - s[0], d[1] - s[1], d[2] - s[2]};
            row[6] = dot3(n, ds);
            int q = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 7; ++j) sums[q++] += (double)(row[i] * row[j]);
        }
    if (matched) *matched = m;
}
