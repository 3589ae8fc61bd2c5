/*
 * tsdf_oracle.c — CPU restatement of kfusion's TSDF clear / integrate / raycast and
 * compute_dists.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED: the reference holds no test, golden vector or CPU implementation
 * for these kernels and its CUDA sources cannot be built here.
 * This file restates src/kfusion/cuda/tsdf_volume.cu, include/kfusion/cuda/device.hpp
 * and src/kfusion/cuda/imgproc.cu:233-245 line by line with IEEE arithmetic.
 *
 * Build with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ----------------------------------------------------------------------------------- */
/* half <-> float, round-to-nearest-even, subnormals preserved                          */

uint16_t orc_float_to_half(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t abs  = x & 0x7fffffffu;
    if (abs > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((abs >> 13) & 0x3ffu)); /* NaN (quiet) */
    if (abs >= 0x47800000u) {
        /* >= 65536: inf. Values in [65520, 65536) round to inf below via the generic path. */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (abs >= 0x38800000u) {
        /* normal half range: rebias exponent 127 -> 15 */
        uint32_t mant = abs & 0x7fffffu;
        uint32_t exp  = (abs >> 23) - 112u;
        uint32_t h    = (exp << 10) | (mant >> 13);
        uint32_t rem  = mant & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++; /* may carry into exponent / inf: correct */
        return (uint16_t)(sign | h);
    }
    if (abs < 0x33000000u) return (uint16_t)sign; /* < 2^-25: rounds to zero (2^-25 itself ties to even = 0) */
    {
        /* subnormal half: value = mant24 * 2^(e-150); half ulp = 2^-24 */
        uint32_t e     = abs >> 23;                      /* 102..112 */
        uint32_t mant  = (abs & 0x7fffffu) | 0x800000u;  /* 24-bit significand */
        uint32_t shift = 126u - e;                       /* 14..24 */
        uint32_t h     = mant >> shift;
        uint32_t rem   = mant & ((1u << shift) - 1u);
        uint32_t half  = 1u << (shift - 1u);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
}

float orc_half_to_float(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp  = (h >> 10) & 0x1fu;
    uint32_t mant = h & 0x3ffu;
    uint32_t x;
    if (exp == 0) {
        if (mant == 0) {
            x = sign;
        } else {
            /* subnormal: normalise */
            int e = -1;
            do {
                e++;
                mant <<= 1;
            } while (!(mant & 0x400u));
            x = sign | ((uint32_t)(112 - e) << 23) | ((mant & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        x = sign | 0x7f800000u | (mant << 13);
    } else {
        x = sign | ((exp + 112u) << 23) | (mant << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* ----------------------------------------------------------------------------------- */
/* small vector helpers (temp_utils.hpp / Opt cudaUtil.h float3 operators)               */

typedef struct {
    float x, y, z;
} f3;

/* a.x*b.x + a.y*b.y + a.z*b.z as nvcc contracts it */
static inline float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline f3 mk3(float x, float y, float z) {
    f3 r = {x, y, z};
    return r;
}
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
/* device.hpp:74-78 */
static inline f3 mat_mul(const float R[9], f3 v) {
    return mk3(dot3(mk3(R[0], R[1], R[2]), v), dot3(mk3(R[3], R[4], R[5]), v), dot3(mk3(R[6], R[7], R[8]), v));
}
/* temp_utils.hpp:93: v * rsqrt(dot(v,v)) with the correctly rounded 1/sqrt */
static inline f3 normalized3(f3 v) { return scale3(v, 1.0f / sqrtf(dot3(v, v))); }

/* ----------------------------------------------------------------------------------- */
/* compute_dists — imgproc.cu:233-245                                                    */

void orc_compute_dists(const uint16_t* depth, int depth_step, uint16_t* dists, int dists_step, int cols, int rows,
                       float fx, float fy, float cx, float cy) {
    /* host wrapper passes finv = 1/f (imgproc.cu:252) */
    const float finvx = 1.f / fx, finvy = 1.f / fy;
    for (int y = 0; y < rows; ++y) {
        const uint16_t* drow = (const uint16_t*)((const char*)depth + (size_t)y * depth_step);
        uint16_t* orow       = (uint16_t*)((char*)dists + (size_t)y * dists_step);
        for (int x = 0; x < cols; ++x) {
            /* the reference guard is (x<cols || y<rows) — harmless only for 32x8-aligned
             * images; the restatement uses the intended && (every pixel inside the image). */
            float xl     = ((float)x - cx) * finvx;
            float yl     = ((float)y - cy) * finvy;
            float lambda = sqrtf(fmaf(yl, yl, xl * xl) + 1.f);
            orow[x]      = orc_float_to_half(((float)drow[x] * lambda) * 0.001f);
        }
    }
}

/* ----------------------------------------------------------------------------------- */
/* clear — tsdf_volume.cu:11-22: every voxel <- pack_tsdf(0.f, 0) == 0x00000000          */

void orc_tsdf_clear(uint32_t* vol, int X, int Y, int Z) { memset(vol, 0, (size_t)X * Y * Z * sizeof(uint32_t)); }

/* ----------------------------------------------------------------------------------- */
/* integrate — tsdf_volume.cu:43-96                                                      */

/* Slices [z0, z1) of the sweep over Z slices, `vol` holding THOSE slices only (slice z0 first): every column still
 * starts at z = 0 and replays its `vc += zstep` additions below z0, so the slab is bit-for-bit the same as the
 * corresponding part of a full sweep — what makes a 1024^3 volume checkable in slabs of a few seconds each. */
long orc_tsdf_integrate_slab(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* vol, int X, int Y, int z0,
                             int z1, const float voxel_size[3], float trunc_dist, int max_weight,
                             const float vol2cam[12], float fx, float fy, float cx, float cy, int threads) {
    const float* R        = vol2cam;
    const f3 t            = mk3(vol2cam[9], vol2cam[10], vol2cam[11]);
    const float trunc_inv = 1.f / trunc_dist; /* tsdf_volume.cu:106 */
    /* :58 */
    const f3 zstep     = scale3(mk3(R[2], R[5], R[8]), voxel_size[2]);
    const size_t slice = (size_t)X * Y;
    long updated       = 0;
    /* The reference marches each (x,y) column over z with vc += zstep (:64).  The loop
     * nest here is z-outer over a block of YB rows (cache-friendly on a CPU) but each
     * column keeps its own running vc, so the float accumulation order is unchanged. */
    enum { YB = 8 };
    (void)threads;
#pragma omp parallel for schedule(static) reduction(+ : updated) num_threads(threads > 0 ? threads : 1)
    for (int yb = 0; yb < Y; yb += YB) {
        const int ye = yb + YB < Y ? yb + YB : Y;
        f3* vcs      = (f3*)malloc(sizeof(f3) * (size_t)X * YB);
        for (int y = yb; y < ye; ++y)
            for (int x = 0; x < X; ++x) {
                /* :60-61 */
                f3 vx                     = mk3((float)x * voxel_size[0], (float)y * voxel_size[1], 0.f);
                vcs[(y - yb) * X + x] = add3(mat_mul(R, vx), t);
            }
        for (int i = 0; i < z0; ++i) /* :64 on the slices below the slab */
            for (int j = 0; j < (ye - yb) * X; ++j) vcs[j] = add3(vcs[j], zstep);
        for (int i = z0; i < z1; ++i) {
            for (int y = yb; y < ye; ++y) {
                uint32_t* vrow = vol + (size_t)X * y + slice * (i - z0);
                f3* vcrow      = vcs + (size_t)(y - yb) * X;
                for (int x = 0; x < X; ++x) {
                    const f3 vc = vcrow[x];
                    vcrow[x]    = add3(vc, zstep); /* :64, also on skipped iterations */
                    /* Projector, device.hpp:40-45 */
                    float coox = fmaf(fx, vc.x / vc.z, cx);
                    float cooy = fmaf(fy, vc.y / vc.z, cy);
                    /* :70 (written so that NaN coordinates are skipped too; they only occur
                     * for vc.z == 0 which :74 skips anyway) */
                    if (!(coox >= 0.f && cooy >= 0.f && coox < (float)cols && cooy < (float)rows)) continue;
                    /* :73 point-sampled texture fetch = texel (floor x, floor y) */
                    int px = (int)floorf(coox), py = (int)floorf(cooy);
                    const uint16_t* drow = (const uint16_t*)((const char*)dists + (size_t)py * dists_step);
                    float Dp             = orc_half_to_float(drow[px]);
                    if (Dp == 0.f || vc.z <= 0.f) continue; /* :74 */
                    float sdf = Dp - sqrtf(dot3(vc, vc));   /* :77 */
                    if (sdf >= -trunc_dist) {               /* :79 */
                        float tsdf      = fminf(1.f, sdf * trunc_inv);
                        uint32_t packed = vrow[x];
                        int weight_prev = (int)(packed >> 16);
                        float tsdf_prev = orc_half_to_float((uint16_t)(packed & 0xffffu));
                        /* :86-87 */
                        float tsdf_new = fmaf(tsdf_prev, (float)weight_prev, tsdf) / (float)(weight_prev + 1);
                        int weight_new = weight_prev + 1 < max_weight ? weight_prev + 1 : max_weight;
                        vrow[x] = (uint32_t)orc_float_to_half(tsdf_new) | ((uint32_t)(uint16_t)weight_new << 16);
                        updated++;
                    }
                }
            }
        }
        free(vcs);
    }
    return updated;
}

long orc_tsdf_integrate(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* vol, int X, int Y, int Z,
                        const float voxel_size[3], float trunc_dist, int max_weight, const float vol2cam[12], float fx,
                        float fy, float cx, float cy, int threads) {
    return orc_tsdf_integrate_slab(dists, dists_step, cols, rows, vol, X, Y, 0, Z, voxel_size, trunc_dist, max_weight, vol2cam,
                                   fx, fy, cx, cy, threads);
}

/* ----------------------------------------------------------------------------------- */
/* raycast — tsdf_volume.cu:128-337                                                      */

typedef struct {
    const uint32_t* vol;
    int X, Y, Z;
    f3 voxel, voxel_inv, volume_size, gradient_delta;
    float trunc, time_step;
    const float* R; /* cam2vol rotation */
    f3 t;           /* cam2vol translation */
    const float* Rinv;
    float finvx, finvy, cx, cy;
} raycaster;

/* numeric_limits<float>::quiet_NaN() of temp_utils.hpp:22 is the bit pattern 0x7fffffff */
static inline float qnan(void) {
    const uint32_t bits = 0x7fffffffu;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline float unpack_tsdf(uint32_t p) { return orc_half_to_float((uint16_t)(p & 0xffffu)); }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* :187-193 — nearest voxel, round-half-even. The reference does no bounds check; the ray
 * is kept inside [0, size-voxel] by construction, the clamp only guards memory safety. */
static inline float fetch_tsdf(const raycaster* rc, f3 p) {
    int x = (int)lrintf(p.x * rc->voxel_inv.x);
    int y = (int)lrintf(p.y * rc->voxel_inv.y);
    int z = (int)lrintf(p.z * rc->voxel_inv.z);
    x     = clampi(x, 0, rc->X - 1);
    y     = clampi(y, 0, rc->Y - 1);
    z     = clampi(z, 0, rc->Z - 1);
    return unpack_tsdf(rc->vol[x + (size_t)rc->X * y + (size_t)rc->X * rc->Y * z]);
}

/* :146-171 */
static inline float interpolate(const raycaster* rc, f3 cf) {
    /* :151-155: g = floor(cf); out of [0, dim-2] -> NaN.  floor(c) >= 0 <=> c >= 0 and
     * floor(c) >= dim-1 <=> c >= dim-1, so the test is done in float (NaN/inf safe: the
     * device's __float2int_rd(NaN) == 0 passes the test but then yields NaN weights, i.e.
     * the same NaN result). */
    if (!(cf.x >= 0.f && cf.x < (float)(rc->X - 1) && cf.y >= 0.f && cf.y < (float)(rc->Y - 1) && cf.z >= 0.f &&
          cf.z < (float)(rc->Z - 1)))
        return qnan();
    int gx = (int)floorf(cf.x), gy = (int)floorf(cf.y), gz = (int)floorf(cf.z);
    float a = cf.x - (float)gx, b = cf.y - (float)gy, c = cf.z - (float)gz;
    const size_t sx = 1, sy = (size_t)rc->X, sz = (size_t)rc->X * rc->Y;
    const uint32_t* base = rc->vol + gx * sx + gy * sy + gz * sz;
    float tsdf = 0.f;
    /* tsdf += u * wa * wb * wc : ((u*wa)*wb) then the final multiply fused with the add */
    tsdf = fmaf((unpack_tsdf(base[0]) * (1.f - a)) * (1.f - b), (1.f - c), tsdf);
    tsdf = fmaf((unpack_tsdf(base[sz]) * (1.f - a)) * (1.f - b), c, tsdf);
    tsdf = fmaf((unpack_tsdf(base[sy]) * (1.f - a)) * b, (1.f - c), tsdf);
    tsdf = fmaf((unpack_tsdf(base[sy + sz]) * (1.f - a)) * b, c, tsdf);
    tsdf = fmaf((unpack_tsdf(base[sx]) * a) * (1.f - b), (1.f - c), tsdf);
    tsdf = fmaf((unpack_tsdf(base[sx + sz]) * a) * (1.f - b), c, tsdf);
    tsdf = fmaf((unpack_tsdf(base[sx + sy]) * a) * b, (1.f - c), tsdf);
    tsdf = fmaf((unpack_tsdf(base[sx + sy + sz]) * a) * b, c, tsdf);
    return tsdf;
}

/* :320-336 */
static inline f3 compute_normal(const raycaster* rc, f3 p) {
    f3 n;
    float Fx1 = interpolate(rc, mul3(mk3(p.x + rc->gradient_delta.x, p.y, p.z), rc->voxel_inv));
    float Fx2 = interpolate(rc, mul3(mk3(p.x - rc->gradient_delta.x, p.y, p.z), rc->voxel_inv));
    n.x       = (Fx1 - Fx2) / rc->gradient_delta.x;
    float Fy1 = interpolate(rc, mul3(mk3(p.x, p.y + rc->gradient_delta.y, p.z), rc->voxel_inv));
    float Fy2 = interpolate(rc, mul3(mk3(p.x, p.y - rc->gradient_delta.y, p.z), rc->voxel_inv));
    n.y       = (Fy1 - Fy2) / rc->gradient_delta.y;
    float Fz1 = interpolate(rc, mul3(mk3(p.x, p.y, p.z + rc->gradient_delta.z), rc->voxel_inv));
    float Fz2 = interpolate(rc, mul3(mk3(p.x, p.y, p.z - rc->gradient_delta.z), rc->voxel_inv));
    n.z       = (Fz1 - Fz2) / rc->gradient_delta.z;
    return normalized3(n);
}

/* :128-144 — note the reference's asymmetric max/min (tmin.x used twice) */
static inline void intersect(f3 org, f3 dir, f3 box_max, float* tnear, float* tfar) {
    f3 invR = mk3(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
    f3 tbot = mul3(invR, sub3(mk3(0.f, 0.f, 0.f), org));
    f3 ttop = mul3(invR, sub3(box_max, org));
    f3 tmin = mk3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
    f3 tmax = mk3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
    *tnear  = fmaxf(fmaxf(tmin.x, tmin.y), fmaxf(tmin.x, tmin.z));
    *tfar   = fminf(fminf(tmax.x, tmax.y), fminf(tmax.x, tmax.z));
}

/* shared body of the two operator() overloads (:195-318). Returns 1 on hit. */
static inline int cast_ray(const raycaster* rc, int x, int y, f3* vertex_cam, f3* normal_cam) {
    f3 ray_org = rc->t;
    /* Reprojector (device.hpp:50-54) with z = 1 */
    f3 pix     = mk3((1.f * ((float)x - rc->cx)) * rc->finvx, (1.f * ((float)y - rc->cy)) * rc->finvy, 1.f);
    f3 ray_dir = normalized3(mat_mul(rc->R, pix));
    f3 box_max = sub3(rc->volume_size, rc->voxel);
    float tmin, tmax;
    intersect(ray_org, ray_dir, box_max, &tmin, &tmax);
    tmin = fmaxf(0.f, tmin);
    if (!(tmin < tmax)) return 0; /* :220 `if (tmin >= tmax) return` ; NaN also bails */
    tmax -= rc->time_step;
    f3 vstep        = scale3(ray_dir, rc->time_step);
    f3 next         = add3(ray_org, scale3(ray_dir, tmin));
    float tsdf_next = fetch_tsdf(rc, next);
    for (float tcurr = tmin; tcurr < tmax; tcurr += rc->time_step) {
        float tsdf_curr = tsdf_next;
        f3 curr         = next;
        next            = add3(next, vstep);
        tsdf_next       = fetch_tsdf(rc, next);
        if (tsdf_curr < 0.f && tsdf_next > 0.f) break;
        if (tsdf_curr > 0.f && tsdf_next < 0.f) {
            float Ft   = interpolate(rc, mul3(curr, rc->voxel_inv));
            float Ftdt = interpolate(rc, mul3(next, rc->voxel_inv));
            float Ts   = tcurr - (rc->time_step * Ft) / (Ftdt - Ft);
            f3 vertex  = add3(ray_org, scale3(ray_dir, Ts));
            f3 normal  = compute_normal(rc, vertex);
            float prod = normal.x * normal.y * normal.z;
            if (prod == prod) { /* !isnan */
                *normal_cam = mat_mul(rc->Rinv, normal);
                *vertex_cam = mat_mul(rc->Rinv, sub3(vertex, rc->t));
                return 1;
            }
            break;
        }
    }
    return 0;
}

static void make_raycaster(raycaster* rc, const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3],
                           float trunc_dist, const float cam2vol[12], const float Rinv[9], float fx, float fy,
                           float cx, float cy, float step_factor, float delta_factor) {
    rc->vol   = vol;
    rc->X     = X;
    rc->Y     = Y;
    rc->Z     = Z;
    rc->voxel = mk3(voxel_size[0], voxel_size[1], voxel_size[2]);
    /* :359-362 (host side, plain float ops) */
    rc->volume_size    = mk3(voxel_size[0] * (float)X, voxel_size[1] * (float)Y, voxel_size[2] * (float)Z);
    rc->time_step      = trunc_dist * step_factor;
    rc->gradient_delta = scale3(rc->voxel, delta_factor);
    rc->voxel_inv      = mk3(1.f / voxel_size[0], 1.f / voxel_size[1], 1.f / voxel_size[2]);
    rc->trunc          = trunc_dist;
    rc->R              = cam2vol;
    rc->t              = mk3(cam2vol[9], cam2vol[10], cam2vol[11]);
    rc->Rinv           = Rinv;
    /* Reprojector ctor (precomp.cpp): finv = 1/f */
    rc->finvx = 1.f / fx;
    rc->finvy = 1.f / fy;
    rc->cx    = cx;
    rc->cy    = cy;
}

/* normals of given points (volume metric frame) by the raycaster's compute_normal (:320-336); points / normals are
 * n x 4 floats */
void orc_tsdf_vertex_normals(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float delta_factor,
                             const float* points, int n, float* normals) {
    static const float id12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, id9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    raycaster rc;
    make_raycaster(&rc, vol, X, Y, Z, voxel_size, 1.f, id12, id9, 1.f, 1.f, 0.f, 0.f, 1.f, delta_factor);
    for (int i = 0; i < n; ++i) {
        const f3 nn        = compute_normal(&rc, mk3(points[4 * i], points[4 * i + 1], points[4 * i + 2]));
        normals[4 * i]     = nn.x;
        normals[4 * i + 1] = nn.y;
        normals[4 * i + 2] = nn.z;
        normals[4 * i + 3] = 0.f;
    }
}

void orc_tsdf_raycast_points(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                             const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                             float step_factor, float delta_factor, float* points, int points_step, float* normals,
                             int normals_step, int cols, int rows, int threads) {
    raycaster rc;
    make_raycaster(&rc, vol, X, Y, Z, voxel_size, trunc_dist, cam2vol, Rinv, fx, fy, cx, cy, step_factor,
                   delta_factor);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
    for (int y = 0; y < rows; ++y) {
        float* prow = (float*)((char*)points + (size_t)y * points_step);
        float* nrow = (float*)((char*)normals + (size_t)y * normals_step);
        for (int x = 0; x < cols; ++x) {
            f3 v, n;
            if (cast_ray(&rc, x, y, &v, &n)) {
                prow[4 * x + 0] = v.x, prow[4 * x + 1] = v.y, prow[4 * x + 2] = v.z, prow[4 * x + 3] = 0.f;
                nrow[4 * x + 0] = n.x, nrow[4 * x + 1] = n.y, nrow[4 * x + 2] = n.z, nrow[4 * x + 3] = 0.f;
            } else {
                for (int c = 0; c < 4; ++c) prow[4 * x + c] = nrow[4 * x + c] = qnan();
            }
        }
    }
}

void orc_tsdf_raycast_depth(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                            const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                            float step_factor, float delta_factor, uint16_t* depth, int depth_step, float* normals,
                            int normals_step, int cols, int rows, int threads) {
    raycaster rc;
    make_raycaster(&rc, vol, X, Y, Z, voxel_size, trunc_dist, cam2vol, Rinv, fx, fy, cx, cy, step_factor,
                   delta_factor);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
    for (int y = 0; y < rows; ++y) {
        uint16_t* drow = (uint16_t*)((char*)depth + (size_t)y * depth_step);
        float* nrow    = (float*)((char*)normals + (size_t)y * normals_step);
        for (int x = 0; x < cols; ++x) {
            f3 v, n;
            if (cast_ray(&rc, x, y, &v, &n)) {
                nrow[4 * x + 0] = n.x, nrow[4 * x + 1] = n.y, nrow[4 * x + 2] = n.z, nrow[4 * x + 3] = 0.f;
                /* :251 static_cast<ushort>(vertex.z * 1000): truncating, saturating convert */
                float mm = v.z * 1000.f;
                mm       = mm < 0.f ? 0.f : (mm > 65535.f ? 65535.f : mm); /* cvt.rzi.u16.f32 saturates */
                drow[x]  = (uint16_t)(int)mm;
            } else {
                drow[x] = 0;
                for (int c = 0; c < 4; ++c) nrow[4 * x + c] = qnan();
            }
        }
    }
}
