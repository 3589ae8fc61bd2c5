/*
 * warp_oracle.c — CPU restatement of the warp-field model: k-NN of deformation nodes,
 * radial-basis transformation weight, the reference's dual-quaternion "blend"
 * (an ordered product, see below) and warpToLive.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows src/dynfu/warp_field.cpp:99-171, src/dynfu/utils/node.cpp:19-36.
 * k-NN is pinned against the reference's vendored nanoflann (oracle/_ref/libref_knn.so,
 * tests/test_oracle_knn.py); the DQ algebra by tests/test_oracle_dq.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ------------------------------------------------------------------------------------ */
/* warp_field.cpp:111-122 (nanoflann knnSearch, L2_Simple_Adaptor: squared distance
 * accumulated as d0*d0 + d1*d1 + d2*d2 in float, nanoflann.hpp L2_Simple_Adaptor::accum_dist) */

static inline float dist2(const float* a, const float* b) {
    float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    return (d0 * d0 + d1 * d1) + d2 * d2;
}

static void knn_one(const float* nodes, int D, const float* q, int k, int32_t* idx, float* dist) {
    int count = 0;
    for (int i = 0; i < D; ++i) {
        float d = dist2(q, nodes + 3 * i);
        /* KNNResultSet::addPoint (nanoflann.hpp:88-122): stable insertion, ascending;
         * an equal distance is placed AFTER the ones already held */
        int j;
        for (j = count; j > 0; --j) {
            if (dist[j - 1] > d) {
                if (j < k) {
                    dist[j] = dist[j - 1];
                    idx[j]  = idx[j - 1];
                }
            } else
                break;
        }
        if (j < k) {
            dist[j] = d;
            idx[j]  = i;
        }
        if (count < k) count++;
    }
    for (int j = count; j < k; ++j) idx[j] = -1;
}

void orc_knn(const float* nodes, int D, const float* query, int n_query, int k, int32_t* idx, int threads) {
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int v = 0; v < n_query; ++v) {
        float dist[64];
        knn_one(nodes, D, query + 3 * (size_t)v, k, idx + (size_t)v * k, dist);
    }
}

/* ------------------------------------------------------------------------------------ */
/* node.cpp:29-36 — pow(float,int) and exp() are evaluated in double, result stored to float */

float orc_transformation_weight(const float g[3], float dg_w, const float v[3]) {
    double dx = (double)(g[0] - v[0]), dy = (double)(g[1] - v[1]), dz = (double)(g[2] - v[2]);
    double dist_sq = dx * dx + dy * dy + dz * dz;
    double w       = (double)dg_w;
    return (float)exp(-dist_sq / (2 * (w * w)));
}

/* ------------------------------------------------------------------------------------ */
/* warp_field.cpp:127-148 */

void orc_calc_dqb(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, const float p[3],
                  float out[8]) {
    int32_t idx[64];
    float dist[64];
    knn_one(node_pos, D, p, k, idx, dist);
    /* transformationSum(0,0,0,0,0,0) = identity (:133) */
    float sum[8] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < k; ++j) {
        int n = idx[j];
        if (n < 0) break;
        float w = orc_transformation_weight(node_pos + 3 * n, node_w[n], p);
        float weighted[8], prod[8];
        orc_dq_scale(node_dq + 8 * n, w, weighted); /* :139 dual part scaled only */
        orc_dq_mul(sum, weighted, prod);            /* :141 transformationSum *= weighted */
        memcpy(sum, prod, sizeof(sum));
    }
    orc_dq_normalize(sum, out); /* :145 */
}

/* warp_field.cpp:150-171 */
void orc_warp_to_live(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                      const float* verts, const float* normals, int N, float* out_verts, float* out_normals,
                      int threads) {
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int i = 0; i < N; ++i) {
        float dq[8];
        orc_calc_dqb(node_pos, node_dq, node_w, D, k, verts + 3 * (size_t)i, dq);
        orc_dq_transform_vertex(dq, verts + 3 * (size_t)i, out_verts + 3 * (size_t)i);
        /* transformNormal uses the same formula as transformVertex (dual_quaternion.hpp:217-228) */
        if (normals && out_normals)
            orc_dq_transform_vertex(dq, normals + 3 * (size_t)i, out_normals + 3 * (size_t)i);
    }
}

/* Warpfield::getUnsupportedVertices (src/dynfu/warp_field.cpp:34-62): flag = 1 when the smallest
 * |v - dg_v| / dg_w over the k nearest nodes is >= 1 (or there is no node). */
void orc_unsupported_flags(const float* node_pos, const float* node_w, int D, int k, const float* verts, int N,
                           uint8_t* flags, int threads) {
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int v = 0; v < N; ++v) {
        int32_t idx[64];
        float d2[64];
        float mn = HUGE_VALF; /* :40 */
        if (D > 0) {
            knn_one(node_pos, D, verts + 3 * (size_t)v, k, idx, d2);
            for (int j = 0; j < k && idx[j] >= 0; ++j) {
                const float* g = node_pos + 3 * (size_t)idx[j];
                /* :45-46: pow(float difference, 2) is double; the root is assigned to a float */
                const double dx = (double)(verts[3 * (size_t)v] - g[0]), dy = (double)(verts[3 * (size_t)v + 1] - g[1]),
                             dz = (double)(verts[3 * (size_t)v + 2] - g[2]);
                const float dist = (float)sqrt(dx * dx + dy * dy + dz * dz);
                const float q    = dist / node_w[idx[j]];
                if (q <= mn) mn = q;
            }
        }
        flags[v] = mn >= 1.f;
    }
}

/* pcl::VoxelGrid<pcl::PointXYZ>::applyFilter as Warpfield::update uses it (warp_field.cpp:68-72: leaf 0.05,
 * every other setting default).  PCL is an un-vendored dependency of the reference (no version pin in the
 * tree: whatever libpcl-dev the Dockerfile's apt installs — 1.7 / 1.8); this restates its published
 * algorithm (filters/include/pcl/filters/impl/voxel_grid.hpp, applyFilter):
 *   min_b = floor(min_p * inverse_leaf), max_b likewise, div_b = max_b - min_b + 1,
 *   cell of a point = floor(p * inverse_leaf) - min_b, linear index x + y div_b.x + z div_b.x div_b.y,
 *   points sorted by cell index, one output point per occupied cell in ascending index:
 *   the centroid (float sum / count).
 * PCL's std::sort leaves the order INSIDE a cell unspecified, so the float sum's last bit is not defined by
 * PCL; here the points of a cell are added in ascending input order.  PARITY UNPINNED.
 * Returns the number of output points (out must hold n x 3). */
typedef struct {
    long cell;
    int point;
} vg_pair;
static int vg_cmp(const void* a, const void* b) {
    const vg_pair *x = (const vg_pair*)a, *y = (const vg_pair*)b;
    if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
    return x->point < y->point ? -1 : (x->point > y->point);
}
int orc_voxel_grid(const float* pts, int n, float leaf, float* out) {
    if (n <= 0) return 0;
    const float inv = 1.0f / leaf; /* inverse_leaf_size_ = 1 / leaf_size_ (float) */
    float mn[3] = {HUGE_VALF, HUGE_VALF, HUGE_VALF}, mx[3] = {-HUGE_VALF, -HUGE_VALF, -HUGE_VALF};
    int finite = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = pts + 3 * (size_t)i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        ++finite;
        for (int c = 0; c < 3; ++c) mn[c] = p[c] < mn[c] ? p[c] : mn[c], mx[c] = p[c] > mx[c] ? p[c] : mx[c];
    }
    if (!finite) return 0;
    long minb[3], divb[3];
    for (int c = 0; c < 3; ++c) {
        minb[c] = (long)floorf(mn[c] * inv);
        divb[c] = (long)floorf(mx[c] * inv) - minb[c] + 1;
    }
    vg_pair* pr = (vg_pair*)malloc(sizeof(vg_pair) * (size_t)finite);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = pts + 3 * (size_t)i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        const long i0 = (long)floorf(p[0] * inv) - minb[0], i1 = (long)floorf(p[1] * inv) - minb[1],
                   i2 = (long)floorf(p[2] * inv) - minb[2];
        pr[m].cell = i0 + i1 * divb[0] + i2 * divb[0] * divb[1], pr[m].point = i, ++m;
    }
    qsort(pr, (size_t)m, sizeof(vg_pair), vg_cmp);
    int nout = 0;
    for (int i = 0; i < m;) {
        int j = i;
        float s[3] = {0.f, 0.f, 0.f};
        for (; j < m && pr[j].cell == pr[i].cell; ++j)
            for (int c = 0; c < 3; ++c) s[c] += pts[3 * (size_t)pr[j].point + c];
        const float cnt = (float)(j - i);
        for (int c = 0; c < 3; ++c) out[3 * (size_t)nout + c] = s[c] / cnt;
        ++nout, i = j;
    }
    free(pr);
    return nout;
}


/* projective association of n vertices into a live vertex / normal map: the gates of ComputeIcpHelper::find_coresp
 * (proj_icp.cu:72-98).  Outputs NaN / -1 where there is no association; normals, nmap, out_n may be NULL. */
void orc_correspond_projective(const float* verts, const float* normals, int n, const float* vmap, int vmap_step,
                               const float* nmap, int nmap_step, int cols, int rows, float fx, float fy, float cx, float cy,
                               float dist_thresh, float min_cosine, float* out_v, float* out_n, int32_t* out_pixel) {
    const float dist2 = dist_thresh * dist_thresh;
    union { uint32_t u; float f; } q;
    q.u = 0x7fc00000u;
    for (int i = 0; i < n; ++i) {
        const float sx = verts[3 * i], sy = verts[3 * i + 1], sz = verts[3 * i + 2];
        float d[3] = {q.f, q.f, q.f}, nd[3] = {q.f, q.f, q.f};
        int pix = -1;
        if (sz > 0.f) {
            const float u = fmaf(fx, sx / sz, cx), w = fmaf(fy, sy / sz, cy);
            if (u >= 0.f && w >= 0.f && u < (float)cols && w < (float)rows) {
                const int iu = (int)floorf(u), iw = (int)floorf(w);
                const float* v = (const float*)((const char*)vmap + (size_t)iw * vmap_step) + 4 * iu;
                int ok = v[0] == v[0];
                const float ex = sx - v[0], ey = sy - v[1], ez = sz - v[2];
                ok = ok && !(fmaf(ez, ez, fmaf(ey, ey, ex * ex)) > dist2); /* the device's dot(): two fused steps */
                float nn[3] = {q.f, q.f, q.f};
                if (ok && nmap) {
                    const float* nv = (const float*)((const char*)nmap + (size_t)iw * nmap_step) + 4 * iu;
                    nn[0] = nv[0], nn[1] = nv[1], nn[2] = nv[2];
                    ok = nv[0] == nv[0];
                    if (ok && normals) {
                        const float dt = fmaf(normals[3 * i + 2], nn[2], fmaf(normals[3 * i + 1], nn[1], normals[3 * i] * nn[0]));
                        ok = !(fabsf(dt) < min_cosine);
                    }
                }
                if (ok) {
                    d[0] = v[0], d[1] = v[1], d[2] = v[2];
                    nd[0] = nn[0], nd[1] = nn[1], nd[2] = nn[2];
                    pix = iw * cols + iu;
                }
            }
        }
        if (out_v) out_v[3 * i] = d[0], out_v[3 * i + 1] = d[1], out_v[3 * i + 2] = d[2];
        if (out_n) out_n[3 * i] = nd[0], out_n[3 * i + 1] = nd[1], out_n[3 * i + 2] = nd[2];
        if (out_pixel) out_pixel[i] = pix;
    }
}
