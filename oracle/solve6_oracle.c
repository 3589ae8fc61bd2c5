/* solve6_oracle.c — CPU statement of the NORTH-STAR solve (6-DoF node twists, dual-quaternion blend,
 * projective point-to-plane data term, ARAP-style regulariser, block-Jacobi PCG).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: this mode is not in the reference's
 * code — BASELINE.json:north_star and SURVEY.md App. B.2 describe it, DESIGN.md §4.5 fixes the
 * formulas; there is no reference implementation, test or golden vector to pin it on (Ceres is
 * not called anywhere in the reference tree).  What pins this file instead: finite-difference
 * checks of the Jacobians, a dense least-squares solve of the same linearisation, and ground-truth
 * recovery on synthetic motion (tests/test_oracle_solve6.py).
 *
 * The only reference code of this family, followed where it applies:
 *   computePointNormals  src/kfusion/cuda/imgproc.cu:187-215 (+ Reprojector device.hpp:50-54)
 *   point-to-plane row   src/kfusion/cuda/proj_icp.cu:343-350  [s x n, n | n.(d - s)]
 *   projective lookup    src/kfusion/cuda/proj_icp.cu:72-98    (round to nearest pixel, gates)
 *   RBF weight           src/dynfu/utils/node.cpp:29-36
 *   Tukey / Huber        src/dynfu/utils/opt_solver.cpp:204-268
 * All arithmetic of the solve is double precision (the HIP path is fp32 with double cost sums;
 * tests state the tolerance). */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ------------------------------------------------------------------ quaternion helpers (w,x,y,z) */
static void qmul(const double a[4], const double b[4], double o[4]) {
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    const double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    o[0] = w, o[1] = x, o[2] = y, o[3] = z;
}
static void qconj(const double a[4], double o[4]) { o[0] = a[0], o[1] = -a[1], o[2] = -a[2], o[3] = -a[3]; }
/* vec(a (0,c) conj(a)) */
static void qsandwich(const double a[4], const double c[3], double o[3]) {
    const double cq[4] = {0, c[0], c[1], c[2]};
    double t[4], ac[4], r[4];
    qmul(a, cq, t);
    qconj(a, ac);
    qmul(t, ac, r);
    o[0] = r[1], o[1] = r[2], o[2] = r[3];
}
/* rotation and translation of a unit dual quaternion: R c = vec(r c r*), t = 2 vec(d r*) */
static void dq_apply(const double q[8], const double c[3], double o[3]) {
    double rc[4], dr[4];
    qsandwich(q, c, o);
    qconj(q, rc);
    qmul(q + 4, rc, dr);
    o[0] += 2 * dr[1], o[1] += 2 * dr[2], o[2] += 2 * dr[3];
}
static void dq_load(const float* f, double q[8]) {
    for (int i = 0; i < 8; ++i) q[i] = f[i];
}

/* ------------------------------------------------------------ computePointNormals (imgproc.cu) */
static void reproj(int u, int v, float z, float finvx, float finvy, float cx, float cy, float o[3]) { /* device.hpp:50-54 */
    o[0] = z * ((float)u - cx) * finvx;
    o[1] = z * ((float)v - cy) * finvy;
    o[2] = z;
}
void orc6_points_normals(const uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                         float cy, float* points, int points_step, float* normals, int normals_step) {
    const float finvx = 1.f / fx, finvy = 1.f / fy; /* Reprojector ctor, precomp.cpp */
    const float qnan = NAN;
    for (int y = 0; y < rows; ++y) {
        const uint16_t* d0 = (const uint16_t*)((const char*)depth + (size_t)y * depth_step);
        const uint16_t* d1 = (const uint16_t*)((const char*)depth + (size_t)(y + 1) * depth_step);
        float* P = (float*)((char*)points + (size_t)y * points_step);
        float* Nn = (float*)((char*)normals + (size_t)y * normals_step);
        for (int x = 0; x < cols; ++x) {
            for (int c = 0; c < 4; ++c) P[4 * x + c] = Nn[4 * x + c] = qnan; /* :195-196 */
            if (x >= cols - 1 || y >= rows - 1) continue;                    /* :198 */
            const float z00 = d0[x] * 0.001f, z01 = d0[x + 1] * 0.001f, z10 = d1[x] * 0.001f;
            if (z00 * z01 * z10 != 0) { /* :206 */
                float v00[3], v01[3], v10[3];
                reproj(x, y, z00, finvx, finvy, cx, cy, v00);
                reproj(x + 1, y, z01, finvx, finvy, cx, cy, v01);
                reproj(x, y + 1, z10, finvx, finvy, cx, cy, v10);
                const float a[3] = {v01[0] - v00[0], v01[1] - v00[1], v01[2] - v00[2]};
                const float b[3] = {v10[0] - v00[0], v10[1] - v00[1], v10[2] - v00[2]};
                float n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
                /* normalized(): v * rsqrt(dot(v,v)), dot = fma chain as device.hpp / temp_utils.hpp */
                const float dd = fmaf(n[2], n[2], fmaf(n[1], n[1], n[0] * n[0]));
                const float inv = 1.0f / sqrtf(dd);
                n[0] *= inv, n[1] *= inv, n[2] *= inv;
                Nn[4 * x] = -n[0], Nn[4 * x + 1] = -n[1], Nn[4 * x + 2] = -n[2], Nn[4 * x + 3] = 0.f; /* :212 */
                P[4 * x] = v00[0], P[4 * x + 1] = v00[1], P[4 * x + 2] = v00[2], P[4 * x + 3] = 0.f;  /* :213 */
            }
        }
    }
}

/* ------------------------------------------------------------------------------- graphs */
/* data graph: k nearest nodes and NORMALISED radial basis weights (fp32, slot order);
 * regularisation graph: the k nearest OTHER nodes of every node (-1 padded) */
void orc6_graph(const float* node_pos, const float* node_w, int D, int k, const float* canon, int N, int32_t* idx,
                float* wn, int32_t* reg_idx, int threads) {
    orc_knn(node_pos, D, canon, N, k, idx, threads);
    for (int v = 0; v < N; ++v) {
        float w[16], sum = 0.f;
        for (int j = 0; j < k; ++j) {
            const int n = idx[(size_t)v * k + j];
            w[j]        = n >= 0 ? orc_transformation_weight(node_pos + 3 * n, node_w[n], canon + 3 * (size_t)v) : 0.f;
            sum += w[j];
        }
        for (int j = 0; j < k; ++j) wn[(size_t)v * k + j] = sum > 0.f ? w[j] / sum : 0.f;
    }
    if (reg_idx) {
        int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)D * (k + 1));
        orc_knn(node_pos, D, node_pos, D, k + 1 > 16 ? 16 : k + 1, tmp, threads);
        const int kk = k + 1 > 16 ? 16 : k + 1;
        for (int n = 0; n < D; ++n) {
            int o = 0;
            for (int j = 0; j < kk && o < k; ++j) {
                const int m = tmp[(size_t)n * kk + j];
                if (m >= 0 && m != n) reg_idx[(size_t)n * k + o++] = m;
            }
            for (; o < k; ++o) reg_idx[(size_t)n * k + o] = -1;
        }
        free(tmp);
    }
}

/* ------------------------------------------------------------------------ DQB of one vertex */
typedef struct {
    double a[4], b[4], m; /* blended (un-normalised) real / dual parts, |a|^2 */
    double s[16];         /* hemisphere signs of the neighbours */
} blend_t;

static void blend(const float* node_dq, const int32_t* idx, const float* wn, int k, blend_t* B) {
    memset(B, 0, sizeof(*B));
    double r0[4] = {1, 0, 0, 0};
    int have = 0;
    for (int j = 0; j < k; ++j) {
        const int n = idx[j];
        if (n < 0 || wn[j] == 0.f) {
            B->s[j] = 0;
            continue;
        }
        double q[8];
        dq_load(node_dq + 8 * (size_t)n, q);
        if (!have) memcpy(r0, q, sizeof(r0)), have = 1;
        const double dot = q[0] * r0[0] + q[1] * r0[1] + q[2] * r0[2] + q[3] * r0[3];
        const double s   = dot < 0 ? -1.0 : 1.0;
        B->s[j]          = s;
        for (int c = 0; c < 4; ++c) B->a[c] += (double)wn[j] * s * q[c], B->b[c] += (double)wn[j] * s * q[4 + c];
    }
    B->m = B->a[0] * B->a[0] + B->a[1] * B->a[1] + B->a[2] * B->a[2] + B->a[3] * B->a[3];
}
/* p = (vec(a c a*) + 2 vec(b a*)) / |a|^2 ; returns 0 if the vertex has no support */
static int blend_point(const blend_t* B, const double c[3], double p[3]) {
    if (!(B->m > 0)) return 0;
    double ac[4], ba[4];
    qsandwich(B->a, c, p);
    qconj(B->a, ac);
    qmul(B->b, ac, ba);
    for (int i = 0; i < 3; ++i) p[i] = (p[i] + 2 * ba[1 + i]) / B->m;
    return 1;
}
static void blend_normal(const blend_t* B, const double n[3], double o[3]) {
    qsandwich(B->a, n, o);
    for (int i = 0; i < 3; ++i) o[i] /= B->m;
}

void orc6_warp(const float* node_dq, int k, const int32_t* idx, const float* wn, const float* canon,
               const float* canon_n, int N, float* out_p, float* out_n) {
    for (int v = 0; v < N; ++v) {
        blend_t B;
        blend(node_dq, idx + (size_t)v * k, wn + (size_t)v * k, k, &B);
        const double c[3] = {canon[3 * (size_t)v], canon[3 * (size_t)v + 1], canon[3 * (size_t)v + 2]};
        double p[3] = {c[0], c[1], c[2]}, nn[3] = {0, 0, 0};
        const int ok = blend_point(&B, c, p);
        for (int i = 0; i < 3; ++i) out_p[3 * (size_t)v + i] = (float)p[i];
        if (canon_n && out_n) {
            const double n[3] = {canon_n[3 * (size_t)v], canon_n[3 * (size_t)v + 1], canon_n[3 * (size_t)v + 2]};
            if (ok) blend_normal(&B, n, nn);
            else memcpy(nn, n, sizeof(nn));
            for (int i = 0; i < 3; ++i) out_n[3 * (size_t)v + i] = (float)nn[i];
        }
    }
}

/* current position of node i: g^ = T_i(g_i) */
static void node_now(const float* node_pos, const float* node_dq, int i, double o[3]) {
    double q[8];
    dq_load(node_dq + 8 * (size_t)i, q);
    const double g[3] = {node_pos[3 * i], node_pos[3 * i + 1], node_pos[3 * i + 2]};
    dq_apply(q, g, o);
}

/* d p / d xi_j for the k neighbours: J[j][col][row] (col = twist component, row = x,y,z).
 * Twist of node i = (omega, v) about the node's current position g^_i:
 * delta y = omega x (y - g^_i) + v. */
static void data_jacobian(const float* node_pos, const float* node_dq, const int32_t* idx, const float* wn, int k,
                          const blend_t* B, const double c[3], const double p[3], double J[16][6][3]) {
    const double cq[4] = {0, c[0], c[1], c[2]};
    double ac[4];
    qconj(B->a, ac);
    for (int j = 0; j < k; ++j) {
        const int n = idx[j];
        if (n < 0 || B->s[j] == 0) {
            memset(J[j], 0, sizeof(J[j]));
            continue;
        }
        double q[8], gh[3];
        dq_load(node_dq + 8 * (size_t)n, q);
        node_now(node_pos, node_dq, n, gh);
        const double ws = (double)wn[j] * B->s[j];
        for (int col = 0; col < 6; ++col) {
            double om[3] = {0, 0, 0}, v0[3] = {0, 0, 0};
            if (col < 3) {
                om[col] = 1; /* origin form of a rotation about g^: v0 = -omega x g^ */
                v0[0] = -(om[1] * gh[2] - om[2] * gh[1]);
                v0[1] = -(om[2] * gh[0] - om[0] * gh[2]);
                v0[2] = -(om[0] * gh[1] - om[1] * gh[0]);
            } else {
                v0[col - 3] = 1;
            }
            const double oq[4] = {0, om[0], om[1], om[2]}, vq[4] = {0, v0[0], v0[1], v0[2]};
            double da[4], db[4], t1[4], t2[4];
            qmul(oq, q, da); /* delta r = 1/2 omega^ r */
            qmul(oq, q + 4, t1);
            qmul(vq, q, t2); /* delta d = 1/2 (omega^ d + v^ r) */
            for (int i = 0; i < 4; ++i) da[i] *= 0.5 * ws, db[i] = 0.5 * ws * (t1[i] + t2[i]);
            /* delta u = vec(da c a* + a c da*) + 2 vec(db a* + b da*) */
            double dac[4], u1[4], u2[4], u3[4], u4[4], tmp[4];
            qconj(da, dac);
            qmul(da, cq, tmp), qmul(tmp, ac, u1);
            qmul(B->a, cq, tmp), qmul(tmp, dac, u2);
            qmul(db, ac, u3);
            qmul(B->b, dac, u4);
            const double dm = 2 * (B->a[0] * da[0] + B->a[1] * da[1] + B->a[2] * da[2] + B->a[3] * da[3]);
            for (int i = 0; i < 3; ++i)
                J[j][col][i] = (u1[1 + i] + u2[1 + i] + 2 * (u3[1 + i] + u4[1 + i]) - p[i] * dm) / B->m;
        }
    }
}

void orc6_data_jacobian(const float* node_pos, const float* node_dq, const int32_t* idx, const float* wn, int k,
                        const float c[3], double* J /* k x 6 x 3 */, double p_out[3]) {
    blend_t B;
    blend(node_dq, idx, wn, k, &B);
    const double cd[3] = {c[0], c[1], c[2]};
    double p[3] = {cd[0], cd[1], cd[2]}, Jt[16][6][3];
    memset(Jt, 0, sizeof(Jt));
    if (blend_point(&B, cd, p)) data_jacobian(node_pos, node_dq, idx, wn, k, &B, cd, p, Jt);
    memcpy(J, Jt, sizeof(double) * (size_t)k * 18);
    memcpy(p_out, p, sizeof(double) * 3);
}

/* T_i <- twist about g^_i applied on the left: R' = Exp(omega) R, t' = Exp(omega)(t - g^) + g^ + v */
void orc6_apply_twist(const float node_pos_i[3], const float dq_in[8], const double twist[6], float dq_out[8]) {
    double q[8], gh[3];
    dq_load(dq_in, q);
    const double g[3] = {node_pos_i[0], node_pos_i[1], node_pos_i[2]};
    dq_apply(q, g, gh);
    const double th = sqrt(twist[0] * twist[0] + twist[1] * twist[1] + twist[2] * twist[2]);
    const double sc = th > 1e-12 ? sin(0.5 * th) / th : 0.5;
    const double qo[4] = {cos(0.5 * th), sc * twist[0], sc * twist[1], sc * twist[2]};
    double rn[4], rc[4], dr[4], t[3], tc[3], tr[3];
    qmul(qo, q, rn);
    qconj(q, rc);
    qmul(q + 4, rc, dr);
    for (int i = 0; i < 3; ++i) t[i] = 2 * dr[1 + i], tc[i] = t[i] - gh[i];
    qsandwich(qo, tc, tr);
    for (int i = 0; i < 3; ++i) t[i] = tr[i] + gh[i] + twist[3 + i];
    const double nr = sqrt(rn[0] * rn[0] + rn[1] * rn[1] + rn[2] * rn[2] + rn[3] * rn[3]);
    for (int i = 0; i < 4; ++i) rn[i] /= nr;
    const double tq[4] = {0, t[0], t[1], t[2]};
    double dn[4];
    qmul(tq, rn, dn);
    for (int i = 0; i < 4; ++i) dq_out[i] = (float)rn[i], dq_out[4 + i] = (float)(0.5 * dn[i]);
}

/* --------------------------------------------------------------------------- the solve */
typedef struct {
    int ncol;
    int* col;
    double* blk; /* ncol x 36, row-major 6x6 */
} brow_t;

static int brow_find(brow_t* r, int c, int create) {
    for (int i = 0; i < r->ncol; ++i)
        if (r->col[i] == c) return i;
    if (!create) return -1;
    r->col = (int*)realloc(r->col, sizeof(int) * (size_t)(r->ncol + 1));
    r->blk = (double*)realloc(r->blk, sizeof(double) * 36 * (size_t)(r->ncol + 1));
    r->col[r->ncol] = c;
    memset(r->blk + 36 * (size_t)r->ncol, 0, sizeof(double) * 36);
    return r->ncol++;
}

/* solve the 6x6 SPD system M x = b by Cholesky; returns 0 if not positive definite */
static int chol6(const double* M, const double* b, double* x) {
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = M[6 * i + j];
            for (int q = 0; q < j; ++q) s -= L[6 * i + q] * L[6 * j + q];
            if (i == j) {
                if (!(s > 0)) return 0;
                L[6 * i + i] = sqrt(s);
            } else {
                L[6 * i + j] = s / L[6 * j + j];
            }
        }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int q = 0; q < i; ++q) s -= L[6 * i + q] * y[q];
        y[i] = s / L[6 * i + i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int q = i + 1; q < 6; ++q) s -= L[6 * q + i] * x[q];
        x[i] = s / L[6 * i + i];
    }
    return 1;
}

static double tukey(double err, double offset, double c) { /* opt_solver.cpp:204-231 */
    const double e = err / offset;
    if (e < c) {
        const double t = 1 - (e * e) / (c * c);
        return t * t;
    }
    return 0;
}
static double huber(double e, double kk) { return e <= kk ? 1.0 : kk / e; } /* opt_solver.cpp:233-268 */

/* development hook (tools/pcg6_experiments.py): the normal equations of Gauss-Newton iteration `gn` of the next
 * orc6_solve call are copied out as block CSR (row_ptr D + 1, cols / 36 doubles per block up to `cap` blocks, g 6 D) */
static struct {
    int gn, *row_ptr, *cols;
    long cap, *nblk;
    double *blk, *g;
} s_dump = {-1, 0, 0, 0, 0, 0, 0};
void orc6_set_dump(int gn, int* row_ptr, int* cols, double* blk, double* g, long cap, long* nblk) {
    s_dump.gn = gn, s_dump.row_ptr = row_ptr, s_dump.cols = cols, s_dump.blk = blk, s_dump.g = g, s_dump.cap = cap,
    s_dump.nblk = nblk;
}

void orc6_solve(const float* node_pos, const float* node_dq_in, const float* node_w, int D, int k, const float* canon,
                const float* canon_n, int N, const float* vmap, int vmap_step, const float* nmap, int nmap_step, int cols,
                int rows, float fx, float fy, float cx, float cy, const orc6_params* prm, float* node_dq_out,
                orc6_stats* stats) {
    const int threads = prm->threads > 0 ? prm->threads : 1;
    int32_t* idx  = (int32_t*)malloc(sizeof(int32_t) * (size_t)N * k);
    float* wn     = (float*)malloc(sizeof(float) * (size_t)N * k);
    int32_t* ridx = (int32_t*)malloc(sizeof(int32_t) * (size_t)D * k);
    orc6_graph(node_pos, node_w, D, k, canon, N, idx, wn, ridx, threads);
    float* dq = (float*)malloc(sizeof(float) * 8 * (size_t)D);
    memcpy(dq, node_dq_in, sizeof(float) * 8 * (size_t)D);

    /* per data row: residual, 6-vectors per slot, weight */
    double* res  = (double*)calloc((size_t)N, sizeof(double));
    double* rho  = (double*)calloc((size_t)N, sizeof(double));
    double* avec = (double*)calloc((size_t)N * k * 6, sizeof(double));
    unsigned char* valid = (unsigned char*)calloc((size_t)N, 1);
    /* per reg edge (n, slot): residual e[3], vectors of node n (3 rows x 6), huber */
    double* rres = (double*)calloc((size_t)D * k * 3, sizeof(double));
    double* rvec = (double*)calloc((size_t)D * k * 18, sizeof(double));
    double* rhub = (double*)calloc((size_t)D * k, sizeof(double));
    const double wreg2 = (double)prm->lambda / ((double)D * (double)k); /* opt_solver.cpp:30 */

    brow_t* H  = (brow_t*)calloc((size_t)D, sizeof(brow_t));
    double* g  = (double*)malloc(sizeof(double) * 6 * (size_t)D);
    double* x  = (double*)malloc(sizeof(double) * 6 * (size_t)D);
    double* r  = (double*)malloc(sizeof(double) * 6 * (size_t)D);
    double* z  = (double*)malloc(sizeof(double) * 6 * (size_t)D);
    double* pp = (double*)malloc(sizeof(double) * 6 * (size_t)D);
    double* qq = (double*)malloc(sizeof(double) * 6 * (size_t)D);
    /* sparsity (fixed by the graphs) */
    for (int v = 0; v < N; ++v)
        for (int s = 0; s < k; ++s)
            for (int j = 0; j < k; ++j) {
                const int a = idx[(size_t)v * k + s], b = idx[(size_t)v * k + j];
                if (a >= 0 && b >= 0 && wn[(size_t)v * k + s] != 0.f && wn[(size_t)v * k + j] != 0.f) brow_find(&H[a], b, 1);
            }
    for (int n = 0; n < D; ++n) {
        brow_find(&H[n], n, 1);
        for (int s = 0; s < k; ++s) {
            const int m = ridx[(size_t)n * k + s];
            if (m >= 0) brow_find(&H[n], m, 1), brow_find(&H[m], n, 1), brow_find(&H[m], m, 1);
        }
    }

    memset(stats, 0, sizeof(*stats));
    int first = 1;
    double rz0_prev = 0;
    /* Gauss-Newton control (oracle.h: gn_tol): E_ref / the transforms before the last step / whether the outer iteration
     * has ended.  Slot `total` of the loop is the closing check of the last step (gn_tol > 0 only). */
    const int early = prm->gn_tol > 0, total = prm->num_iter * prm->gn_iter;
    int stopped = 0;
    double cost_ref = 0;
    long valid_ref  = 0;
    float* dq_prev  = (float*)malloc(sizeof(float) * 8 * (size_t)D);
    memcpy(dq_prev, dq, sizeof(float) * 8 * (size_t)D);
    {
        for (int gi = 0; gi < total + early; ++gi) {
            const int closing = gi == total;
            const int gn      = closing ? prm->gn_iter : gi % prm->gn_iter; /* (closing: behaves as an iteration > 0) */
            const int hist    = gi < ORC6_HIST ? gi : -1;
            if (gn == 0) stopped = 0;
            if (closing && (stopped || total == 0)) break;
            if (hist >= 0) stats->hist_n = hist + 1;
            if (stopped) {
                if (hist >= 0) stats->stop_hist[hist] = 3;
                continue;
            }
            const int update_w = gn == 0;
            /* ---- linearise: data rows */
            double cost = 0;
            long nvalid = 0;
#pragma omp parallel for schedule(static) num_threads(threads) reduction(+ : cost, nvalid)
            for (int v = 0; v < N; ++v) {
                const int32_t* iv = idx + (size_t)v * k;
                const float* wv   = wn + (size_t)v * k;
                double* av        = avec + (size_t)v * k * 6;
                valid[v] = 0, res[v] = 0;
                memset(av, 0, sizeof(double) * (size_t)k * 6);
                blend_t B;
                blend(dq, iv, wv, k, &B);
                const double c[3] = {canon[3 * (size_t)v], canon[3 * (size_t)v + 1], canon[3 * (size_t)v + 2]};
                double p[3];
                if (!blend_point(&B, c, p)) continue;
                if (!(p[2] > 0)) continue;
                /* projective association: nearest pixel (proj_icp.cu:80-86 rounds with __float2int_rn) */
                const double uf = (double)fx * (p[0] / p[2]) + cx, vf = (double)fy * (p[1] / p[2]) + cy;
                const long u = lrint(uf), w = lrint(vf);
                if (u < 0 || w < 0 || u >= cols || w >= rows) continue;
                const float* L  = (const float*)((const char*)vmap + (size_t)w * vmap_step) + 4 * u;
                const float* Ln = (const float*)((const char*)nmap + (size_t)w * nmap_step) + 4 * u;
                if (isnan(L[0]) || isnan(Ln[0])) continue;
                const double dl[3] = {p[0] - L[0], p[1] - L[1], p[2] - L[2]};
                const double dist  = sqrt(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]);
                if (dist > prm->dist_thresh) continue;
                if (canon_n) {
                    const double n0[3] = {canon_n[3 * (size_t)v], canon_n[3 * (size_t)v + 1], canon_n[3 * (size_t)v + 2]};
                    double nw[3];
                    blend_normal(&B, n0, nw);
                    if (nw[0] * Ln[0] + nw[1] * Ln[1] + nw[2] * Ln[2] < prm->cos_thresh) continue;
                }
                const double rr = Ln[0] * dl[0] + Ln[1] * dl[1] + Ln[2] * dl[2]; /* n . (p - l) */
                if (update_w) rho[v] = tukey(fabs(rr), prm->tukey_offset, prm->psi_data);
                double J[16][6][3];
                data_jacobian(node_pos, dq, iv, wv, k, &B, c, p, J);
                for (int j = 0; j < k; ++j)
                    for (int col = 0; col < 6; ++col)
                        av[j * 6 + col] = Ln[0] * J[j][col][0] + Ln[1] * J[j][col][1] + Ln[2] * J[j][col][2];
                valid[v] = 1, res[v] = rr;
                cost += rho[v] * rr * rr;
                nvalid += rho[v] > 0;
            }
            /* ---- linearise: regularisation rows */
            double rcost = 0;
            for (int n = 0; n < D; ++n) {
                double qn[8], ghn[3];
                dq_load(dq + 8 * (size_t)n, qn);
                node_now(node_pos, dq, n, ghn);
                for (int s = 0; s < k; ++s) {
                    const int m   = ridx[(size_t)n * k + s];
                    double* e     = rres + ((size_t)n * k + s) * 3;
                    double* vec   = rvec + ((size_t)n * k + s) * 18;
                    memset(e, 0, sizeof(double) * 3), memset(vec, 0, sizeof(double) * 18);
                    if (m < 0) continue;
                    const double gm[3] = {node_pos[3 * m], node_pos[3 * m + 1], node_pos[3 * m + 2]};
                    double y[3], ghm[3];
                    dq_apply(qn, gm, y);
                    node_now(node_pos, dq, m, ghm);
                    for (int c = 0; c < 3; ++c) e[c] = y[c] - ghm[c];
                    const double en = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
                    if (update_w) rhub[(size_t)n * k + s] = huber(en, prm->psi_reg);
                    /* d(y - g^_m)/d xi_n = [ -[y - g^_n]x | I ];  d/d xi_m = [ 0 | -I ] */
                    const double l[3] = {y[0] - ghn[0], y[1] - ghn[1], y[2] - ghn[2]};
                    /* row c of -[l]x : (-[l]x w)_c = (w x l)_c = -(l x w)_c */
                    const double S[3][3] = {{0, l[2], -l[1]}, {-l[2], 0, l[0]}, {l[1], -l[0], 0}};
                    for (int c = 0; c < 3; ++c) {
                        for (int q = 0; q < 3; ++q) vec[c * 6 + q] = S[c][q];
                        vec[c * 6 + 3 + c] = 1;
                    }
                    rcost += wreg2 * rhub[(size_t)n * k + s] * en * en;
                }
            }
            cost += rcost;
            if (first) stats->initial_cost = cost, stats->valid_first = nvalid, first = 0;
            if (hist >= 0)
                stats->cost_hist[hist] = cost, stats->valid_hist[hist] = nvalid, stats->pcg_it_hist[hist] = 0,
                stats->pcg_rel_hist[hist] = 1.0;
            ++stats->gn_iters;
            if (early && gn > 0) {
                if (cost > (1.0 + (double)prm->gn_tol) * cost_ref) { /* the step raised the energy: undo it */
                    memcpy(dq, dq_prev, sizeof(float) * 8 * (size_t)D);
                    stopped = 1, ++stats->gn_rejected;
                    if (hist >= 0) stats->stop_hist[hist] = 2;
                    stats->final_cost = cost_ref, stats->valid_last = valid_ref;
                    continue;
                }
                if (closing || cost_ref - cost <= (double)prm->gn_tol * cost_ref) { /* converged (closing: the step is kept) */
                    stopped = 1, stats->gn_converged += !closing;
                    if (hist >= 0) stats->stop_hist[hist] = 1;
                    stats->final_cost = cost, stats->valid_last = nvalid;
                    continue;
                }
            }
            cost_ref = cost, valid_ref = nvalid;
            stats->final_cost = cost, stats->valid_last = nvalid; /* energy at the last accepted linearisation */
            ++stats->gn_solves;
            const int build_H = !(prm->reuse_matrix && gn > 0); /* reuse_matrix: H, M^-1 of iteration 0 stay */

            /* (no PCG iterations asked for — orc6_cost: the energy is all that is wanted — no normal equations either) */
            if (prm->linear_iter <= 0) memset(x, 0, sizeof(double) * 6 * (size_t)D);
            else {
            /* ---- assemble H, g */
            for (int n = 0; n < D; ++n) {
                if (build_H) memset(H[n].blk, 0, sizeof(double) * 36 * (size_t)H[n].ncol);
                memset(g + 6 * (size_t)n, 0, sizeof(double) * 6);
            }
            for (int v = 0; v < N; ++v) {
                if (!valid[v] || rho[v] == 0) continue;
                const double* av = avec + (size_t)v * k * 6;
                for (int s = 0; s < k; ++s) {
                    const int a = idx[(size_t)v * k + s];
                    if (a < 0 || wn[(size_t)v * k + s] == 0.f) continue;
                    for (int c = 0; c < 6; ++c) g[6 * (size_t)a + c] -= rho[v] * av[s * 6 + c] * res[v];
                    for (int j = 0; j < k && build_H; ++j) {
                        const int b = idx[(size_t)v * k + j];
                        if (b < 0 || wn[(size_t)v * k + j] == 0.f) continue;
                        double* blk = H[a].blk + 36 * (size_t)brow_find(&H[a], b, 0);
                        for (int c = 0; c < 6; ++c)
                            for (int d2 = 0; d2 < 6; ++d2) blk[6 * c + d2] += rho[v] * av[s * 6 + c] * av[j * 6 + d2];
                    }
                }
            }
            for (int n = 0; n < D; ++n)
                for (int s = 0; s < k; ++s) {
                    const int m = ridx[(size_t)n * k + s];
                    if (m < 0) continue;
                    const double wt   = wreg2 * rhub[(size_t)n * k + s];
                    const double* e   = rres + ((size_t)n * k + s) * 3;
                    const double* vec = rvec + ((size_t)n * k + s) * 18;
                    double* Hnn = H[n].blk + 36 * (size_t)brow_find(&H[n], n, 0);
                    double* Hnm = H[n].blk + 36 * (size_t)brow_find(&H[n], m, 0);
                    double* Hmn = H[m].blk + 36 * (size_t)brow_find(&H[m], n, 0);
                    double* Hmm = H[m].blk + 36 * (size_t)brow_find(&H[m], m, 0);
                    for (int c = 0; c < 3; ++c) {
                        const double* an = vec + c * 6; /* node n's 6-vector of row c; node m's is -e_{3+c} */
                        for (int q = 0; q < 6; ++q) {
                            g[6 * (size_t)n + q] -= wt * an[q] * e[c];
                            if (!build_H) continue;
                            for (int q2 = 0; q2 < 6; ++q2) Hnn[6 * q + q2] += wt * an[q] * an[q2];
                            Hnm[6 * q + 3 + c] -= wt * an[q];
                            Hmn[6 * (3 + c) + q] -= wt * an[q];
                        }
                        g[6 * (size_t)m + 3 + c] += wt * e[c];
                        if (build_H) Hmm[6 * (3 + c) + 3 + c] += wt;
                    }
                }
            for (int n = 0; n < D && build_H; ++n) {
                double* Hd = H[n].blk + 36 * (size_t)brow_find(&H[n], n, 0);
                for (int c = 0; c < 6; ++c) Hd[7 * c] += prm->damping;
            }

            if (s_dump.gn == gi && s_dump.row_ptr) {
                long nb = 0;
                for (int n = 0; n < D; ++n) {
                    s_dump.row_ptr[n] = (int)nb;
                    for (int e = 0; e < H[n].ncol && nb < s_dump.cap; ++e, ++nb) {
                        s_dump.cols[nb] = H[n].col[e];
                        memcpy(s_dump.blk + 36 * nb, H[n].blk + 36 * (size_t)e, sizeof(double) * 36);
                    }
                }
                s_dump.row_ptr[D] = (int)nb;
                *s_dump.nblk      = nb;
                memcpy(s_dump.g, g, sizeof(double) * 6 * (size_t)D);
                s_dump.gn = -1;
            }
            /* ---- block-Jacobi PCG: H x = g */
            memset(x, 0, sizeof(double) * 6 * (size_t)D);
            memcpy(r, g, sizeof(double) * 6 * (size_t)D);
            double rz = 0;
            for (int n = 0; n < D; ++n) {
                const double* Hd = H[n].blk + 36 * (size_t)brow_find(&H[n], n, 0);
                if (!chol6(Hd, r + 6 * (size_t)n, z + 6 * (size_t)n)) memset(z + 6 * (size_t)n, 0, sizeof(double) * 6);
                for (int c = 0; c < 6; ++c) rz += r[6 * (size_t)n + c] * z[6 * (size_t)n + c];
            }
            memcpy(pp, z, sizeof(double) * 6 * (size_t)D);
            const double rz0 = rz;
            double tol = prm->pcg_tol;
            if (prm->pcg_tol_first > 0 && prm->pcg_tol_adapt > 0) { /* Eisenstat-Walker */
                double eta = prm->pcg_tol_first;
                if (gn > 0 && rz0_prev > 0) eta = (double)prm->pcg_tol_adapt * rz0 / rz0_prev;
                if (eta > prm->pcg_tol_first) eta = prm->pcg_tol_first;
                if (eta > tol) tol = eta;
            } else if (prm->pcg_tol_first > 0) {
                double eta = prm->pcg_tol_first;
                for (int i = 0; i < gn; ++i) eta *= prm->pcg_tol_decay;
                if (eta > tol) tol = eta;
            }
            rz0_prev = rz0;
            if (hist >= 0) stats->pcg_tol_hist[hist] = tol;
            if (hist >= 0 && !(rz0 > 0)) stats->pcg_rel_hist[hist] = 0.0;
            for (int it = 0; it < prm->linear_iter && rz > 0; ++it) {
                double pq = 0;
#pragma omp parallel for schedule(static) num_threads(threads) reduction(+ : pq)
                for (int n = 0; n < D; ++n) {
                    double acc[6] = {0, 0, 0, 0, 0, 0};
                    for (int e = 0; e < H[n].ncol; ++e) {
                        const double* blk = H[n].blk + 36 * (size_t)e;
                        const double* pv  = pp + 6 * (size_t)H[n].col[e];
                        for (int c = 0; c < 6; ++c)
                            for (int d2 = 0; d2 < 6; ++d2) acc[c] += blk[6 * c + d2] * pv[d2];
                    }
                    for (int c = 0; c < 6; ++c) qq[6 * (size_t)n + c] = acc[c], pq += acc[c] * pp[6 * (size_t)n + c];
                }
                if (!(pq > 0)) break;
                const double alpha = rz / pq;
                double rz_new      = 0;
                for (int n = 0; n < D; ++n) {
                    for (int c = 0; c < 6; ++c) {
                        x[6 * (size_t)n + c] += alpha * pp[6 * (size_t)n + c];
                        r[6 * (size_t)n + c] -= alpha * qq[6 * (size_t)n + c];
                    }
                    const double* Hd = H[n].blk + 36 * (size_t)brow_find(&H[n], n, 0);
                    if (!chol6(Hd, r + 6 * (size_t)n, z + 6 * (size_t)n)) memset(z + 6 * (size_t)n, 0, sizeof(double) * 6);
                    for (int c = 0; c < 6; ++c) rz_new += r[6 * (size_t)n + c] * z[6 * (size_t)n + c];
                }
                ++stats->pcg_iters;
                const double beta = rz_new / rz;
                rz                = rz_new;
                if (hist >= 0) stats->pcg_it_hist[hist] += 1, stats->pcg_rel_hist[hist] = sqrt(rz > 0 ? rz / rz0 : 0.0);
                for (size_t i = 0; i < 6 * (size_t)D; ++i) pp[i] = z[i] + beta * pp[i];
                if (rz <= tol * tol * rz0) break;
            }
            }
            /* ---- update */
            memcpy(dq_prev, dq, sizeof(float) * 8 * (size_t)D);
            for (int n = 0; n < D; ++n) {
                float o[8];
                orc6_apply_twist(node_pos + 3 * n, dq + 8 * (size_t)n, x + 6 * (size_t)n, o);
                memcpy(dq + 8 * (size_t)n, o, sizeof(o));
            }
        }
    }
    free(dq_prev);
    memcpy(node_dq_out, dq, sizeof(float) * 8 * (size_t)D);
    for (int n = 0; n < D; ++n) free(H[n].col), free(H[n].blk);
    free(H), free(g), free(x), free(r), free(z), free(pp), free(qq);
    free(res), free(rho), free(avec), free(valid), free(rres), free(rvec), free(rhub);
    free(idx), free(wn), free(ridx), free(dq);
}

/* energy at given node transforms (association + weights recomputed; for tests / reports) */
double orc6_cost(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, const float* canon,
                 const float* canon_n, int N, const float* vmap, int vmap_step, const float* nmap, int nmap_step, int cols,
                 int rows, float fx, float fy, float cx, float cy, const orc6_params* prm, long* nvalid_out) {
    orc6_params p = *prm;
    p.num_iter = 1, p.gn_iter = 1, p.linear_iter = 0;
    float* out = (float*)malloc(sizeof(float) * 8 * (size_t)D);
    orc6_stats st;
    orc6_solve(node_pos, node_dq, node_w, D, k, canon, canon_n, N, vmap, vmap_step, nmap, nmap_step, cols, rows, fx, fy,
               cx, cy, &p, out, &st);
    free(out);
    if (nvalid_out) *nvalid_out = st.valid_first;
    return st.initial_cost;
}
