/*
 * solve_oracle.c — CPU restatement of the reference's warp-field solve
 * (src/dynfu/utils/opt_solver.cpp + include/dynfu/utils/terra/energy.t, solved by Opt's
 * Gauss-Newton / PCG in the reference).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Third-party arithmetic that is NOT in the reference tree: Opt (github.com/mbrookes1304/Opt,
 * branch env-variables, no commit pin — Dockerfile:28-33) generates the GN/PCG kernels from
 * energy.t.  Its published algorithm (DeVito et al., "Opt: A Domain Specific Language for
 * Non-linear Least Squares Optimization", 2017, §5: Gauss-Newton with a Jacobi-preconditioned
 * conjugate-gradient inner solve, matrix-free J^T J p over the graph edges) is restated in
 * solve_oracle_body.inc.  Parity is anchored on the reference's own call sites
 * (opt_solver.cpp:15-147) and the 8 OptTest end-state assertions
 * (test/opt_optimisation_test.cpp:212-698, tolerance 1e-3) — tests/test_oracle_solve.py.
 *
 * Outer-iteration contract (SURVEY.md §3.5 "semantics hazard"): the reference's
 * preNonlinearSolve composes the absolute unknown t onto the nodes every outer iteration;
 * the restatement keeps ONE unknown t per solve, re-evaluates the Tukey weights with the
 * nodes' pre-solve transforms composed with the current t, and composes the final t onto
 * the nodes ONCE (node.cpp:19-23).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define REAL float
#define SUF _f32
#include "solve_oracle_body.inc"
#undef REAL
#undef SUF

#define REAL double
#define SUF _f64
#include "solve_oracle_body.inc"
#undef REAL
#undef SUF

/* opt_solver.cpp:204-212 */
static float calc_tukey(float tukey_offset, float c, const float e[3]) {
    /* sqrt(float) / float; pow(float,2) -> double in the reference (<cmath> pow(float,int)) */
    float d = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) / tukey_offset;
    if (d < c) {
        double q = 1.0 - ((double)d * (double)d) / ((double)c * (double)c);
        return (float)(q * q);
    }
    return 0.f;
}

/* opt_solver.cpp:214-231 */
void orc_tukey_weights(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                       const float* canon, const float* live, int N, float tukey_offset, float psi_data, float* tukey,
                       int threads) {
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int i = 0; i < N; ++i) {
        float dq[8], warped[3], e[3];
        orc_calc_dqb(node_pos, node_dq, node_w, D, k, canon + 3 * (size_t)i, dq);
        orc_dq_transform_vertex(dq, canon + 3 * (size_t)i, warped);
        for (int c = 0; c < 3; ++c) e[c] = live[3 * (size_t)i + c] - warped[c];
        tukey[i] = calc_tukey(tukey_offset, psi_data, e);
    }
}

/* opt_solver.cpp:233-268 — the per-node weight is overwritten per neighbour: last wins */
void orc_huber_weights(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, float psi_reg,
                       float* huber) {
    (void)node_w;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)D * k);
    orc_knn(node_pos, D, node_pos, D, k, idx, 1);
    for (int i = 0; i < D; ++i) {
        huber[i] = 0.f;
        for (int j = 0; j < k; ++j) {
            int m = idx[(size_t)i * k + j];
            if (m < 0) break;
            float a[3], b[3];
            orc_dq_transform_vertex(node_dq + 8 * i, node_pos + 3 * m, a);
            orc_dq_transform_vertex(node_dq + 8 * m, node_pos + 3 * m, b);
            float ex = a[0] - b[0], ey = a[1] - b[1], ez = a[2] - b[2];
            float e = sqrtf(ex * ex + ey * ey + ez * ez);
            huber[i] = fabsf(e) <= psi_reg ? 1.f : psi_reg / fabsf(e);
        }
    }
    free(idx);
}

void orc_solve_ref(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, const float* canon,
                   const float* live, int N, const orc_solve_params* prm, float* translations, float* node_dq_out,
                   orc_solve_stats* stats) {
    const int threads = prm->threads > 0 ? prm->threads : 1;
    orc_solve_stats st;
    memset(&st, 0, sizeof(st));

    /* graphs: opt_solver.cpp:56-105 */
    int32_t* data_idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)N * k);
    int32_t* reg_idx  = (int32_t*)malloc(sizeof(int32_t) * (size_t)D * k);
    float* data_w     = (float*)malloc(sizeof(float) * (size_t)N * k);
    float* tukey      = (float*)malloc(sizeof(float) * (size_t)N);
    float* huber      = (float*)malloc(sizeof(float) * (size_t)D);
    float* cur_dq     = (float*)malloc(sizeof(float) * 8 * (size_t)D);
    orc_knn(node_pos, D, canon, N, k, data_idx, threads);
    orc_knn(node_pos, D, node_pos, D, k, reg_idx, threads);
    /* energy.t:15-17,50-52 weights (same closed form as node.cpp:29-36) */
    for (size_t e = 0; e < (size_t)N * k; ++e) {
        int n     = data_idx[e];
        data_w[e] = n < 0 ? 0.f : orc_transformation_weight(node_pos + 3 * n, node_w[n], canon + 3 * (e / k));
    }

    const size_t n3 = 3 * (size_t)D;
    float* t32  = (float*)calloc(n3, sizeof(float));
    double* t64 = (double*)calloc(n3, sizeof(double));
    /* opt_solver.cpp:30 — w_reg = sqrt(lambda / (D*KNN)) */
    const double w_reg = sqrt((double)prm->lambda / ((double)D * (double)k));

    for (int outer = 0; outer < prm->num_iter; ++outer) {
        /* preNonlinearSolve (:135-140): nodes' transforms as seen with the current t */
        for (int i = 0; i < D; ++i) {
            float tdq[8];
            float tx = prm->use_double ? (float)t64[3 * i + 0] : t32[3 * i + 0];
            float ty = prm->use_double ? (float)t64[3 * i + 1] : t32[3 * i + 1];
            float tz = prm->use_double ? (float)t64[3 * i + 2] : t32[3 * i + 2];
            orc_dq_from_euler(0.f, 0.f, 0.f, tx, ty, tz, tdq);
            orc_dq_mul(tdq, node_dq + 8 * i, cur_dq + 8 * i);
        }
        orc_tukey_weights(node_pos, cur_dq, node_w, D, k, canon, live, N, prm->tukey_offset, prm->psi_data, tukey,
                          threads);
        orc_huber_weights(node_pos, cur_dq, node_w, D, k, prm->psi_reg, huber); /* computed, unused (energy.t:70) */
        if (prm->use_double) {
            problem_f64 P = {D, N, k, data_idx, data_w, reg_idx, canon, live, tukey, w_reg, threads};
            gauss_newton_f64(&P, prm, t64, &st, outer == 0);
        } else {
            problem_f32 P = {D, N, k, data_idx, data_w, reg_idx, canon, live, tukey, (float)w_reg, threads};
            gauss_newton_f32(&P, prm, t32, &st, outer == 0);
        }
    }

    for (size_t i = 0; i < n3; ++i) translations[i] = prm->use_double ? (float)t64[i] : t32[i];
    if (node_dq_out) {
        /* copyResultToCPUFromFloat3 (:270-285) + Node::updateTransformation (node.cpp:19-23) */
        for (int i = 0; i < D; ++i) {
            float tdq[8];
            orc_dq_from_euler(0.f, 0.f, 0.f, translations[3 * i], translations[3 * i + 1], translations[3 * i + 2],
                              tdq);
            orc_dq_mul(tdq, node_dq + 8 * i, node_dq_out + 8 * i);
        }
    }
    if (stats) *stats = st;
    free(data_idx), free(reg_idx), free(data_w), free(tukey), free(huber), free(cur_dq), free(t32), free(t64);
}
