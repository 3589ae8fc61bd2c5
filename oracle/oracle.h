/*
 * oracle.h — CPU restatement of the dynfu warp-solve + TSDF-fuse hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so; the product path
 * (dynfu_amd/, include/dynfu_amd.h) never links or calls anything in this directory.
 *
 * Each function cites the reference file:line it restates (paths relative to the
 * reference checkout, swarth100/dynfu).  Arithmetic conventions (shared with the HIP
 * kernels so integer/half results can be compared bit-exactly):
 *   - IEEE-754 binary32, round-to-nearest-even, subnormals kept, no FMA contraction
 *     except where written as fmaf() — those are the places where nvcc's default
 *     -fmad=true would fuse the reference's a*b+c;
 *   - the reference's approximate CUDA intrinsics (__fdividef, rsqrt, --prec-div=false,
 *     CMakeLists.txt:76-78) are replaced by the correctly-rounded operation they
 *     approximate ( / , 1/sqrtf );
 *   - half conversion is software round-to-nearest-even (== __float2half_rn).
 *
 * Pinning status:
 *   dual quaternions  : pinned  — 21 known-answer tests of test/quaternion_test.cpp
 *   k-NN graph        : pinned  — against the reference's own vendored nanoflann
 *                                 (oracle/_ref, built from /root/reference/include/nanoflann)
 *   warp-field solve  : pinned  — the 8 OptTest end-state assertions of
 *                                 test/opt_optimisation_test.cpp (tolerance 1e-3)
 *   TSDF / raycast / compute_dists : PARITY UNPINNED — the reference has no tests,
 *                                 golden vectors or CPU path for them and its CUDA
 *                                 sources cannot be built here.
 */
#ifndef DYNFU_ORACLE_H
#define DYNFU_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- half ---- */
uint16_t orc_float_to_half(float f);  /* __float2half_rn, device.hpp:59-61 */
float orc_half_to_float(uint16_t h);  /* __half2float,    device.hpp:63-67 */

/* ---------------------------------------------------------------- TSDF ---- */
/* Volume element = {u16 half-bits tsdf, u16 weight} packed little-endian in a
 * uint32 (internal.hpp:36-55, device.hpp:59-67); idx = x + y*X + z*X*Y
 * (device.hpp:20-35). Affine = 12 floats: R row-major (9) then t (3). */

/* imgproc.cu:233-245 (+ host imgproc.cpp:38-41). Steps are in BYTES. */
void orc_compute_dists(const uint16_t* depth, int depth_step, uint16_t* dists, int dists_step, int cols, int rows,
                       float fx, float fy, float cx, float cy);

/* tsdf_volume.cu:11-22 */
void orc_tsdf_clear(uint32_t* vol, int X, int Y, int Z);

/* tsdf_volume.cu:43-121. Returns number of voxels updated. threads<=1: serial. */
long orc_tsdf_integrate_slab(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* slab, int X, int Y, int z0,
                             int z1, const float voxel_size[3], float trunc_dist, int max_weight,
                             const float vol2cam[12], float fx, float fy, float cx, float cy, int threads);
long orc_tsdf_integrate(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* vol, int X, int Y, int Z,
                        const float voxel_size[3], float trunc_dist, int max_weight, const float vol2cam[12], float fx,
                        float fy, float cx, float cy, int threads);

/* tsdf_volume.cu:128-386 (points variant :258-318): points/normals are float4
 * images, steps in BYTES; misses are quiet-NaN. */
void orc_tsdf_raycast_points(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                             const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                             float step_factor, float delta_factor, float* points, int points_step, float* normals,
                             int normals_step, int cols, int rows, int threads);

/* depth variant :195-256: depth u16 millimetres (0 on miss) + normals float4. */
void orc_tsdf_raycast_depth(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                            const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                            float step_factor, float delta_factor, uint16_t* depth, int depth_step, float* normals,
                            int normals_step, int cols, int rows, int threads);

/* ------------------------------------------------------- marching cubes -- */
/* src/kfusion/cuda/marching_cubes.cu (see mc_oracle.c); PARITY UNPINNED.  Case tables are
 * arguments (256 x 16 edge ids, -1 padded; 256 vertex counts).  Returns the total vertex count;
 * writes at most max_vertices float4 points in ascending linear voxel order. */
long orc_marching_cubes(const uint32_t* vol, int X, int Y, int Z, const float cell_size[3], const int32_t* tri_table,
                        const int32_t* num_verts_table, float* out_points, long max_vertices, long* occupied);

/* ------------------------------------------------------- dual quaternion -- */
/* DualQuaternion<float> (include/dynfu/utils/dual_quaternion.hpp). Storage:
 * 8 floats = real (w,x,y,z) then dual (w,x,y,z); Hamilton product as
 * boost::math::quaternion. */
void orc_dq_from_euler(float yaw, float pitch, float roll, float x, float y, float z, float out[8]); /* :48-67 */
void orc_dq_from_quat_trans(const float q[4], const float t[3], float out[8]);                       /* :42-45 */
void orc_dq_from_rodrigues(const float rod[3], const float t[3], float out[8]);                      /* :70-86 */
void orc_dq_add(const float a[8], const float b[8], float out[8]);                                   /* :99-107 */
void orc_dq_sub(const float a[8], const float b[8], float out[8]);                                   /* :109-117 */
void orc_dq_scale(const float a[8], float s, float out[8]);                                          /* :120-125 */
void orc_dq_mul(const float a[8], const float b[8], float out[8]);                                   /* :127-135 */
void orc_dq_normalize(const float a[8], float out[8]);                                               /* :139-144 */
void orc_dq_get_translation(const float a[8], float out[3]);                                         /* :94-97 */
void orc_dq_transform_vertex(const float a[8], const float v[3], float out[3]);                      /* :204-215 */
float orc_dq_roll(const float a[8]);                                                                 /* :148-160 */
float orc_dq_pitch(const float a[8]);                                                                /* :162-176 */
float orc_dq_yaw(const float a[8]);                                                                  /* :178-190 */
void orc_dq_get_rodrigues(const float a[8], float out[3]);                                           /* :194-200 */

/* ------------------------------------------------------------ warp field -- */
/* Exact k-NN, ascending L2^2, ties broken by lower node index (nanoflann keeps
 * the first-found candidate on ties, nanoflann.hpp:104-111; equal to this
 * except on exact distance ties). warp_field.cpp:111-122. idx is (n_query x k)
 * row-major int32, -1 padded when D<k. Returns nothing. */
void orc_knn(const float* nodes, int D, const float* query, int n_query, int k, int32_t* idx, int threads);

/* node.cpp:29-36: w = exp(-|g-v|^2 / (2 dg_w^2)), evaluated in double, rounded to float. */
float orc_transformation_weight(const float g[3], float dg_w, const float v[3]);

/* warp_field.cpp:127-148: DQ "blend" = ordered product of the k neighbours'
 * dual-scaled transforms, real part normalised. node_dq is D x 8. */
void orc_calc_dqb(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, const float p[3],
                  float out[8]);

/* warp_field.cpp:150-171: warped vertices and "normals" (normals transformed as
 * points — reference quirk). */
void orc_warp_to_live(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                      const float* verts, const float* normals, int N, float* out_verts, float* out_normals,
                      int threads);

/* warp_field.cpp:34-62 */
void orc_unsupported_flags(const float* node_pos, const float* node_w, int D, int k, const float* verts, int N,
                           uint8_t* flags, int threads);
/* pcl::VoxelGrid as Warpfield::update uses it (warp_field.cpp:68-72); restated from PCL's published algorithm,
 * un-vendored dependency, PARITY UNPINNED (see warp_oracle.c).  Returns the number of output points. */
int orc_voxel_grid(const float* pts, int n, float leaf, float* out);

/* ----------------------------------------------------------------- solve -- */
typedef struct {
    int num_iter;        /* outer iterations: Tukey/Huber re-weighting (CombinedSolverParameters.numIter)      */
    int nonlinear_iter;  /* Gauss-Newton iterations per outer iteration (nonLinearIter)                        */
    int linear_iter;     /* max PCG iterations per GN iteration (linearIter)                                   */
    float tukey_offset;  /* opt_solver.cpp:204-212 */
    float psi_data;      /* Tukey cut-off c */
    float lambda;        /* regularisation weight; w_reg = sqrt(lambda/(D*k)) opt_solver.cpp:30 */
    float psi_reg;       /* Huber k (computed, unused by energy.t) */
    float pcg_tol;       /* relative preconditioned-residual tolerance, 0 = run linear_iter iterations */
    float gn_tol;        /* relative cost-decrease tolerance, 0 = run all GN iterations */
    int use_double;      /* 1: accumulate in double (reference tests: optDoublePrecision=true) */
    int threads;
} orc_solve_params;

typedef struct {
    double initial_cost;
    double final_cost;
    double grad_first; /* g.M^-1.g of the first linearisation (scale for the convergence floor) */
    int gn_iters;      /* total GN iterations executed */
    int pcg_iters;     /* total PCG iterations executed */
} orc_solve_stats;

/* Reference-parity solve (energy.t:19-78 + opt_solver.cpp:15-285).
 *   node_pos D x 3, node_dq D x 8 (current dg_se3, used for the Tukey weights
 *   exactly as updateTukeyBiweights does), node_w D;
 *   canon/live N x 3. Output: translations D x 3 (the Opt unknown), and — if
 *   node_dq_out != NULL — the written-back transforms DQ(t_i) * dg_se3_i
 *   (opt_solver.cpp:270-285, node.cpp:19-23), composed ONCE. */
void orc_solve_ref(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, const float* canon,
                   const float* live, int N, const orc_solve_params* prm, float* translations, float* node_dq_out,
                   orc_solve_stats* stats);

/* Building blocks, exposed for kernel-level parity tests. */
/* opt_solver.cpp:204-231 */
void orc_tukey_weights(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                       const float* canon, const float* live, int N, float tukey_offset, float psi_data, float* tukey,
                       int threads);
/* opt_solver.cpp:233-268 */
void orc_huber_weights(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, float psi_reg,
                       float* huber);

/* -------------------------------------------------- depth pre-processing -- */
/* src/kfusion/cuda/imgproc.cu (see img_oracle.c); PARITY UNPINNED.  Steps are in bytes. */
float orc_exp_neg(float x); /* the stand-in for __expf in the bilateral weight (x <= 0) */
void orc_bilateral(const uint16_t* src, int src_step, uint16_t* dst, int dst_step, int cols, int rows, int ksz,
                   float sigma_spatial, float sigma_depth);
void orc_truncate_depth(uint16_t* depth, int step, int cols, int rows, float max_dist);
void orc_depth_pyr(const uint16_t* src, int src_step, int cols, int rows, uint16_t* dst, int dst_step, float sigma_depth);
void orc_normals_mask_depth(uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx, float cy,
                            float* normals, int normals_step);
void orc_resize_depth_normals(const uint16_t* dsrc, int dsrc_step, const float* nsrc, int nsrc_step, int cols, int rows,
                              uint16_t* ddst, int ddst_step, float* ndst, int ndst_step);
void orc_resize_points_normals(const float* vsrc, int vsrc_step, const float* nsrc, int nsrc_step, int cols, int rows,
                               float* vdst, int vdst_step, float* ndst, int ndst_step);

/* ------------------------------------------------------------- rigid ICP -- */
/* one linearisation of src/kfusion/cuda/proj_icp.cu (see icp_oracle.c); PARITY UNPINNED.
 * depth_variant: curr / prev are u16 depth images; otherwise float4 vertex maps.  aff: R row-major then t. */
void orc_icp_sums(int depth_variant, const void* curr, int curr_step, const float* ncurr, int ncurr_step, const void* prev,
                  int prev_step, const float* nprev, int nprev_step, int cols, int rows, const float aff[12], float fx,
                  float fy, float cx, float cy, float dist_thres, float angle_thres, double sums[27], long* matched);

/* ------------------------------------------------- north-star solve (6-DoF) -- */
/* solve6_oracle.c: NOT in the reference's code (BASELINE.json north_star);
 * PARITY UNPINNED, formulas in DESIGN.md §4.5. */
typedef struct {
    int num_iter;       /* outer iterations: Tukey / Huber re-weighting            */
    int gn_iter;        /* Gauss-Newton iterations per outer iteration             */
    int linear_iter;    /* max PCG iterations per GN iteration                     */
    float tukey_offset; /* data residual scale (opt_solver.cpp:204-212)            */
    float psi_data;     /* Tukey cut-off                                           */
    float lambda;       /* regulariser weight: w_reg^2 = lambda / (D k)            */
    float psi_reg;      /* Huber threshold of the regulariser                      */
    float dist_thresh;  /* association gate: |p - l| <= dist_thresh  (metres)      */
    float cos_thresh;   /* association gate: n_warped . n_live >= cos_thresh       */
    float damping;      /* added to the diagonal of the normal matrix              */
    float pcg_tol;      /* stop when (r.z) <= pcg_tol^2 (r.z)_0                    */
    /* inexact-Newton forcing schedule: Gauss-Newton iteration i (counted from 0 within its outer iteration) stops
     * its PCG at max(pcg_tol, pcg_tol_first * pcg_tol_decay^i); pcg_tol_first <= 0 = constant pcg_tol */
    float pcg_tol_first, pcg_tol_decay;
    /* pcg_tol_adapt > 0: Eisenstat-Walker forcing (choice 2, alpha = 2) instead of the geometric schedule — the first
     * iteration of an outer iteration stops at pcg_tol_first, iteration i > 0 at
     * clamp(pcg_tol_adapt (r.z)_0,i / (r.z)_0,i-1, pcg_tol, pcg_tol_first): tight while the gradient still falls fast,
     * loose once it stagnates (at the noise floor of the data) */
    float pcg_tol_adapt;
    int threads;
    /* Gauss-Newton stopping rule and step acceptance (the reference runs Opt with earlyOut = true and nonLinearIter as a
     * cap: src/dynfu/dyn_fusion.cpp:183-189, test/opt_optimisation_test.cpp:43).  gn_tol <= 0: every outer iteration runs
     * its gn_iter iterations.  gn_tol > 0: with E_ref the energy at the last accepted linearisation of the outer iteration
     * (weights frozen) and E the energy re-linearised after a step,
     *     E > (1 + gn_tol) E_ref         the step is REJECTED: the transforms before it are restored, the outer iteration ends
     *     E_ref - E <= gn_tol E_ref      CONVERGED: the step is kept, the outer iteration ends (no normal equations)
     *     otherwise                      E_ref = E, the iteration goes on
     * and the last step of the solve, if its outer iteration ran to the cap, is checked by one closing linearisation
     * (rejected or kept by the first test). */
    float gn_tol;
    /* != 0: Gauss-Newton iterations >= 1 of an outer iteration keep the normal matrix and the preconditioner of
     * iteration 0 and re-linearise residuals and gradient only (the pattern of the reference-parity solve's inner
     * iterations: dynfu_amd/csrc/solve.hip regradient) */
    int reuse_matrix;
} orc6_params;

#define ORC6_HIST 32

typedef struct {
    double initial_cost; /* energy at the first linearisation                      */
    double final_cost;   /* energy at the last (gn_tol > 0: last accepted) linearisation */
    int gn_iters, pcg_iters;
    long valid_first, valid_last; /* data rows with a valid association and non-zero weight */
    /* per Gauss-Newton iteration (the first ORC6_HIST): energy at its linearisation, PCG iterations it ran, and the
     * relative residual sqrt((r.z) / (r.z)_0) its PCG stopped at */
    double cost_hist[ORC6_HIST], pcg_rel_hist[ORC6_HIST];
    int pcg_it_hist[ORC6_HIST];
    double pcg_tol_hist[ORC6_HIST]; /* the relative residual every PCG was asked for */
    /* With gn_tol > 0 the histories are indexed by outer * gn_iter + gn (slot num_iter * gn_iter: the closing check);
     * stop_hist: 0 linearised and solved, 1 converged here, 2 rejected here (cost_hist holds the rejected energy), 3 skipped
     * (the outer iteration had ended).  gn_iters counts the linearisations evaluated, gn_solves the normal equations solved. */
    long valid_hist[ORC6_HIST];
    int stop_hist[ORC6_HIST];
    int gn_solves, gn_rejected, gn_converged, hist_n;
} orc6_stats;

/* kfusion::device::computePointNormals (src/kfusion/cuda/imgproc.cu:187-215): float4 vertex and
 * normal maps of a depth image (NaN where undefined), fp32 as the kernel. */
void orc6_points_normals(const uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                         float cy, float* points, int points_step, float* normals, int normals_step);
/* k-NN + normalised RBF weights (N x k), and the k nearest other nodes of every node (D x k) */
void orc6_graph(const float* node_pos, const float* node_w, int D, int k, const float* canon, int N, int32_t* idx,
                float* wn, int32_t* reg_idx, int threads);
/* dual-quaternion blend warp of vertices (and normals: rotation only) */
void orc6_warp(const float* node_dq, int k, const int32_t* idx, const float* wn, const float* canon,
               const float* canon_n, int N, float* out_p, float* out_n);
/* d p / d (twist of neighbour j): J[j][component][xyz], and the warped point */
void orc6_data_jacobian(const float* node_pos, const float* node_dq, const int32_t* idx, const float* wn, int k,
                        const float c[3], double* J, double p_out[3]);
/* left twist (omega, v) about the node's current position */
void orc6_apply_twist(const float node_pos_i[3], const float dq_in[8], const double twist[6], float dq_out[8]);
void orc6_solve(const float* node_pos, const float* node_dq_in, const float* node_w, int D, int k, const float* canon,
                const float* canon_n, int N, const float* vmap, int vmap_step, const float* nmap, int nmap_step, int cols,
                int rows, float fx, float fy, float cx, float cy, const orc6_params* prm, float* node_dq_out,
                orc6_stats* stats);
/* development hook: copy out the normal equations (block CSR) of Gauss-Newton iteration `gn` of the next orc6_solve */
void orc6_set_dump(int gn, int* row_ptr, int* cols, double* blk, double* g, long cap, long* nblk);
double orc6_cost(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, const float* canon,
                 const float* canon_n, int N, const float* vmap, int vmap_step, const float* nmap, int nmap_step, int cols,
                 int rows, float fx, float fy, float cx, float cy, const orc6_params* prm, long* nvalid_out);

#ifdef __cplusplus
}
#endif
#endif
