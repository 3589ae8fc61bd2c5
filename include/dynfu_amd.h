/*
 * dynfu_amd.h — C ABI of the MI355X-native (gfx950) implementation of dynfu's per-frame hot
 * path: TSDF clear / integrate / raycast + compute_dists, and the warp-field solve
 * (k-NN graph, RBF weights, DQ warp, Tukey re-weighting, Gauss-Newton / block-Jacobi PCG).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ / torch / OpenCV types.
 * Every entry point names the reference interface it replaces (paths relative to the
 * swarth100/dynfu checkout).  INTEGRATION.md shows the reference-side bindings.
 *
 * Conventions
 *   - all data pointers are DEVICE pointers (hipMalloc / torch CUDA tensors) unless the
 *     parameter is documented as host; the caller owns them, kernels allocate nothing
 *     (same ownership as kfusion::cuda::DeviceMemory, device_memory.cpp:50-113).  Entry points
 *     without a plan keep small internal scratch (search grids, scan partials, the depth tile
 *     table of the TSDF sweeps) per (device, stream): calls on different streams never share
 *     it, calls on one stream are ordered by the stream; it grows on demand and is kept;
 *   - threads: any host thread may call any entry point; two threads must not drive the SAME
 *     stream or the SAME solver plan at the same time (as with any stream-ordered API);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is
 *     asynchronous on it.  NOTE the reference's device::integrate ends with
 *     cudaDeviceSynchronize (tsdf_volume.cu:120); the adaptor class syncs where the caller
 *     relied on that;
 *   - return value: DFA_OK or an error code; dfa_last_error() gives the message for the
 *     calling thread (the reference prints and exit(0)s instead, safe_call.hpp:11-22 — the
 *     C++ adaptor may translate);
 *   - affine transforms are 12 floats: R row-major (9) then t (3) == device::Aff3f
 *     (internal.hpp:28-34); image `step`s are in BYTES == PtrStep (kernel_containers.hpp);
 *   - TSDF voxel = uint32: low 16 bits half-float tsdf (RNE), high 16 bits weight
 *     == ushort2{x,y} (internal.hpp:38, device.hpp:59-67); idx = x + y*X + z*X*Y.
 */
#ifndef DYNFU_AMD_H
#define DYNFU_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dfa_stream_t;

enum {
    DFA_OK           = 0,
    DFA_ERR_INVALID  = 1, /* bad argument (null pointer, non-positive size, unsupported k, ...) */
    DFA_ERR_HIP      = 2, /* a HIP runtime call or kernel launch failed */
    DFA_ERR_CAPACITY = 3, /* a solver plan is too small for the problem handed to it */
    DFA_ERR_NO_GPU   = 4  /* no HIP device visible */
};

/* message of the last failing call on this thread ("" if none) */
const char* dfa_last_error(void);
/* library / device identification: "dynfu_amd <ver> gfx950 ..." ; static storage */
const char* dfa_version(void);

/* ABI guard.  The parameter / statistics structs below are passed by pointer and carry no size field of their own:
 * DFA_ABI_VERSION changes whenever the layout of one of them does.  dfa_abi_version() is the value the LIBRARY was built
 * with, dfa_abi_struct_size(id) the sizeof it assumes (0 for an unknown id) — a caller built against another header, or a
 * binding that mirrors the structs by hand (dynfu_amd/_lib.py), compares both with its own before the first call. */
#define DFA_ABI_VERSION 6
enum {
    DFA_STRUCT_SOLVE_PARAMS  = 0, /* dfa_solve_params  */
    DFA_STRUCT_SOLVE_STATS   = 1, /* dfa_solve_stats   */
    DFA_STRUCT_SOLVE_TIMING  = 2, /* dfa_solve_timing  */
    DFA_STRUCT_SOLVE6_PARAMS = 3, /* dfa_solve6_params */
    DFA_STRUCT_SOLVE6_STATS  = 4, /* dfa_solve6_stats  */
    DFA_STRUCT_SOLVE6_TIMING = 5  /* dfa_solve6_timing */
};
int dfa_abi_version(void);
size_t dfa_abi_struct_size(int struct_id);

/* ===================================================================================== */
/* TSDF seam — replaces the free functions of namespace kfusion::device                  */
/* (include/kfusion/internal.hpp:158-166,204)                                            */

/* compute_dists — internal.hpp:204, imgproc.cu:233-254.
 * dists(y,x) = half(depth_mm * sqrt(((x-cx)/fx)^2 + ((y-cy)/fy)^2 + 1) * 0.001) */
int dfa_compute_dists(const uint16_t* depth, int depth_step, uint16_t* dists, int dists_step, int cols, int rows,
                      float fx, float fy, float cx, float cy, dfa_stream_t stream);

/* clear_volume — internal.hpp:159, tsdf_volume.cu:11-34 */
int dfa_tsdf_clear(uint32_t* volume, int X, int Y, int Z, dfa_stream_t stream);

/* integrate — internal.hpp:160, tsdf_volume.cu:43-121.
 * dists: half-float ray lengths in metres (cols x rows, dists_step bytes per row). */
int dfa_tsdf_integrate(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X, int Y, int Z,
                       const float voxel_size[3], float trunc_dist, int max_weight, const float vol2cam[12], float fx,
                       float fy, float cx, float cy, dfa_stream_t stream);

/* clear_volume + integrate in one sweep (what DynFusion::operator() does every frame,
 * src/dynfu/dyn_fusion.cpp:113-116): bit-identical to dfa_tsdf_clear followed by
 * dfa_tsdf_integrate, with half the HBM traffic (no read of the old volume). */
int dfa_tsdf_clear_integrate(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X, int Y,
                             int Z, const float voxel_size[3], float trunc_dist, int max_weight,
                             const float vol2cam[12], float fx, float fy, float cx, float cy, dfa_stream_t stream);

/* Occupancy map of a volume (optional).  The sweep classifies every run of 8 voxels of a column against the depth image
 * before it touches it (csrc/tsdf_classify.hpp) and so knows, for free, which parts of the volume it left empty — 5/6 of a
 * 512^3 volume fused from one depth frame.  The `_occ` entry points keep that knowledge in a byte map — one byte per box
 * of 32 x 2 x 8 voxels (x, y, z), byte ((z / 8) ceil(Y / 2) + y / 2) ceil(X / 32) + x / 32; bit 0 = a voxel of the box MAY
 * have a non-zero weight, bit 1 = one MAY hold a negative distance (a run the sweep had to evaluate voxel by voxel; the runs
 * in front of the surface are filled with +1) — that dfa_marching_cubes_occ reads instead of the empty voxels (the reference's OccupiedVoxels
 * pass reads the whole volume, src/kfusion/cuda/marching_cubes.cu:77-142).  Same volume bits, same mesh bits as the
 * entry points without a map.  The caller owns the map (dfa_tsdf_occupancy_bytes bytes, device); the fused sweep writes
 * all of it, dfa_tsdf_integrate_occ only sets bytes, dfa_tsdf_clear_occ zeroes volume and map. */
size_t dfa_tsdf_occupancy_bytes(int X, int Y, int Z);
int dfa_tsdf_clear_occ(uint32_t* volume, int X, int Y, int Z, uint8_t* occupancy, dfa_stream_t stream);
int dfa_tsdf_integrate_occ(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X, int Y, int Z,
                           const float voxel_size[3], float trunc_dist, int max_weight, const float vol2cam[12], float fx,
                           float fy, float cx, float cy, uint8_t* occupancy, dfa_stream_t stream);
int dfa_tsdf_clear_integrate_occ(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X, int Y,
                                 int Z, const float voxel_size[3], float trunc_dist, int max_weight,
                                 const float vol2cam[12], float fx, float fy, float cx, float cy, uint8_t* occupancy,
                                 dfa_stream_t stream);
/* dfa_tsdf_clear_integrate_occ for a map that DESCRIBES THE VOLUME ON ENTRY (a map as dfa_tsdf_clear_occ or one of the _occ
 * sweeps left it, the volume untouched by anything else since): a box of zeros whose byte says so and that receives no weight
 * from this frame either is not stored again — 5/6 of a 512^3 sweep's traffic.  Same volume bits and same map as
 * dfa_tsdf_clear_integrate_occ.  With a map that does not describe the volume, boxes of the old content survive. */
int dfa_tsdf_clear_integrate_known_occ(const uint16_t* dists, int dists_step, int cols, int rows, uint32_t* volume, int X,
                                       int Y, int Z, const float voxel_size[3], float trunc_dist, int max_weight,
                                       const float vol2cam[12], float fx, float fy, float cx, float cy, uint8_t* occupancy,
                                       dfa_stream_t stream);

/* Normals of surface points from the TSDF gradient (the reference extracts the mesh without
 * normals, dyn_fusion.cpp:80-88 "temporary workaround until normals are computed via mc").  The raycaster's own
 * compute_normal (tsdf_volume.cu:320-336): central differences of the trilinear interpolant, gradient_delta_factor
 * voxels apart, normalised; NaN where a sample leaves the interpolation range [0, dim - 1).
 * points / normals: n float4 (xyz + pad), device, 16-byte aligned; points in the volume's metric frame, as
 * dfa_marching_cubes emits them. */
int dfa_tsdf_vertex_normals(const uint32_t* volume, int X, int Y, int Z, const float voxel_size[3],
                            float gradient_delta_factor, const float* points, int n, float* normals,
                            dfa_stream_t stream);

/* raycast (points variant) — internal.hpp:165-166, tsdf_volume.cu:258-318,371-386.
 * points / normals: float4 images; misses are quiet NaN. */
int dfa_tsdf_raycast_points(const uint32_t* volume, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                            const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                            float step_factor, float delta_factor, float* points, int points_step, float* normals,
                            int normals_step, int cols, int rows, dfa_stream_t stream);

/* raycast (depth variant) — internal.hpp:162-163, tsdf_volume.cu:195-256,354-369.
 * depth: u16 millimetres, 0 on miss. */
int dfa_tsdf_raycast_depth(const uint32_t* volume, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                           const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                           float step_factor, float delta_factor, uint16_t* depth, int depth_step, float* normals,
                           int normals_step, int cols, int rows, dfa_stream_t stream);

/* Measurement entry point (no reference counterpart): the rays of dfa_tsdf_raycast_points / _depth cast once more with
 * their WORK counted — the quantities SURVEY 8(d) prices the raycast by.  counts[4] (device, zeroed here): rays that
 * enter the volume, nearest-voxel fetches of the march (tsdf_volume.cu:187-193 via :222,:230), hits (:246), voxel
 * fetches of the trilinear samples (:146-171: 8 per sample).  touched_bits: one bit per voxel, X*Y*Z/32 words zeroed by
 * the caller, set for every voxel any ray reads (its population count is the lower bound "unique voxels"); may be null. */
int dfa_tsdf_raycast_tally(const uint32_t* volume, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                           const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                           float step_factor, float delta_factor, int cols, int rows, uint64_t* counts,
                           uint32_t* touched_bits, dfa_stream_t stream);

/* ===================================================================================== */
/* Depth pre-processing seam — replaces the image kernels of kfusion::device declared in    */
/* include/kfusion/internal.hpp:190-204 (src/kfusion/cuda/imgproc.cu), called by            */
/* DynFusion::operator() (src/dynfu/dyn_fusion.cpp:58-66) and KinFu::operator()             */
/* (src/kfusion/kinfu.cpp:150-175).  Images are pitched: *_step = bytes per row.            */

/* cuda::depthBilateralFilter (imgproc.cpp:3-7, imgproc.cu:8-52).  sigma_depth in metres.
 * The reference's __expf is replaced by a fixed IEEE operation sequence (csrc/img.hip). */
int dfa_depth_bilateral_filter(const uint16_t* src, int src_step, uint16_t* dst, int dst_step, int cols, int rows,
                               int kernel_size, float sigma_spatial, float sigma_depth, dfa_stream_t stream);
/* cuda::depthTruncation (imgproc.cpp:9, imgproc.cu:60-79): depth > max_dist metres -> 0, in place. */
int dfa_depth_truncate(uint16_t* depth, int depth_step, int cols, int rows, float max_dist, dfa_stream_t stream);
/* cuda::depthBuildPyramid (imgproc.cpp:11-14, imgproc.cu:84-124): dst is (rows/2) x (cols/2). */
int dfa_depth_build_pyramid(const uint16_t* src, int src_step, int cols, int rows, uint16_t* dst, int dst_step,
                            float sigma_depth, dfa_stream_t stream);
/* cuda::computeNormalsAndMaskDepth (imgproc.cpp:18-25, imgproc.cu:129-183): float4 normals (NaN where
 * undefined), and depth zeroed where the normal is undefined. */
int dfa_compute_normals_mask_depth(uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                                   float cy, float* normals, int normals_step, dfa_stream_t stream);
/* cuda::resizeDepthNormals (imgproc.cpp:43-52, imgproc.cu:258-311): outputs are (rows/2) x (cols/2). */
int dfa_resize_depth_normals(const uint16_t* depth, int depth_step, const float* normals, int normals_step, int cols,
                             int rows, uint16_t* depth_out, int depth_out_step, float* normals_out, int normals_out_step,
                             dfa_stream_t stream);
/* cuda::resizePointsNormals (imgproc.cpp:54-66, imgproc.cu:314-358). */
int dfa_resize_points_normals(const float* points, int points_step, const float* normals, int normals_step, int cols,
                              int rows, float* points_out, int points_out_step, float* normals_out, int normals_out_step,
                              dfa_stream_t stream);

/* ===================================================================================== */
/* Rigid-ICP seam — replaces device::ComputeIcpHelper::operator() (include/kfusion/internal.hpp:121-156,   */
/* src/kfusion/cuda/proj_icp.cu), one linearisation of cuda::ProjectiveICP::estimateTransform              */
/* (src/kfusion/projective_icp.cpp:118-200).  The reference's DynFusion skips the rigid tracker             */
/* (dyn_fusion.cpp:100-105); KinFu::operator() uses it.                                                     */

/* sums27 (device, 27 floats): for i in 0..5, for j in i..6 the sum over the current frame's pixels of
 * row[i] * row[j], row = (s x n, n, n.(d - s)) — the buffer layout ProjectiveICP::StreamHelper::get reads
 * (projective_icp.cpp:39-57).  depth_variant != 0: curr / prev are u16 depth images (proj_icp.cu:41-71), else
 * float4 vertex maps (:73-101); normals are float4 maps.  aff = current estimate (R row-major, then t);
 * fx..cy = intrinsics OF THE PYRAMID LEVEL (setLevelIntr); matched (device, optional) = pixels used. */
int dfa_icp_sums(int depth_variant, const void* curr, int curr_step, const float* ncurr, int ncurr_step, const void* prev,
                 int prev_step, const float* nprev, int nprev_step, int cols, int rows, const float aff[12], float fx,
                 float fy, float cx, float cy, float dist_thres, float angle_thres, float* sums27, unsigned int* matched,
                 dfa_stream_t stream);

/* ===================================================================================== */
/* Marching-cubes seam — replaces kfusion::device::{bindTextures, getOccupiedVoxels,        */
/* computeOffsetsAndTotalVertices, generateTriangles} (include/kfusion/internal.hpp:121-150, */
/* src/kfusion/cuda/marching_cubes.cu), driven by cuda::MarchingCubes::run                  */
/* (src/kfusion/marching_cubes.cpp:20-61)                                                   */

/* One call = the whole extraction, nothing synchronises with the host.
 * volume: packed TSDF as above, any dims >= 2 (the reference is fixed to 128^3).
 * cell_size: edge lengths of a cell (the reference passes volume_size / 128 = the voxel size,
 *   marching_cubes.cu:283-285).
 * tri_table (256 x 16 edge ids, -1 padded) and num_verts_table (256): DEVICE pointers, the
 *   buffers the reference binds as textures (marching_cubes.cu:14-19).
 * out_points: max_vertices float4 {x, y, z, 1} (store_point :255-257), three per triangle, in
 *   ascending linear voxel order z*X*Y + y*X + x (the reference's order is atomics-dependent);
 *   may be NULL with max_vertices = 0 to count only.
 * total_vertices (device, optional): vertices the volume produces; when it exceeds max_vertices
 *   only the first max_vertices were written. */
int dfa_marching_cubes(const uint32_t* volume, int X, int Y, int Z, const float cell_size[3],
                       const int32_t* tri_table, const int32_t* num_verts_table, float* out_points, int max_vertices,
                       int32_t* total_vertices, dfa_stream_t stream);

/* ... with the volume's occupancy map (see dfa_tsdf_occupancy_bytes): the count sweep loads only the slices whose boxes
 * the map marks; identical output. */
int dfa_marching_cubes_occ(const uint32_t* volume, const uint8_t* occupancy, int X, int Y, int Z, const float cell_size[3],
                           const int32_t* tri_table, const int32_t* num_verts_table, float* out_points, int max_vertices,
                           int32_t* total_vertices, dfa_stream_t stream);

/* A valid case table pair for callers that do not have the reference's (HOST buffers, 256 x 16
 * and 256 ints): derived from the face-by-face rule described in csrc/mc.hip.  Same corner / edge
 * numbering and winding as the reference's table (src/kfusion/marching_cubes.cpp:86-343) but not
 * the same triangulation of every case — pass the reference's arrays for identical meshes. */
int dfa_mc_default_tables(int32_t* tri_table, int32_t* num_verts_table);

/* ===================================================================================== */
/* Warp-field seam — replaces the per-vertex CPU loops of class Warpfield                */
/* (src/dynfu/warp_field.cpp:99-171, src/dynfu/utils/node.cpp:29-36)                     */

#define DFA_MAX_KNN 16

/* Warpfield::findNeighborsIndex for n_query points at once (warp_field.cpp:111-122; the
 * reference's KNN is the compile-time constant 8, warp_field.hpp:27 — here 1 <= k <= 16).
 * idx: n_query x k int32, ascending squared distance, ties to the lower node index,
 * -1 padded when D < k.  weights (optional, may be NULL): Node::getTransformationWeight
 * (node.cpp:29-36) of each neighbour, 0 for padding. */
int dfa_knn(const float* node_pos, const float* node_w, int D, const float* query, int n_query, int k, int32_t* idx,
            float* weights, dfa_stream_t stream);

/* Warpfield::warpToLive (warp_field.cpp:150-171) = per vertex calcDQB (:127-148) +
 * DualQuaternion::transformVertex / transformNormal (dual_quaternion.hpp:204-228).
 * node_dq: D x 8 floats (real w,x,y,z ; dual w,x,y,z).  normals / out_normals may be NULL. */
int dfa_warp_to_live(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                     const float* vertices, const float* normals, int N, float* out_vertices, float* out_normals,
                     dfa_stream_t stream);

/* The same warp with the neighbours given (idx: N x k int32 as dfa_knn returns them, -1 padded): identical output to
 * dfa_warp_to_live without the search.  For callers that warp the same vertices again while the node set is unchanged
 * (every frame of a sequence warps the canonical cloud, dyn_fusion.cpp:196). */
int dfa_warp_to_live_graph(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                           const int32_t* idx, const float* vertices, const float* normals, int N, float* out_vertices,
                           float* out_normals, dfa_stream_t stream);

/* Warpfield::calcDQB (warp_field.cpp:127-148) at n points: the blended dual quaternion itself, n x 8 floats
 * (Warpfield::update seeds a new node with it, warp_field.cpp:78).  node_dq may be NULL only if out_dq is. */
int dfa_calc_dqb(const float* node_pos, const float* node_dq, const float* node_w, int D, int k, const float* points,
                 int n, float* out_dq, dfa_stream_t stream);

/* Warpfield::getUnsupportedVertices (warp_field.cpp:34-62): flag[v] = 1 when min over the k nearest nodes of
 * |v - dg_v| / dg_w is >= 1 (also when there is no node at all).  The reference's KNN is 8. */
int dfa_unsupported_vertices(const float* node_pos, const float* node_w, int D, int k, const float* vertices, int N,
                             uint8_t* flags, dfa_stream_t stream);

/* Point-cloud plumbing between the seams (no arithmetic, bit copies).  The reference moves clouds between its
 * stages as pcl::PointCloud objects on the host; with the clouds resident in HBM the same two steps are:
 *
 * dfa_repack_points — n xyz triples from an array with src_stride floats per point to one with dst_stride floats per
 *   point (both >= 3); when dst_stride > 3 the remaining floats of every point are set to `pad`.  Converts between the
 *   float4 {x, y, z, 1} points of dfa_marching_cubes / pcl::PointXYZ (pcl::copyPointCloud, dyn_fusion.cpp:80-88) and the
 *   packed N x 3 arrays of the warp-field and solver seams.
 * dfa_compact_points — the points whose flag is non-zero, in ascending index order: the push_back loop of
 *   Warpfield::getUnsupportedVertices (warp_field.cpp:42-59) over dfa_unsupported_vertices' flags.  out_points
 *   (capacity N x 3) and out_index (capacity N) may each be NULL; count: device int32, the number of survivors. */
int dfa_repack_points(const float* src, int src_stride, float* dst, int dst_stride, int n, float pad,
                      dfa_stream_t stream);
int dfa_compact_points(const float* points, const uint8_t* flags, int N, float* out_points, int32_t* out_index,
                       int32_t* count, dfa_stream_t stream);
/* dfa_transform_points — out = R p + t (points) or R p (with_translation = 0: normals) for n packed xyz triples; aff = R
 * row-major (9) then t (3), the 12 floats of every affine in this header; out may be the input.  cv::Affine3f applied to a
 * cloud: the volume-frame vertices of marching cubes in the camera frame (TsdfVolume::getPose, tsdf_volume.cpp:83). */
int dfa_transform_points(const float* points, int n, const float aff[12], int with_translation, float* out,
                         dfa_stream_t stream);

/* DynFusion::findCorrespondingFrame (src/dynfu/dyn_fusion.cpp:212-242): for each of the
 * n_live live vertices the nearest of the n_canon (warped) canonical vertices — exact 1-NN,
 * ties to the lower index — and the canonical vertex / normal at that index gathered into
 * clouds index-aligned with the live one (the "corresponding canonical frame" handed to
 * CombinedSolver::initializeProblemInstance, dyn_fusion.cpp:206).  Replaces the per-frame
 * nanoflann KD-tree over the canonical cloud (:221-224) by a device-built uniform grid.
 * canon_normals / out_normals / out_vertices / out_index may each be NULL (skipped). */
int dfa_correspond(const float* canon_vertices, const float* canon_normals, int n_canon, const float* live_vertices,
                   int n_live, float* out_vertices, float* out_normals, int32_t* out_index, dfa_stream_t stream);

/* Projective association — the O(N) alternative to dfa_correspond: every vertex (camera frame of the
 * live maps; typically the warped canonical cloud) is projected with the intrinsics and takes the live vertex / normal
 * of the pixel it lands on, under the gates of ComputeIcpHelper::find_coresp (proj_icp.cu:72-98): point-sampled fetch,
 * |v - v'| <= dist_thresh, and — when both `normals` and `nmap` are given — |n . n'| >= min_cosine.
 * vmap / nmap: rows x cols float4 images (NaN x = undefined), pitches in bytes, as dfa_compute_points_normals writes
 * them.  Outputs n x 3 floats (NaN where there is no association) and the pixel index y * cols + x (-1: none);
 * normals, nmap, out_normals, out_vertices, out_pixel may be NULL (out_normals needs nmap). */
int dfa_correspond_projective(const float* vertices, const float* normals, int n, const float* vmap, int vmap_step,
                              const float* nmap, int nmap_step, int cols, int rows, float fx, float fy, float cx, float cy,
                              float dist_thresh, float min_cosine, float* out_vertices, float* out_normals,
                              int32_t* out_pixel, dfa_stream_t stream);

/* ===================================================================================== */
/* Solver seam — replaces class CombinedSolver (include/dynfu/utils/opt_solver.hpp:19-110, */
/* src/dynfu/utils/opt_solver.cpp) and the Opt GN/PCG it drives with energy.t            */

typedef struct dfa_solver dfa_solver; /* opaque plan: owns all scratch device memory */

typedef struct {
    int num_iter;       /* outer iterations with Tukey/Huber re-weighting (CombinedSolverParameters.numIter) */
    int nonlinear_iter; /* Gauss-Newton iterations per outer iteration (nonLinearIter)                       */
    int linear_iter;    /* max PCG iterations per GN iteration (linearIter)                                  */
    float tukey_offset; /* CombinedSolver ctor, opt_solver.cpp:3-13                                          */
    float psi_data;
    float lambda;  /* >= 0 and finite (anything else: DFA_ERR_INVALID).  w_reg^2 = lambda / (D k) is one of the addends of the
                      normal matrix, whose 64-bit fixed-point sums scale themselves by the largest addend of the problem
                      (tiny RBF weights or a huge lambda cost no bits; addends below 2^-17 of the largest lose theirs) */
    float psi_reg;
    float pcg_tol; /* relative preconditioned-residual tolerance; 0 = machine floor only */
    float gn_tol;  /* relative cost-decrease tolerance; 0 = run all GN iterations         */
} dfa_solve_params;

typedef struct {
    double initial_cost; /* sum |r|^2 at t = 0 (Opt convention) */
    double final_cost;
    int gn_iters;  /* Gauss-Newton iterations executed in total */
    int pcg_iters; /* PCG iterations executed in total */
    int max_row_nnz; /* widest row of the assembled normal matrix (blocks) */
    int gn_noop;   /* of gn_iters: iterations behind one whose gradient was at the round-off floor.  The unknown can
                      no longer change — for the rest of the solve when that linearisation had re-evaluated the robust
                      weights, until the next re-weighting (the end of the outer iteration) otherwise: the energy is
                      linear least squares while the weights are frozen — so their kernels return at entry (same
                      result as running them) */
} dfa_solve_stats;

/* Plan for up to max_D nodes / max_N vertices with k neighbours (1..16). */
int dfa_solver_create(int max_D /* <= 32768 */, int max_N, int k, dfa_solver** out);
void dfa_solver_destroy(dfa_solver* s);

/* CombinedSolver::initializeProblemInstance (opt_solver.cpp:15-54): uploads nothing (inputs
 * are already device-resident), builds the data graph (vertex -> k nodes, :56-72), the
 * regularisation graph (node -> k nodes, :74-105) and the RBF weights on the device.
 * node_dq are the nodes' current transforms dg_se3 (used by the Tukey weights exactly as
 * updateTukeyBiweights :214-231 uses calcDQB).  Normals are accepted for interface parity
 * (energy.t declares them, :28-31) and unused by the reference energy; may be NULL. */
int dfa_solver_set_problem(dfa_solver* s, const float* node_pos, const float* node_dq, const float* node_w, int D,
                           const float* canon_vertices, const float* canon_normals, const float* live_vertices,
                           const float* live_normals, int N, dfa_stream_t stream);

/* CombinedSolverBase::solveAll() with the hooks of opt_solver.cpp:107-147: per outer
 * iteration Tukey + Huber weights, then Gauss-Newton with a block-Jacobi PCG on the normal
 * equations of energy.t.  Up to 2048 nodes and 8 Gauss-Newton iterations (num_iter x nonlinear_iter) everything is
 * enqueued on `stream` without any host synchronisation; with a larger iteration budget (the reference's 24 x 16) the
 * inner iterations (nonlinear_iter > 1: robust weights frozen, so the energy is linear least squares) keep the matrix of
 * the outer iteration's first linearisation and restart the PCG on g - A (t - t_0) when lambda > 0 and gn_tol = 0; the
 * call reads the plan's `converged` flag back every 4th iteration of an outer iteration and stops launching its
 * remaining (or all remaining) iterations once it is set (the iterations not launched are booked as no-ops, like those
 * whose kernels return at entry).  Plans of 2 049 .. ~19 000 nodes run their PCG as three teams of persistent workgroups
 * (a coordinate per XCD, one barrier per iteration through that XCD's L2: DESIGN.md 4.3) — also without any host
 * synchronisation up to 8 Gauss-Newton iterations, with the same read-back every 4th iteration beyond —; a plan whose
 * team has given up once (dfa_solver_team_pcg_info), larger plans, and linear_iter > 256 use a many-workgroup PCG whose
 * launches go out in chunks, and the call then waits for `stream` about once per Gauss-Newton iteration to read the stop
 * flag.  Capturable into a caller's HIP graph: only <= 2048 nodes with <= 8 Gauss-Newton iterations.  The team PCG
 * numbers its barrier rounds by a launch argument and REFUSES a capturing stream (DFA_ERR_HIP: a replayed launch would meet
 * the flag words of the replay before); the many-workgroup form blocks the calling thread (hipStreamSynchronize on
 * `stream`) between its launches, also between two invocations of the overlap callback. */
/* Order-stable variant of the reference-parity solve: the same bits from the same inputs ("or
 * deterministic segmented reduction for bit-stable results").  Both paths sum a node's rows as 64-bit fixed-point integers
 * in LDS (exact, whatever the order of the adds: csrc/solve.hip, FixedScale) — the matrix ENTRIES are the same bits in
 * every run.  What differs from run to run on the default path is their ORDER: the rows are compacted in hash order and
 * rows of equal length are placed by an atomic cursor, so the float sums of the PCG's products (and the gradient's wave
 * sums over an unsorted list) round differently in the last bit (measured in round 3, before the integer sums: up to 2e-5 m
 * in the translations after 24 x 16 iterations; not re-measured since).  With on != 0 the node lists are sorted, the rows
 * are written sorted by column (two passes over a node's rows: ~1.3x the assembly time), and the PCG kernels keep
 * equal-length rows in index order: same inputs, same bits.
 * Takes effect at the next dfa_solver_set_problem.  New plans start with the value of the environment variable
 * DFA_ASSEMBLE_DETERMINISTIC (unset / 0: off). */
int dfa_solver_set_deterministic(dfa_solver* s, int on);

int dfa_solver_solve(dfa_solver* s, const dfa_solve_params* params, dfa_stream_t stream);

/* Results (device pointers, valid until the next set_problem/solve on this plan):
 *   translations : D x 3, the Opt unknown (energy.t:24)
 *   node_dq      : D x 8, DQ(0,0,0,t_i) * dg_se3_i — copyResultToCPUFromFloat3 +
 *                  Node::updateTransformation (opt_solver.cpp:270-285, node.cpp:19-23), once
 *   tukey        : N, huber : D (opt_solver.cpp:204-268) */
const float* dfa_solver_translations(const dfa_solver* s);
const float* dfa_solver_node_dq(const dfa_solver* s);
const float* dfa_solver_tukey_weights(const dfa_solver* s);
const float* dfa_solver_huber_weights(const dfa_solver* s);
const int32_t* dfa_solver_data_graph(const dfa_solver* s); /* N x k */
const int32_t* dfa_solver_reg_graph(const dfa_solver* s);  /* D x k */
/* The normal equations of the LAST Gauss-Newton iteration (for inspection and the reproducibility tests): ELL, slot-major
 * — entry q of row a at [q * D + a] as two floats (value, column as int bits), capacity 256 slots per row; row lengths
 * (D); right-hand side -J^T r (D x 3). */
const float* dfa_solver_matrix_entries(const dfa_solver* s);
const int32_t* dfa_solver_matrix_row_lengths(const dfa_solver* s);
const float* dfa_solver_gradient(const dfa_solver* s);

/* Copies the statistics of the last solve to host memory; synchronises `stream`. */
/* Warpfield::warpToLive (warp_field.cpp:150-171) of the plan's canonical vertices with the SOLVED node transforms
 * (the step DynFusion takes next, dyn_fusion.cpp:196), re-using the plan's k-NN graph of the same vertices instead
 * of searching again: identical output to dfa_warp_to_live(node_pos, dfa_solver_node_dq(s), ...).  normals /
 * out_normals may be NULL. */
int dfa_solver_warp_to_live(dfa_solver* s, const float* normals, float* out_vertices, float* out_normals,
                            dfa_stream_t stream);

int dfa_solver_get_stats(dfa_solver* s, dfa_solve_stats* host_out, dfa_stream_t stream);

/* Optional per-kernel timing for roofline reports (no reference counterpart; Opt has
 * profileSolve, opt_solver.cpp:144-147).  dfa_solver_enable_timing(s, 1) starts a measurement: from then on
 * dfa_solver_solve brackets every assembly launch and every PCG kernel with hipEvents on the launch stream
 * (an event costs ~5 us of stream time: enable = 0 pauses and enable = 2 resumes without discarding, so that
 * a caller can sample some of its solves).
 * dfa_solver_get_timing synchronises the stream and returns the sums over ALL solves since the measurement
 * started: kernel milliseconds, launch counts, solves and PCG iterations. */
typedef struct {
    float pcg_ms;
    float assemble_ms;
    int pcg_launches;
    int assemble_launches;
    long long matrix_nnz; /* non-zeros of the assembled normal matrix (last linearisation) */
    long long pcg_iters;  /* PCG iterations of the measured solves */
    int solves;
    int reserved;
} dfa_solve_timing;
int dfa_solver_enable_timing(dfa_solver* s, int enable);

/* Scheduling hook (no reference counterpart).  Up to 2048 nodes the PCG of a solve runs on three CUs and is most of
 * the solve's time; the kernels around it (linearisation, assembly) are short and chip-wide.  A caller with independent
 * chip-wide work of its own — the TSDF sweep of the same frame, the graph build of the next one — wants it in a PCG's
 * shadow, not beside those short kernels (a 512^3 sweep launched at the start of the solve delays the first
 * linearisation by ~0.1 ms).  `fn` (NULL removes it) is called by every following dfa_solver_solve, on the calling
 * thread, right after the assembly launch of Gauss-Newton iteration `gn_iteration` (0, 1, ... counted over the whole
 * solve) has been enqueued on `solve_stream` and before that iteration's PCG launch; a solve that runs no iteration
 * calls it once with gn_iteration = -1 before its final cost evaluation.  Typical body: record an event on
 * `solve_stream`, let another stream wait for it, enqueue the independent work there.  The callback must not call into
 * the same plan. */
typedef void (*dfa_overlap_fn)(void* user, dfa_stream_t solve_stream, int gn_iteration);
int dfa_solver_set_overlap_callback(dfa_solver* s, dfa_overlap_fn fn, void* user);
int dfa_solver_get_timing(dfa_solver* s, dfa_solve_timing* host_out, dfa_stream_t stream);

/* The PCG of plans with 2 049 .. ~9 300 nodes runs as three teams of persistent workgroups (one coordinate each, each team
 * on one XCD: no kernel boundary per iteration).  A team that cannot assemble, or meets a row of the normal
 * matrix too long for its register slots, gives up before it has changed anything; the guard launch behind it solves that
 * coordinate and the plan takes the launch-per-iteration form from its next PCG on.  This reports what has happened so far
 * (no synchronisation; counts of the calling thread's view): team launches enqueued, teams that gave up, whether the plan
 * has gone back to the launched form (1 also for plans the team form does not serve).  Replaces nothing in the reference:
 * Opt's solver (src/dynfu/utils/opt_solver.cpp:107-147) reports no such thing. */
int dfa_solver_team_pcg_info(dfa_solver* s, int* launches, int* aborts, int* disabled);

/* ===================================================================================== */
/* North-star solver — the 6-DoF mode BASELINE.json:north_star asks for; NOT in the          */
/* reference's code (its energy.t solves translations only).  Formulas: DESIGN.md §4.5.     */
/* Unknown per node: a twist (omega, v) about the node's current position; model: dual-      */
/* quaternion blend of the k nearest nodes with normalised RBF weights; data term: projective */
/* point-to-plane against the live vertex / normal maps (row shape of                       */
/* src/kfusion/cuda/proj_icp.cu:343-350), Tukey-weighted; regulariser: T_n g_m - T_m g_m over */
/* the k nearest other nodes, Huber-weighted, w_reg^2 = lambda/(D k) (opt_solver.cpp:30);     */
/* solver: Gauss-Newton with block-Jacobi (6x6) PCG.                                        */

/* kfusion::cuda::computePointNormals (src/kfusion/imgproc.cpp:27-36, cuda/imgproc.cu:187-226):
 * float4 vertex and normal maps of a depth image, quiet NaN where undefined. */
int dfa_compute_points_normals(const uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                               float cy, float* points, int points_step, float* normals, int normals_step,
                               dfa_stream_t stream);

typedef struct dfa_solver6 dfa_solver6;

typedef struct dfa_solve6_params {
    int num_iter;       /* outer iterations: Tukey / Huber weights recomputed (numIter)        */
    int gn_iter;        /* Gauss-Newton iterations per outer iteration (nonLinearIter)         */
    int linear_iter;    /* PCG iterations per Gauss-Newton iteration, at most (linearIter)     */
    float tukey_offset; /* dyn_fusion.cpp:13 */
    float psi_data;     /* dyn_fusion.cpp:18 */
    float lambda;       /* dyn_fusion.cpp:16 */
    float psi_reg;      /* dyn_fusion.cpp:20 */
    float dist_thresh;  /* association gate |p - l| (metres)                                   */
    float cos_thresh;   /* association gate n_warped . n_live                                  */
    float damping;      /* added to the diagonal of the normal matrix                          */
    float pcg_tol;      /* PCG stops when r.z <= pcg_tol^2 (r.z)_0                             */
    /* Inexact-Newton forcing schedule: Gauss-Newton iteration i (from 0, inside its outer iteration) stops its PCG at
     * the relative residual max(pcg_tol, pcg_tol_first * pcg_tol_decay^i).  pcg_tol_first <= 0: constant pcg_tol. */
    float pcg_tol_first, pcg_tol_decay;
    /* pcg_tol_adapt > 0 (with pcg_tol_first > 0): the Eisenstat-Walker forcing term ("Choosing the forcing terms in an
     * inexact Newton method", 1996, choice 2 with alpha = 2) instead of the geometric schedule.  The first Gauss-Newton
     * iteration of an outer iteration stops its PCG at pcg_tol_first; iteration i > 0 at
     *     clamp(pcg_tol_adapt * (r.z)_0,i / (r.z)_0,i-1, pcg_tol, pcg_tol_first),
     * (r.z)_0 = g^T M^-1 g of the linearisation: a gradient that is still falling fast is followed by a tight solve, one
     * that stagnates (the fit has reached the noise of the depth data) by a loose one.  Decided on the device, by the
     * first step of every PCG; the value used is reported in dfa_solve6_stats.pcg_tol_hist. */
    float pcg_tol_adapt;
    /* A PCG iteration is one kernel launch, and a launch whose PCG has already converged still costs ~3.5 us of stream
     * time.  adaptive_launch != 0: the plan enqueues, for Gauss-Newton iteration i of solve n, only as many launches as
     * iteration i needed in the plan's solves up to n - 2 — a running maximum, raised at once and lowered by one per
     * solve — plus a quarter (at least 2 more), never more than linear_iter.  The counts come from a pinned-memory ring the device writes; solve n folds solves <= n - 2 into
     * its history, in order, each behind its completion event — the budgets are a function of the sequence of solves,
     * NOT of how far the device has got, so a sequence gives the same results in every run whether or not the caller
     * waits between solves.  The first two solves of a plan, and the first two after the stopping rules (iteration
     * counts, tolerances, forcing) change or the problem's node / vertex count moves by more than an eighth, enqueue the
     * full budget.  Results are those of the full budget while the prediction holds (the launches left out are no-ops); a
     * PCG that would have needed more is stopped where its launches end (a truncated PCG: still a descent step), counted
     * in dfa_solve6_stats.pcg_short, and its budget doubles.  0: always linear_iter launches. */
    int adaptive_launch;
    /* Gauss-Newton stopping rule and step acceptance.  The reference runs Opt with earlyOut = true and nonLinearIter as a
     * cap (src/dynfu/dyn_fusion.cpp:183-189, test/opt_optimisation_test.cpp:43); dfa_solve_params.gn_tol is the same switch
     * of the reference-parity solve.  0: every outer iteration runs its gn_iter iterations.  > 0 (relative, below 1): with
     * E_ref the energy at the last accepted linearisation of the outer iteration (weights frozen) and E the energy
     * re-linearised after a step,
     *     E > (1 + gn_tol) E_ref        the step is REJECTED: the transforms before it come back, the outer iteration ends;
     *     E_ref - E <= gn_tol E_ref     CONVERGED: the step stays, the outer iteration ends without another solve;
     *     otherwise                     E_ref = E and the iteration goes on;
     * the last step of the solve, if its outer iteration ran to the cap, is checked by one closing linearisation (first
     * test only).  Decided on the device (one small launch per linearisation); every launch of the solve is enqueued
     * regardless and the ones behind the end of an outer iteration return at entry, so nothing waits for the host.
     * What happened is reported per iteration in dfa_solve6_stats.stop_hist. */
    float gn_tol;
} dfa_solve6_params;

#define DFA_SOLVE6_HIST 32

typedef struct dfa_solve6_stats {
    double initial_cost, final_cost; /* energy at the first / last (gn_tol > 0: last ACCEPTED) linearisation */
    int gn_iters, pcg_iters;         /* linearisations evaluated; PCG iterations */
    int gn_solves, gn_rejected, gn_converged; /* normal equations solved; steps undone; outer iterations ended by convergence */
    int hist_n;                      /* slots of the histories below that are in use */
    long long valid_first, valid_last; /* data rows with a valid association and non-zero weight */
    int max_row_blocks, overflow;
    int pcg_short;    /* PCGs of this solve cut short by the adaptive launch budget (see dfa_solve6_params.adaptive_launch) */
    int pcg_launches; /* PCG step launches the last solve enqueued (converged or not; without adaptive_launch:
                         (linear_iter + 1) per Gauss-Newton iteration) */
    /* per Gauss-Newton iteration (the first DFA_SOLVE6_HIST of the solve): energy at its linearisation, PCG iterations it
     * ran, and the relative residual sqrt(r.z / (r.z)_0) its PCG stopped at (whether by tolerance or by linear_iter) */
    double cost_hist[DFA_SOLVE6_HIST];
    float pcg_rel_hist[DFA_SOLVE6_HIST];
    int pcg_it_hist[DFA_SOLVE6_HIST];
    float pcg_tol_hist[DFA_SOLVE6_HIST]; /* the relative residual every PCG was asked for (the forcing term) */
    /* slot = outer * gn_iter + gn (with gn_tol > 0, slot num_iter * gn_iter is the closing check).  valid_hist: data rows
     * with an association and non-zero weight at that linearisation — cost_hist / valid_hist tells a fit that drifts from
     * one that gains rows.  stop_hist: 0 linearised and solved; 1 converged here; 2 the step before it was rejected here
     * (cost_hist holds the rejected energy); 3 skipped, the outer iteration had ended. */
    long long valid_hist[DFA_SOLVE6_HIST];
    int stop_hist[DFA_SOLVE6_HIST];
} dfa_solve6_stats;

int dfa_solver6_create(int max_D, int max_N, int k /* 1..8 */, dfa_solver6** out);
void dfa_solver6_destroy(dfa_solver6* s);
/* graphs of the frame: k-NN + normalised weights of the canonical vertices, the k nearest other
 * nodes of every node.  The node arrays are borrowed until the next set_problem; the canonical vertices / normals are read
 * by the work this call enqueues only (the plan keeps its own copy, sorted by nearest node — every output comes back in the
 * caller's vertex order). */
int dfa_solver6_set_problem(dfa_solver6* s, const float* node_pos, const float* node_dq, const float* node_w, int D,
                            const float* canon_vertices, const float* canon_normals /* may be NULL */, int N,
                            dfa_stream_t stream);
/* The next frame of a sequence whose canonical cloud and node set have not changed: the graphs of the last
 * dfa_solver6_set_problem stay, only the transforms the solve starts from are replaced (D x 8, borrowed like the arrays of
 * set_problem).  What a per-frame caller does instead of rebuilding k-NN, transposition and pair lists (~0.5 ms at 232 k
 * vertices) when nothing they depend on has moved. */
int dfa_solver6_set_node_transforms(dfa_solver6* s, const float* node_dq);
/* live maps: float4 pixels in the camera frame (the canonical cloud and the nodes are in the same
 * frame), NaN where undefined — dfa_compute_points_normals or dfa_tsdf_raycast_points output */
int dfa_solver6_solve(dfa_solver6* s, const float* live_vertex_map, int vertex_step, const float* live_normal_map,
                      int normal_step, int cols, int rows, float fx, float fy, float cx, float cy,
                      const dfa_solve6_params* params, dfa_stream_t stream);
const float* dfa_solver6_node_dq(const dfa_solver6* s); /* device, D x 8: solved node transforms */
/* canonical vertices (and normals) of the plan warped by the solved transforms (DQ blend) */
int dfa_solver6_warp(dfa_solver6* s, float* out_vertices, float* out_normals, dfa_stream_t stream);
int dfa_solver6_get_stats(dfa_solver6* s, dfa_solve6_stats* out, dfa_stream_t stream);

/* hipEvent timings of the LAST solve, summed over its Gauss-Newton iterations (for bench.py's roofline) */
typedef struct dfa_solve6_timing {
    float linearise_ms, assemble_ms, pcg_ms;
    int gn_iterations;
    long long matrix_blocks; /* 6x6 blocks of the assembled normal matrix */
} dfa_solve6_timing;
int dfa_solver6_enable_timing(dfa_solver6* s, int enable);
int dfa_solver6_get_timing(dfa_solver6* s, dfa_solve6_timing* out, dfa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNFU_AMD_H */
