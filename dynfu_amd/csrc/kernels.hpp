// kernels.hpp — host-callable launchers of the gfx950 kernels (internal; the public surface
// is include/dynfu_amd.h).
#pragma once
#include <map>
#include <mutex>
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfa {

// tsdf.hip
hipError_t launch_compute_dists(const uint16_t* depth, int depth_step, uint16_t* dists, int dists_step, int cols,
                                int rows, float fx, float fy, float cx, float cy, hipStream_t s);
hipError_t launch_tsdf_clear(uint32_t* vol, int X, int Y, int Z, hipStream_t s);
// Occupancy map of a volume (optional; include/dynfu_amd.h: dfa_tsdf_occupancy_bytes): one byte per box of 32 x 2 x 8 voxels
// — the patch of columns a wave of the sweep owns times one classified run —: bit 0 if a sweep may have left a voxel with a
// non-zero weight in it, bit 1 if one may have left a NEGATIVE distance there (a run classified FULL: FRONT runs write +1).
// Byte ((z / 8) oy + y / 2) ox + x / 32.
struct OccDims {
    int ox, oy, oz;
    __host__ __device__ size_t bytes() const { return (size_t)ox * oy * oz; }
};
__host__ __device__ inline OccDims occ_dims(int X, int Y, int Z) { return OccDims{(X + 31) / 32, (Y + 1) / 2, (Z + 7) / 8}; }
// occ (may be null): fused_clear writes every byte of it (the sweep writes every voxel), the accumulating sweep only sets bytes
hipError_t launch_tsdf_integrate(bool fused_clear, const uint16_t* dists, int dists_step, int cols, int rows,
                                 uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                                 int max_weight, const float vol2cam[12], float fx, float fy, float cx, float cy,
                                 uint8_t* occ, bool occ_known /* fused: the map describes the volume on entry */, hipStream_t s);
hipError_t launch_vertex_normals(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float delta_factor,
                                 const float* points, int n, float* normals, hipStream_t s);
hipError_t launch_raycast_points(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3],
                                 float trunc_dist, const float cam2vol[12], const float Rinv[9], float fx, float fy,
                                 float cx, float cy, float step_factor, float delta_factor, float* points,
                                 int points_step, float* normals, int normals_step, int cols, int rows,
                                 hipStream_t s);
hipError_t launch_raycast_tally(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                                const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                                float step_factor, float delta_factor, int cols, int rows, unsigned long long* counts,
                                uint32_t* touched, hipStream_t s);
hipError_t launch_raycast_depth(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                                const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                                float step_factor, float delta_factor, uint16_t* depth, int depth_step,
                                float* normals, int normals_step, int cols, int rows, hipStream_t s);

// warp.hip
// Uniform grid over the deformation nodes (device-resident, geometry computed on the device).
constexpr int KNN_GRID_MAX_DIM   = 32;
constexpr int KNN_GRID_MAX_CELLS = KNN_GRID_MAX_DIM * KNN_GRID_MAX_DIM * KNN_GRID_MAX_DIM;
struct KnnGridDesc {
    float bmin[3];
    float cs, inv_cs;
    int dim[3];
};
struct KnnGridView {
    KnnGridDesc* desc;
    int32_t* cell_count;  // KNN_GRID_MAX_CELLS   (counts, then fill cursors)
    int32_t* cell_start;  // KNN_GRID_MAX_CELLS + 1
    int32_t* node_cell;   // D
    float4* sorted;       // D   (x, y, z, node index bits), grouped by cell
};
// The same structure at up to 256^3 cells for large point sets (dfa_correspond's canonical cloud)
constexpr int PGRID_MAX_DIM     = 256;   // (128 up to half a million points: see pgrid_finalize_kernel)
constexpr int PGRID_MAX_CELLS   = PGRID_MAX_DIM * PGRID_MAX_DIM * PGRID_MAX_DIM;
constexpr int PGRID_CHUNK       = 8192;  // cells per scan workgroup -> up to 2048 chunks
constexpr int PGRID_BBOX_BLOCKS = 512;
struct PointGridView {
    KnnGridView g;           // cell_count: PGRID_MAX_CELLS, cell_start: PGRID_MAX_CELLS + 1
    int32_t* chunk_sums;     // PGRID_MAX_CELLS / PGRID_CHUNK
    float* bbox_partials;    // PGRID_BBOX_BLOCKS x 6
};
hipError_t point_grid_build(const PointGridView& pg, const float* pts, int n, hipStream_t s);
hipError_t knn_grid_build(const KnnGridView& g, const float* node_pos, int D, hipStream_t s);
// grid == nullptr: exhaustive scan
hipError_t launch_knn(const float* node_pos, const float* node_w, int D, const float* query, int n_query, int k,
                      int32_t* idx, float* weights, const KnnGridView* grid, hipStream_t s);
hipError_t launch_warp_to_live(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                               const float* verts, const float* normals, int N, float* out_verts, float* out_normals,
                               const KnnGridView* grid, hipStream_t s);

hipError_t launch_correspond_projective(const float* verts, const float* normals, int n, const float* vmap, int vmap_step,
                                        const float* nmap, int nmap_step, int cols, int rows, float fx, float fy, float cx,
                                        float cy, float dist_thres, float min_cosine, float* out_v, float* out_n,
                                        int32_t* out_pixel, hipStream_t s);
hipError_t launch_correspond(const float* canon_v, const float* canon_n, int n_canon, const float* live_v,
                             int n_live, float* out_v, float* out_n, int32_t* out_idx, const KnnGridView* grid,
                             hipStream_t s);

hipError_t launch_dqb_support(const float* node_pos, const float* node_dq, const float* node_w, int D, int k,
                              const float* pts, int n, float* out_dq, uint8_t* out_flag, const KnnGridView* grid,
                              hipStream_t s);

hipError_t launch_warp_graph(const float* node_pos, const float* node_dq, const float* node_w, int k, const int32_t* idx,
                             const float* verts, const float* normals, int N, float* out_verts, float* out_normals,
                             hipStream_t s);

// points.hip
hipError_t launch_repack_points(const float* src, int sstride, float* dst, int dstride, int n, float pad, hipStream_t s);
hipError_t launch_transform_points(const float* in, int n, const float aff[12], bool with_translation, float* out, hipStream_t s);
int compact_chunks(int n);  // entries of chunk_scratch
hipError_t launch_compact_points(const float* pts, const uint8_t* flags, int n, float* out_pts, int32_t* out_idx,
                                 int32_t* count, int32_t* chunk_scratch, hipStream_t s);

// mc.hip
long mc_segments(int X, int Y, int Z, bool vec4);  // entries of seg_off (+1)
long mc_scan_chunks(long nsegs);                   // entries of chunk_sums
hipError_t launch_marching_cubes(const uint32_t* vol, int X, int Y, int Z, const float cell_size[3],
                                 const int32_t* tri_table, const int32_t* num_verts_table, float* out_points,
                                 int max_vertices, int32_t* total_vertices, int32_t* seg_off, int32_t* chunk_sums,
                                 const uint8_t* occ /* occupancy map of the volume or null */, hipStream_t s);
void mc_default_tables(int32_t tri_table[256 * 16], int32_t num_verts_table[256]);

// img.hip
hipError_t launch_bilateral(const uint16_t* src, int src_step, uint16_t* dst, int dst_step, int cols, int rows, int ksz,
                            float sigma_spatial, float sigma_depth, hipStream_t s);
hipError_t launch_truncate_depth(uint16_t* depth, int step, int cols, int rows, float max_dist, hipStream_t s);
hipError_t launch_depth_pyr(const uint16_t* src, int src_step, int cols, int rows, uint16_t* dst, int dst_step,
                            float sigma_depth, hipStream_t s);
hipError_t launch_normals_mask_depth(uint16_t* depth, int depth_step, int cols, int rows, float fx, float fy, float cx,
                                     float cy, float* normals, int normals_step, hipStream_t s);
hipError_t launch_resize_depth_normals(const uint16_t* dsrc, int dsrc_step, const float* nsrc, int nsrc_step, int cols,
                                       int rows, uint16_t* ddst, int ddst_step, float* ndst, int ndst_step, hipStream_t s);
hipError_t launch_resize_points_normals(const float* vsrc, int vsrc_step, const float* nsrc, int nsrc_step, int cols,
                                        int rows, float* vdst, int vdst_step, float* ndst, int ndst_step, hipStream_t s);

// icp.hip
size_t icp_partial_floats(int cols, int rows);
hipError_t launch_icp_sums(bool depth_variant, const void* curr, int curr_step, const float* ncurr, int ncurr_step,
                           const void* prev, int prev_step, const float* nprev, int nprev_step, int cols, int rows,
                           const float aff[12], float fx, float fy, float cx, float cy, float dist_thres, float angle_thres,
                           float* partial, float* sums27, unsigned int* matched, hipStream_t s);


// Several kernels publish partial results with write-through stores and then arrive at a ticket / raise a flag word, ordering
// the two by `s_waitcnt vmcnt(0)` (solve.hip: linearise tail, team PCG; solve6.hip: linearise tail).  That is an ordering
// on gfx9 parts only, where stores count in vmcnt; gfx10 and later track them in vscnt.  This library is built for gfx950.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "dynfu_amd orders published stores with s_waitcnt vmcnt(0): gfx90a / gfx942 / gfx950 only"
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is an attribute of a kernel ON A DEVICE: the opt-in to more than 48 KiB of
// dynamic LDS is made once per (device, kernel), under a lock (the C ABI serves several devices and host threads)
inline hipError_t allow_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> done;
    int dev      = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    int& have = done[std::make_pair(dev, kernel)];
    if (have >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

}  // namespace dfa
