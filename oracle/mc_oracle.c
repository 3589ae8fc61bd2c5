/* mc_oracle.c — CPU restatement of the reference's marching cubes over the TSDF volume.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: the reference holds no test or golden
 * vector for marching cubes; this file restates src/kfusion/cuda/marching_cubes.cu line by line
 * and is itself the only pin of the HIP kernels.
 *
 * Reference behaviour followed here:
 *   computeCubeIndex   marching_cubes.cu:35-73   (corner order, "any weight == 0 -> no cube", f < iso)
 *   OccupiedVoxels     marching_cubes.cu:75-141  (x + 1 < X, y + 1 < Y, z < Z - 1; numVerts table;
 *                                                 cases 0 and 255 skipped)
 *   getNodeCoo         marching_cubes.cu:183-191 ((i + 0.5) * cell_size)
 *   vertex_interp      marching_cubes.cu:193-199 (t = (iso - f0) / (f1 - f0 + 1e-15f))
 *   TrianglesGenerator marching_cubes.cu:201-253 (12 edge vertices, triTable rows, float4 {x,y,z,1})
 * Differences, both deliberate:
 *   - the reference hard-codes 128^3 (internal.hpp:74, marching_cubes.cu:147,283-285); here the
 *     dimensions are arguments and cell_size is passed in (the reference's value is size / 128);
 *   - the reference's voxel order depends on the order in which warps win an atomicAdd
 *     (marching_cubes.cu:111-114); here voxels are emitted in ascending linear index
 *     z*X*Y + y*X + x, which is one of the orders the reference can produce per voxel group.
 * The case tables are ARGUMENTS, as in kfusion::device::bindTextures (marching_cubes.cu:14-19). */
#include <stddef.h>

#include "oracle.h"

static int cube_index(const uint32_t* vol, int X, int Y, int x, int y, int z, float iso, float f[8]) {
    static const int off[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    for (int c = 0; c < 8; ++c) { /* :37-60 */
        const uint32_t v = vol[(size_t)(x + off[c][0]) + (size_t)X * ((size_t)(y + off[c][1]) + (size_t)Y * (size_t)(z + off[c][2]))];
        if ((v >> 16) == 0) return 0;
        f[c] = orc_half_to_float((uint16_t)(v & 0xffffu));
    }
    int ci = 0; /* :63-71 */
    for (int c = 0; c < 8; ++c) ci += (f[c] < iso) << c;
    return ci;
}

static void interp(const float p0[3], const float p1[3], float f0, float f1, float iso, float out[3]) {
    const float t = (iso - f0) / (f1 - f0 + 1e-15f); /* :195 */
    for (int a = 0; a < 3; ++a) out[a] = p0[a] + t * (p1[a] - p0[a]);
}

/* returns the total number of vertices the volume produces; writes at most max_vertices float4
 * points; *occupied (optional) = number of voxels with at least one triangle */
long orc_marching_cubes(const uint32_t* vol, int X, int Y, int Z, const float cell_size[3], const int32_t* tri_table,
                        const int32_t* num_verts_table, float* out_points, long max_vertices, long* occupied) {
    static const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    static const int edge[12][2]  = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    const float iso = 0.f; /* internal.hpp:72 */
    long total = 0, occ = 0;
    for (int z = 0; z < Z - 1; ++z)
        for (int y = 0; y + 1 < Y; ++y)
            for (int x = 0; x + 1 < X; ++x) {
                float f[8];
                const int ci = cube_index(vol, X, Y, x, y, z, iso, f);
                const int nv = (ci == 0 || ci == 255) ? 0 : num_verts_table[ci]; /* :99 */
                if (nv <= 0) continue;
                ++occ;
                float v[8][3], vl[12][3];
                for (int c = 0; c < 8; ++c) { /* :183-191 */
                    v[c][0] = ((float)(x + corner[c][0]) + 0.5f) * cell_size[0];
                    v[c][1] = ((float)(y + corner[c][1]) + 0.5f) * cell_size[1];
                    v[c][2] = ((float)(z + corner[c][2]) + 0.5f) * cell_size[2];
                }
                for (int e = 0; e < 12; ++e) interp(v[edge[e][0]], v[edge[e][1]], f[edge[e][0]], f[edge[e][1]], iso, vl[e]);
                for (int i = 0; i < nv; ++i) { /* :247-252 */
                    const long idx = total + i;
                    if (idx < max_vertices) {
                        const int e        = tri_table[ci * 16 + i];
                        out_points[4 * idx] = vl[e][0], out_points[4 * idx + 1] = vl[e][1], out_points[4 * idx + 2] = vl[e][2];
                        out_points[4 * idx + 3] = 1.0f;
                    }
                }
                total += nv;
            }
    if (occupied) *occupied = occ;
    return total;
}
