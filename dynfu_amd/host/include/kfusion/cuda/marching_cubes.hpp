// kfusion/cuda/marching_cubes.hpp — kfusion::cuda::MarchingCubes (reference interface:
// include/kfusion/cuda/marching_cubes.hpp:19-60; behaviour: src/kfusion/marching_cubes.cpp:12-61) on
// dfa_marching_cubes.
//
// What differs from the reference, all visible here:
//   * any volume dimensions (the reference's kernels are fixed to 128^3);
//   * vertices come out in ascending linear voxel order (the reference's order depends on atomics);
//   * ONE host synchronisation per run() — to size the returned array — instead of three;
//   * the case tables are a constructor argument: none = the library's derived tables
//     (dfa_mc_default_tables), or the reference's `triTable` / `numVertsTable`
//     (src/kfusion/marching_cubes.cpp:86-354) for meshes identical to the reference's.
#pragma once
#include <kfusion/cuda/tsdf_volume.hpp>
#include <kfusion/types.hpp>

namespace kfusion {
namespace cuda {

class MarchingCubes {
    dfa::DeviceArray<int> tri_dev_, nverts_dev_, total_dev_;  // 256 x 16, 256, 1
    int last_total_ = 0;
    void uploadTables(const int* tri, const int* nverts);

public:
    typedef dfa::PointXYZ PointType;  // 16 bytes {x, y, z, 1}, the layout of pcl::PointXYZ
    enum { POINTS_PER_TRIANGLE = 3, DEFAULT_TRIANGLES_BUFFER_SIZE = 2 * 1000 * 1000 * POINTS_PER_TRIANGLE };

    MarchingCubes();
    MarchingCubes(const int* triTable /* 256 x 16, -1 padded */, const int* numVertsTable /* 256 */);
    ~MarchingCubes();

    // Extracts the zero level set into `triangles_buffer` (allocated at DEFAULT_TRIANGLES_BUFFER_SIZE when empty,
    // marching_cubes.cpp:23-25) and returns a NON-owning array over the vertices written: three per triangle.
    dfa::DeviceArray<PointType> run(const TsdfVolume& volume, dfa::DeviceArray<PointType>& triangles_buffer);

    // vertices the last run() found; larger than the buffer means only the buffer's worth was written
    int totalVertices() const { return last_total_; }

    // Normals of the extracted vertices from the gradient of the TSDF (the raycaster's compute_normal,
    // tsdf_volume.cu:320-336, with the volume's gradient delta factor).  Extension: the reference leaves the mesh
    // without normals (dyn_fusion.cpp:80-88 "temporary workaround until normals are computed via mc").
    void computeNormals(const TsdfVolume& volume, const dfa::DeviceArray<PointType>& vertices,
                        dfa::DeviceArray<dfa::Normal>& normals);
};

}  // namespace cuda
}  // namespace kfusion
