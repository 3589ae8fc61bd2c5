// kfusion/cuda/marching_cubes.hpp — class kfusion::cuda::MarchingCubes with the reference's
// interface (include/kfusion/cuda/marching_cubes.hpp:19-60, src/kfusion/marching_cubes.cpp:12-61)
// on dfa_marching_cubes.  Differences: any volume dimensions (the reference's kernels are fixed to
// 128^3); the vertices come out in ascending linear voxel order (the reference's order depends on
// atomics); one host synchronisation per run() (to size the returned array) instead of three.
#pragma once
#include <kfusion/cuda/tsdf_volume.hpp>
#include <kfusion/types.hpp>

namespace kfusion {
namespace cuda {
class MarchingCubes {
public:
    enum { POINTS_PER_TRIANGLE = 3, DEFAULT_TRIANGLES_BUFFER_SIZE = 2 * 1000 * 1000 * POINTS_PER_TRIANGLE };
    typedef dfa::PointXYZ PointType;  // 16 bytes {x, y, z, 1}, as pcl::PointXYZ

    // the library's derived case tables (dfa_mc_default_tables)
    MarchingCubes();
    // caller-supplied tables: pass the reference's `triTable` / `numVertsTable`
    // (src/kfusion/marching_cubes.cpp:86-354) for meshes identical to the reference's
    MarchingCubes(const int* triTable /* 256 x 16 */, const int* numVertsTable /* 256 */);
    ~MarchingCubes();

    // marching_cubes.cpp:20-61: allocates triangles_buffer at its default size when empty; returns a
    // (non-owning) array over the first total-vertices points of triangles_buffer
    dfa::DeviceArray<PointType> run(const TsdfVolume& volume, dfa::DeviceArray<PointType>& triangles_buffer);

    // vertices the last run() found (may exceed the buffer: then only the buffer's worth was written)
    int totalVertices() const { return total_; }

private:
    dfa::DeviceArray<int> triTable_, numVertsTable_, total_dev_;
    int total_ = 0;
};
}  // namespace cuda
}  // namespace kfusion
