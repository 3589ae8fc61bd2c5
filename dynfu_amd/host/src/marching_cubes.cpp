// marching_cubes.cpp — kfusion::cuda::MarchingCubes on the dynfu_amd C ABI
// (reference: src/kfusion/marching_cubes.cpp:12-61).
#include <kfusion/cuda/marching_cubes.hpp>

#include <vector>

#include "../../../include/dynfu_amd.h"

namespace kfusion {
namespace cuda {

void MarchingCubes::uploadTables(const int* tri, const int* nverts) {
    tri_dev_.upload(std::vector<int>(tri, tri + 256 * 16));
    nverts_dev_.upload(std::vector<int>(nverts, nverts + 256));
    total_dev_.create(1);
}

MarchingCubes::MarchingCubes() {
    std::vector<int> tri(256 * 16), nv(256);
    dfa::check(dfa_mc_default_tables(tri.data(), nv.data()), "MarchingCubes: default tables");
    uploadTables(tri.data(), nv.data());
}

MarchingCubes::MarchingCubes(const int* triTable, const int* numVertsTable) { uploadTables(triTable, numVertsTable); }

MarchingCubes::~MarchingCubes() = default;

dfa::DeviceArray<MarchingCubes::PointType> MarchingCubes::run(const TsdfVolume& volume,
                                                              dfa::DeviceArray<PointType>& triangles_buffer) {
    if (triangles_buffer.empty()) triangles_buffer.create(DEFAULT_TRIANGLES_BUFFER_SIZE);  // :23-25
    const Vec3i dims = volume.getDims();
    const Vec3f size = volume.getSize();
    // the reference divides by its hard-coded 128 (marching_cubes.cu:283-285) = the voxel size there
    const float cell[3] = {size[0] / dims[0], size[1] / dims[1], size[2] / dims[2]};
    // (with the volume's occupancy map, when the volume knows it: the count sweep skips what the fuse left empty)
    if (const unsigned char* occ = volume.occupancy())
        dfa::check(dfa_marching_cubes_occ(volume.data().ptr<uint32_t>(), occ, dims[0], dims[1], dims[2], cell, tri_dev_.ptr(),
                                          nverts_dev_.ptr(), (float*)triangles_buffer.ptr(), (int)triangles_buffer.size(),
                                          total_dev_.ptr(), nullptr),
                   "MarchingCubes::run");
    else
        dfa::check(dfa_marching_cubes(volume.data().ptr<uint32_t>(), dims[0], dims[1], dims[2], cell, tri_dev_.ptr(),
                                      nverts_dev_.ptr(), (float*)triangles_buffer.ptr(), (int)triangles_buffer.size(),
                                      total_dev_.ptr(), nullptr),
                   "MarchingCubes::run");
    std::vector<int> t;
    total_dev_.download(t);  // synchronises
    last_total_ = t[0];
    const size_t n = (size_t)last_total_ < triangles_buffer.size() ? (size_t)last_total_ : triangles_buffer.size();
    if (n == 0) return dfa::DeviceArray<PointType>();  // :42-46
    return dfa::DeviceArray<PointType>(triangles_buffer.ptr(), n);
}

void MarchingCubes::computeNormals(const TsdfVolume& volume, const dfa::DeviceArray<PointType>& vertices,
                                   dfa::DeviceArray<dfa::Normal>& normals) {
    if (normals.size() < vertices.size()) normals.create(vertices.size());
    if (vertices.empty()) return;
    const Vec3i dims = volume.getDims();
    const Vec3f size = volume.getSize();
    const float voxel[3] = {size[0] / dims[0], size[1] / dims[1], size[2] / dims[2]};
    dfa::check(dfa_tsdf_vertex_normals(volume.data().ptr<uint32_t>(), dims[0], dims[1], dims[2], voxel,
                                       volume.getGradientDeltaFactor(), (const float*)vertices.ptr(),
                                       (int)vertices.size(), (float*)normals.ptr(), nullptr),
               "MarchingCubes::computeNormals");
}

}  // namespace cuda
}  // namespace kfusion
