// dfa_host/io.hpp — the data formats either side of the per-frame path (last paragraph), as the
// reference's demo application reads and writes them (src/apps/demo.cpp):
//   in   <dir>/depth/*.png   16-bit greyscale, millimetres, lexicographic order   (demo.cpp:39-55, :81  cv::imread ANYDEPTH)
//        <dir>/color/*.png   listed and counted, never used by the pipeline        (demo.cpp:82, dyn_fusion.cpp:48)
//   out  <dir>/out/pcl_canonical_to_live<i>.pcd   ASCII PCD of XYZ points          (demo.cpp:21-31  pcl::io::savePCDFileASCII)
//        <dir>/out/<i>_tsdf_mesh.vtk              legacy-VTK polydata of the mesh   (demo.cpp:33-37  pcl::io::saveVTKFile)
// Host code only (no GPU, no OpenCV / PCL / Boost: they do not exist in this image).  The PNG decoder handles what a
// depth frame is — non-interlaced greyscale, 16 bits (8 bits accepted and widened) — over zlib's inflate, with every
// chunk CRC checked.  The writers restate PCL's published ASCII writers (PCL is un-vendored and no version is pinned in
// the reference: UNPINNED, see tests/cpp/test_host_io.cpp for what is checked).
// Not here: the demo's command line, windows and key handling (out of scope).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include <dfa_host/types.hpp>

namespace dfa {

// pcl::PolygonMesh as the reference builds it (KinFu::convertToMesh, kinfu.cpp:236-260): the triangle soup of marching
// cubes as a cloud, polygon i = vertices (3i, 3i + 2, 3i + 1)
struct PolygonMesh {
    PointCloud<PointXYZ> cloud;
    std::vector<std::vector<uint32_t>> polygons;
};
PolygonMesh convertToMesh(const std::vector<PointXYZ>& triangles);

namespace io {

struct DepthImage {
    int cols = 0, rows = 0;
    std::vector<uint16_t> data;  // row-major, dense
};

// throws dfa::Error (DFA_ERR_INVALID = 1) with the reason on any malformed / unsupported file
DepthImage readDepthPng(const std::string& path);
DepthImage decodeDepthPng(const uint8_t* bytes, size_t size);
// the inverse, for tests and for writing synthetic sequences: 16-bit greyscale, filter 0, one IDAT
std::vector<uint8_t> encodeDepthPng(const uint16_t* pixels, int cols, int rows);
void writeDepthPng(const std::string& path, const uint16_t* pixels, int cols, int rows);

// files of <dir>/depth and <dir>/color, each sorted lexicographically (cv::glob + std::sort, demo.cpp:39-55);
// throws if <dir>, <dir>/depth or <dir>/color is missing (the demo exits there)
struct SequenceFiles {
    std::vector<std::string> depths, images;
};
SequenceFiles listSequence(const std::string& dir);

// pcl::io::savePCDFileASCII of a PointXYZ cloud (PCD v0.7 header, one "x y z" line per point, 8 significant digits,
// "nan" for NaN) and pcl::io::saveVTKFile of a PolygonMesh (legacy VTK 3.0 POLYDATA: POINTS, VERTICES, POLYGONS, 5
// significant digits); the *_string forms return the file's text
std::string pcdAsciiString(const PointCloud<PointXYZ>& cloud);
std::string vtkMeshString(const PolygonMesh& mesh);
void savePCDFileASCII(const std::string& path, const PointCloud<PointXYZ>& cloud);
void saveVTKFile(const std::string& path, const PolygonMesh& mesh);

}  // namespace io
}  // namespace dfa
