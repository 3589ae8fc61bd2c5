// kfusion/types.hpp — the reference's include/kfusion/types.hpp names on the adaptor's types.
#pragma once
#include <dfa_host/device.hpp>
#include <dfa_host/types.hpp>

namespace kfusion {
typedef dfa::Vec3f Vec3f;
typedef dfa::Vec3i Vec3i;
typedef dfa::Affine3f Affine3f;
typedef dfa::Intr Intr;
struct Point {  // include/kfusion/types.hpp:27-34
    float x, y, z, w;
};
typedef Point Normal;
namespace cuda {
typedef dfa::DeviceMemory CudaData;                // :51
typedef dfa::DeviceArray2D<unsigned short> Depth;  // :52
typedef dfa::DeviceArray2D<unsigned short> Dists;  // :53
typedef dfa::DeviceArray2D<Normal> Normals;        // :55
typedef dfa::DeviceArray2D<Point> Cloud;           // :56
// cuda::computeDists (src/kfusion/imgproc.cpp:38-41)
void computeDists(const Depth& depth, Dists& dists, const Intr& intr);
// cuda::waitAllDefaultStream
inline void waitAllDefaultStream() { dfa::device_synchronize(); }
}  // namespace cuda
}  // namespace kfusion
