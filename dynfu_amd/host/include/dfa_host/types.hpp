// types.hpp — plain value types the host adaptor classes are written against.
//
// The reference's public interfaces speak OpenCV (cv::Vec3f, cv::Affine3f), PCL
// (pcl::PointXYZ, pcl::Normal, pcl::PointCloud) and its own ref-counted CUDA containers
// (kfusion::cuda::DeviceArray2D, device_array.hpp).  None of those libraries exist in this
// image, so the adaptors use these layout-compatible minimal types; with OpenCV / PCL present
// the mapping is a reinterpret of the same floats.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace dfa {

struct Vec3f {
    float v[3];
    Vec3f() : v{0.f, 0.f, 0.f} {}
    Vec3f(float x, float y, float z) : v{x, y, z} {}
    static Vec3f all(float a) { return Vec3f(a, a, a); }
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
};
struct Vec3i {
    int v[3];
    Vec3i() : v{0, 0, 0} {}
    Vec3i(int x, int y, int z) : v{x, y, z} {}
    static Vec3i all(int a) { return Vec3i(a, a, a); }
    int& operator[](int i) { return v[i]; }
    int operator[](int i) const { return v[i]; }
};

// rigid / affine transform, the subset of cv::Affine3f the hot path uses
struct Affine3f {
    float R[9];  // row-major
    float t[3];
    Affine3f() : R{1, 0, 0, 0, 1, 0, 0, 0, 1}, t{0, 0, 0} {}
    static Affine3f Identity() { return Affine3f(); }
    Affine3f translate(const Vec3f& d) const {
        Affine3f a = *this;
        for (int i = 0; i < 3; ++i) a.t[i] += d[i];
        return a;
    }
    Vec3f translation() const { return Vec3f(t[0], t[1], t[2]); }
    // this * o  (apply o first)
    Affine3f operator*(const Affine3f& o) const {
        Affine3f r;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j)
                r.R[3 * i + j] = R[3 * i] * o.R[j] + R[3 * i + 1] * o.R[3 + j] + R[3 * i + 2] * o.R[6 + j];
            r.t[i] = R[3 * i] * o.t[0] + R[3 * i + 1] * o.t[1] + R[3 * i + 2] * o.t[2] + t[i];
        }
        return r;
    }
    // general 3x3 inverse (the reference uses cv's SVD inverse, tsdf_volume.cpp:101; identical for
    // the well-conditioned rotations it is applied to)
    void inverse_rotation(float out[9]) const {
        const float* m = R;
        const double det = (double)m[0] * (m[4] * m[8] - m[5] * m[7]) - (double)m[1] * (m[3] * m[8] - m[5] * m[6]) +
                           (double)m[2] * (m[3] * m[7] - m[4] * m[6]);
        if (det == 0.0) throw std::runtime_error("Affine3f: singular rotation");
        const double id = 1.0 / det;
        out[0] = (float)((m[4] * m[8] - m[5] * m[7]) * id), out[1] = (float)((m[2] * m[7] - m[1] * m[8]) * id);
        out[2] = (float)((m[1] * m[5] - m[2] * m[4]) * id), out[3] = (float)((m[5] * m[6] - m[3] * m[8]) * id);
        out[4] = (float)((m[0] * m[8] - m[2] * m[6]) * id), out[5] = (float)((m[2] * m[3] - m[0] * m[5]) * id);
        out[6] = (float)((m[3] * m[7] - m[4] * m[6]) * id), out[7] = (float)((m[1] * m[6] - m[0] * m[7]) * id);
        out[8] = (float)((m[0] * m[4] - m[1] * m[3]) * id);
    }
    Affine3f inv() const {
        Affine3f r;
        inverse_rotation(r.R);
        for (int i = 0; i < 3; ++i) r.t[i] = -(r.R[3 * i] * t[0] + r.R[3 * i + 1] * t[1] + r.R[3 * i + 2] * t[2]);
        return r;
    }
    // 12 floats of the C ABI: R row-major then t
    void to12(float out[12]) const {
        for (int i = 0; i < 9; ++i) out[i] = R[i];
        for (int i = 0; i < 3; ++i) out[9 + i] = t[i];
    }
};

// kfusion::Intr (include/kfusion/types.hpp:17-23)
struct Intr {
    float fx, fy, cx, cy;
    Intr() : fx(0), fy(0), cx(0), cy(0) {}
    Intr(float fx_, float fy_, float cx_, float cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
    // intrinsics of pyramid level `level` (src/kfusion/core.cpp: every component divided by 2^level)
    Intr operator()(int level) const {
        const float div = (float)(1 << level);
        return Intr(fx / div, fy / div, cx / div, cy / div);
    }
};

// pcl::PointXYZ / pcl::Normal / pcl::PointCloud stand-ins (xyz + pad, 16 bytes like PCL's)
struct PointXYZ {
    float x, y, z, pad;
    PointXYZ() : x(0), y(0), z(0), pad(1.f) {}
    PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_), pad(1.f) {}
};
struct Normal {
    union {
        float data_c[4];
        struct {
            float normal_x, normal_y, normal_z, curvature;
        };
    };
    Normal() : data_c{0, 0, 0, 0} {}
    Normal(float x, float y, float z) : data_c{x, y, z, 0} {}
};
template <class T>
struct PointCloud {
    std::vector<T> points;
    void push_back(const T& p) { points.push_back(p); }
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void clear() { points.clear(); }
    T& operator[](size_t i) { return points[i]; }
    const T& operator[](size_t i) const { return points[i]; }
    typename std::vector<T>::iterator begin() { return points.begin(); }
    typename std::vector<T>::iterator end() { return points.end(); }
    typename std::vector<T>::const_iterator begin() const { return points.begin(); }
    typename std::vector<T>::const_iterator end() const { return points.end(); }
};

// Thrown where the reference's cudaSafeCall prints "KinFu2 error" and exit(0)s
// (include/kfusion/safe_call.hpp:11-22): the message is dfa_last_error().
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
void check(int rc, const char* where);

// pcl::VoxelGrid<pcl::PointXYZ> with default settings and a cubic leaf, as Warpfield::update uses it
// (warp_field.cpp:68-72): one centroid per occupied leaf, in ascending leaf index.  Restates PCL's published
// applyFilter (PCL is not available to this build); points of a leaf are summed in input order.
PointCloud<PointXYZ> voxelGridFilter(const PointCloud<PointXYZ>& cloud, float leaf);  // throws dfa::Error when rc != DFA_OK

}  // namespace dfa
