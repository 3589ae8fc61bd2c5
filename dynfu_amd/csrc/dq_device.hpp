// dq_device.hpp — DualQuaternion<float> algebra on the device, storage = 8 floats
// (real w,x,y,z ; dual w,x,y,z), same conventions and operation order as the reference's
// include/dynfu/utils/dual_quaternion.hpp (Hamilton product of boost::math::quaternion).
#pragma once
#include "device_math.hpp"

namespace dfa {

struct Quat {
    float w, x, y, z;
};
struct DQ {
    Quat r, d;
};

__device__ __forceinline__ Quat qmul(Quat a, Quat b) {
    return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Quat qadd(Quat a, Quat b) { return Quat{a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Quat qscale(Quat a, float s) { return Quat{a.w * s, a.x * s, a.y * s, a.z * s}; }

__device__ __forceinline__ DQ dq_identity() { return DQ{Quat{1.f, 0.f, 0.f, 0.f}, Quat{0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ DQ dq_load(const float* p) {
    return DQ{Quat{p[0], p[1], p[2], p[3]}, Quat{p[4], p[5], p[6], p[7]}};
}
__device__ __forceinline__ void dq_store(float* p, DQ q) {
    p[0] = q.r.w, p[1] = q.r.x, p[2] = q.r.y, p[3] = q.r.z;
    p[4] = q.d.w, p[5] = q.d.x, p[6] = q.d.y, p[7] = q.d.z;
}
// dual_quaternion.hpp:120-125 — a scalar scales the dual part only
__device__ __forceinline__ DQ dq_scale(DQ a, float s) { return DQ{a.r, qscale(a.d, s)}; }
// :127-135
__device__ __forceinline__ DQ dq_mul(DQ a, DQ b) { return DQ{qmul(a.r, b.r), qadd(qmul(a.r, b.d), qmul(a.d, b.r))}; }
// :139-144 — real part only
__device__ __forceinline__ DQ dq_normalize(DQ a) {
    const float mag = sqrtf(a.r.w * a.r.w + a.r.x * a.r.x + a.r.y * a.r.y + a.r.z * a.r.z);
    return DQ{qscale(a.r, 1.0f / mag), a.d};
}
// DualQuaternion(0,0,0,tx,ty,tz) (:48-67 with zero angles): real = (1,0,0,0)/|.|^2, dual = 0.5*(0,t)*real
__device__ __forceinline__ DQ dq_from_translation(float tx, float ty, float tz) {
    const Quat real = Quat{1.f, 0.f, 0.f, 0.f};
    const Quat dual = qscale(qmul(Quat{0.f, tx, ty, tz}, real), 0.5f);
    return DQ{real, dual};
}
// :204-215 (transformNormal :217-228 is the same formula)
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 dq_transform(DQ q, f3 v) {
    const float w = q.r.w, d0 = q.d.w;
    const f3 r = mk3(q.r.x, q.r.y, q.r.z), d = mk3(q.d.x, q.d.y, q.d.z);
    const f3 rxv = cross(r, v);
    const f3 t1  = cross(r, mk3(rxv.x + w * v.x, rxv.y + w * v.y, rxv.z + w * v.z));
    const f3 rxd = cross(r, d);
    const f3 t2  = mk3((w * d.x - d0 * r.x) + rxd.x, (w * d.y - d0 * r.y) + rxd.y, (w * d.z - d0 * r.z) + rxd.z);
    return mk3((v.x + 2.f * t1.x) + 2.f * t2.x, (v.y + 2.f * t1.y) + 2.f * t2.y, (v.z + 2.f * t1.z) + 2.f * t2.z);
}

// Node::getTransformationWeight (node.cpp:29-36): double pow/exp, result rounded to float
__device__ __forceinline__ float transformation_weight(f3 g, float dg_w, f3 v) {
    const double dx = (double)(g.x - v.x), dy = (double)(g.y - v.y), dz = (double)(g.z - v.z);
    const double dist_sq = dx * dx + dy * dy + dz * dz;
    const double w       = (double)dg_w;
    return (float)exp(-dist_sq / (2 * (w * w)));
}

}  // namespace dfa
