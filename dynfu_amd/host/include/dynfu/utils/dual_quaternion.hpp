// dynfu/utils/dual_quaternion.hpp — DualQuaternion<T> with the reference's interface and
// conventions (include/dynfu/utils/dual_quaternion.hpp:18-233) on a self-contained quaternion
// type (the reference uses boost::math::quaternion: Hamilton product, R_component_1 = scalar,
// norm() = SQUARED length).  Written from the reference's documented behaviour and its 21
// known-answer tests (tests/cpp/test_host_dq.cpp).
#pragma once
#include <cassert>
#include <cmath>
#include <ostream>

#include <dfa_host/types.hpp>

namespace dfa {
template <class T>
struct quaternion {
    T a, b, c, d;  // a + b i + c j + d k
    quaternion(T a_ = T(), T b_ = T(), T c_ = T(), T d_ = T()) : a(a_), b(b_), c(c_), d(d_) {}
    T R_component_1() const { return a; }
    T R_component_2() const { return b; }
    T R_component_3() const { return c; }
    T R_component_4() const { return d; }
    quaternion operator+(const quaternion& o) const { return quaternion(a + o.a, b + o.b, c + o.c, d + o.d); }
    quaternion operator-(const quaternion& o) const { return quaternion(a - o.a, b - o.b, c - o.c, d - o.d); }
    quaternion operator*(T s) const { return quaternion(a * s, b * s, c * s, d * s); }
    quaternion operator/(T s) const { return quaternion(a / s, b / s, c / s, d / s); }
    quaternion operator*(const quaternion& o) const {
        return quaternion(a * o.a - b * o.b - c * o.c - d * o.d, a * o.b + b * o.a + c * o.d - d * o.c,
                          a * o.c - b * o.d + c * o.a + d * o.b, a * o.d + b * o.c - c * o.b + d * o.a);
    }
    quaternion& operator+=(const quaternion& o) { return *this = *this + o; }
    quaternion& operator-=(const quaternion& o) { return *this = *this - o; }
    quaternion& operator*=(T s) { return *this = *this * s; }
    quaternion& operator*=(const quaternion& o) { return *this = *this * o; }
};
template <class T>
quaternion<T> conj(const quaternion<T>& q) {
    return quaternion<T>(q.a, -q.b, -q.c, -q.d);
}
template <class T>
T norm(const quaternion<T>& q) {  // Cayley norm, like boost::math::norm
    return q.a * q.a + q.b * q.b + q.c * q.c + q.d * q.d;
}
template <class T>
std::ostream& operator<<(std::ostream& os, const quaternion<T>& q) {
    return os << "(" << q.a << "," << q.b << "," << q.c << "," << q.d << ")";
}
}  // namespace dfa

template <class T>
class DualQuaternion {
private:
    dfa::quaternion<T> real;  // rotation
    dfa::quaternion<T> dual;  // displacement

    static dfa::quaternion<T> normalize(const dfa::quaternion<T>& q) { return q / dfa::norm(q); }  // :32
    static constexpr float epsilon = 1.192092896e-07f;

public:
    DualQuaternion(dfa::quaternion<T> rotation, dfa::quaternion<T> translation) : real(rotation), dual(translation) {}

    DualQuaternion(dfa::quaternion<T> rotation, dfa::Vec3f translation) {  // :42-45
        real = normalize(rotation);
        dual = (dfa::quaternion<T>(0, translation[0], translation[1], translation[2]) * real) * T(0.5);
    }

    DualQuaternion(T yaw, T pitch, T roll, T x, T y, T z) {  // :48-67
        // (cos 0 = 1 and sin 0 = 0 exactly: the pure translations the solver writes back — one per node and frame,
        // opt_solver.cpp:281 — skip the six libm calls and give the same bits)
        const bool no_rotation = yaw == T(0) && pitch == T(0) && roll == T(0);
        T cy = no_rotation ? T(1) : (T)std::cos(yaw * 0.5), sy = no_rotation ? T(0) : (T)std::sin(yaw * 0.5);
        T cr = no_rotation ? T(1) : (T)std::cos(roll * 0.5), sr = no_rotation ? T(0) : (T)std::sin(roll * 0.5);
        T cp = no_rotation ? T(1) : (T)std::cos(pitch * 0.5), sp = no_rotation ? T(0) : (T)std::sin(pitch * 0.5);
        dfa::quaternion<T> rotation(cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp,
                                    cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp);
        DualQuaternion<T> dq(rotation, dfa::Vec3f(x, y, z));
        real = dq.getReal();
        dual = dq.getDual();
    }

    DualQuaternion(dfa::Vec3f rodrigues, dfa::Vec3f translation) {  // :70-86
        const double nrm = std::sqrt((double)rodrigues[0] * rodrigues[0] + (double)rodrigues[1] * rodrigues[1] +
                                     (double)rodrigues[2] * rodrigues[2]);
        const double theta = 2 * std::atan(nrm);
        float ax[3] = {(float)(rodrigues[0] / theta), (float)(rodrigues[1] / theta), (float)(rodrigues[2] / theta)};
        const double an = std::sqrt((double)ax[0] * ax[0] + (double)ax[1] * ax[1] + (double)ax[2] * ax[2]);
        const double s  = std::sin(0.5 * theta);
        dfa::quaternion<T> rotation((T)std::cos(0.5 * theta), (T)(s * (float)(ax[0] / an)),
                                    (T)(s * (float)(ax[1] / an)), (T)(s * (float)(ax[2] / an)));
        DualQuaternion<T> dq(normalize(rotation), translation);
        real = dq.getReal();
        dual = dq.getDual();
    }

    dfa::quaternion<T> getReal() const { return real; }
    dfa::quaternion<T> getDual() const { return dual; }
    dfa::quaternion<T> getRotation() const { return real; }

    dfa::Vec3f getTranslation() const {  // :94-97
        dfa::quaternion<T> q = (dual * T(2)) * dfa::conj(real);
        return dfa::Vec3f(q.R_component_2(), q.R_component_3(), q.R_component_4());
    }

    DualQuaternion<T> operator+(const DualQuaternion<T>& o) const { return DualQuaternion<T>(real + o.real, dual + o.dual); }
    DualQuaternion<T>& operator+=(const DualQuaternion<T>& o) {
        real += o.real, dual += o.dual;
        return *this;
    }
    DualQuaternion<T> operator-(const DualQuaternion<T>& o) const { return DualQuaternion<T>(real - o.real, dual - o.dual); }
    DualQuaternion<T>& operator-=(const DualQuaternion<T>& o) {
        real -= o.real, dual -= o.dual;
        return *this;
    }
    // a scalar scales the DUAL part only (:120-125)
    DualQuaternion<T> operator*(T scale) const { return DualQuaternion<T>(real, dual * scale); }
    DualQuaternion<T>& operator*=(T scale) {
        dual *= scale;
        return *this;
    }
    DualQuaternion<T> operator*(const DualQuaternion<T>& o) const {  // :127-129
        return DualQuaternion<T>(real * o.real, real * o.dual + dual * o.real);
    }
    // the reference's operator*= has no return statement (:131-135, UB if used as a value);
    // the adaptor returns *this
    DualQuaternion<T>& operator*=(const DualQuaternion<T>& o) {
        dual = real * o.dual + dual * o.real;
        real *= o.real;
        return *this;
    }
    DualQuaternion<T> conj() const { return DualQuaternion<T>(dfa::conj(real), dfa::conj(dual)); }

    DualQuaternion<T>& normalize() {  // :139-144: real part only
        T magnitude = std::sqrt(real.a * real.a + real.b * real.b + real.c * real.c + real.d * real.d);
        assert(magnitude > epsilon);
        real *= (T(1) / magnitude);
        return *this;
    }

    T getRoll() const {  // :148-160
        float sinr = (float)(+2.0 * (real.a * real.b + real.c * real.d));
        float cosr = (float)(+1.0 - 2.0 * (real.b * real.b + real.c * real.c));
        T roll     = (T)std::atan2(sinr, cosr);
        if (roll > M_PI) roll -= (T)M_PI_2;
        return roll;
    }
    T getPitch() const {  // :162-176
        float sinp = (float)(+2.0 * (real.a * real.c - real.d * real.b));
        if (std::fabs(sinp) >= 1) return (T)std::copysign(M_PI / 2, sinp);
        return (T)std::asin(sinp);
    }
    T getYaw() const {  // :178-190
        float siny = (float)(+2.0 * (real.a * real.d + real.b * real.c));
        float cosy = (float)(+1.0 - 2.0 * (real.c * real.c + real.d * real.d));
        T yaw      = (T)std::atan2(siny, cosy);
        if (yaw > M_PI) yaw -= (T)M_PI_2;
        return yaw;
    }
    dfa::Vec3f getEulerAngles() const { return dfa::Vec3f(getRoll(), getPitch(), getYaw()); }

    dfa::Vec3f getRodrigues() const {  // :194-200
        const double nrm   = std::sqrt((double)real.b * real.b + (double)real.c * real.c + (double)real.d * real.d);
        const double theta = 2 * std::acos(real.a);
        const double tn    = std::tan(0.5 * theta);
        return dfa::Vec3f((float)((float)(tn * real.b) / nrm), (float)((float)(tn * real.c) / nrm),
                          (float)((float)(tn * real.d) / nrm));
    }

    dfa::PointXYZ transformVertex(dfa::PointXYZ v) const {  // :204-215
        float o[3];
        transform(v.x, v.y, v.z, o);
        return dfa::PointXYZ(o[0], o[1], o[2]);
    }
    dfa::Normal transformNormal(dfa::Normal n) const {  // :217-228, the same formula (translation included)
        float o[3];
        transform(n.normal_x, n.normal_y, n.normal_z, o);
        return dfa::Normal(o[0], o[1], o[2]);
    }

    friend std::ostream& operator<<(std::ostream& os, const DualQuaternion<T>& dq) {
        return os << "real: " << dq.getReal() << "\ndual: " << dq.getDual() << std::endl;
    }

private:
    void transform(float x, float y, float z, float o[3]) const {
        const float w = real.a, d0 = dual.a;
        const float r[3] = {real.b, real.c, real.d}, d[3] = {dual.b, dual.c, dual.d}, v[3] = {x, y, z};
        auto cross = [](const float* a, const float* b, float* c) {
            c[0] = a[1] * b[2] - a[2] * b[1], c[1] = a[2] * b[0] - a[0] * b[2], c[2] = a[0] * b[1] - a[1] * b[0];
        };
        float rxv[3], tmp[3], t1[3], rxd[3];
        cross(r, v, rxv);
        for (int i = 0; i < 3; ++i) tmp[i] = rxv[i] + w * v[i];
        cross(r, tmp, t1);
        cross(r, d, rxd);
        for (int i = 0; i < 3; ++i) o[i] = (v[i] + 2.f * t1[i]) + 2.f * ((w * d[i] - d0 * r[i]) + rxd[i]);
    }
};
