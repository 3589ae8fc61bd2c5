/*
 * dq_oracle.c — CPU restatement of DualQuaternion<float>
 * (include/dynfu/utils/dual_quaternion.hpp) on plain float[8] = real(w,x,y,z), dual(w,x,y,z).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Pinned by the 21 known-answer tests of
 * test/quaternion_test.cpp (tests/test_oracle_dq.py, tests/golden/dq_kat.json).
 *
 * boost::math::quaternion<float> semantics used by the reference:
 *   R_component_1 = scalar; operator* = Hamilton product; norm(q) = sum of squares
 *   (boost's "norm" is the Cayley norm, i.e. |q|^2 — so the reference's private
 *   normalize(q) = q / boost::math::norm(q) divides by the SQUARED length; it is a no-op
 *   for the unit quaternions the reference feeds it, and restated faithfully here).
 */
#include <math.h>
#include <string.h>

#include "oracle.h"

typedef struct {
    float w, x, y, z;
} quat;

static inline quat qmk(float w, float x, float y, float z) {
    quat q = {w, x, y, z};
    return q;
}
static inline quat qload(const float* p) { return qmk(p[0], p[1], p[2], p[3]); }
static inline void qstore(float* p, quat q) { p[0] = q.w, p[1] = q.x, p[2] = q.y, p[3] = q.z; }

/* boost/math/quaternion.hpp operator*= (Hamilton product) */
static inline quat qmul(quat a, quat b) {
    return qmk(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
               a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w);
}
static inline quat qadd(quat a, quat b) { return qmk(a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z); }
static inline quat qsub(quat a, quat b) { return qmk(a.w - b.w, a.x - b.x, a.y - b.y, a.z - b.z); }
static inline quat qscale(quat a, float s) { return qmk(a.w * s, a.x * s, a.y * s, a.z * s); }
static inline quat qconj(quat a) { return qmk(a.w, -a.x, -a.y, -a.z); }
/* boost::math::norm(quaternion) = Cayley norm = w^2+x^2+y^2+z^2 */
static inline float qnorm_boost(quat a) { return a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z; }

/* dual_quaternion.hpp:32 private normalize(q) = q / boost::math::norm(q) */
static inline quat qnormalize_ref(quat q) {
    float n = qnorm_boost(q);
    return qmk(q.w / n, q.x / n, q.y / n, q.z / n);
}

/* :42-45 */
void orc_dq_from_quat_trans(const float q[4], const float t[3], float out[8]) {
    quat real = qnormalize_ref(qload(q));
    quat dual = qscale(qmul(qmk(0.f, t[0], t[1], t[2]), real), 0.5f);
    qstore(out, real);
    qstore(out + 4, dual);
}

/* :48-67 — the trig is evaluated in double (yaw * 0.5 promotes) and rounded to T */
void orc_dq_from_euler(float yaw, float pitch, float roll, float x, float y, float z, float out[8]) {
    float cy = (float)cos(yaw * 0.5);
    float sy = (float)sin(yaw * 0.5);
    float cr = (float)cos(roll * 0.5);
    float sr = (float)sin(roll * 0.5);
    float cp = (float)cos(pitch * 0.5);
    float sp = (float)sin(pitch * 0.5);
    float q[4];
    q[0] = cy * cr * cp + sy * sr * sp;
    q[1] = cy * sr * cp - sy * cr * sp;
    q[2] = cy * cr * sp + sy * sr * cp;
    q[3] = sy * cr * cp - cy * sr * sp;
    float t[3] = {x, y, z};
    orc_dq_from_quat_trans(q, t, out);
}

/* :70-86 */
void orc_dq_from_rodrigues(const float rod[3], const float t[3], float out[8]) {
    /* cv::norm(Vec3f) returns double */
    double nrm   = sqrt((double)rod[0] * rod[0] + (double)rod[1] * rod[1] + (double)rod[2] * rod[2]);
    double theta = 2 * atan(nrm);
    /* axis = rodrigues / theta (Vec3f / double -> Vec3f), then normalised */
    float ax[3] = {(float)(rod[0] / theta), (float)(rod[1] / theta), (float)(rod[2] / theta)};
    double an   = sqrt((double)ax[0] * ax[0] + (double)ax[1] * ax[1] + (double)ax[2] * ax[2]);
    float axn[3] = {(float)(ax[0] / an), (float)(ax[1] / an), (float)(ax[2] / an)};
    double s     = sin(0.5 * theta);
    float q[4]   = {(float)cos(0.5 * theta), (float)(s * axn[0]), (float)(s * axn[1]), (float)(s * axn[2])};
    /* DualQuaternion<T> dq(normalize(rotation), translation): normalised twice (:84 + :43) */
    quat qn = qnormalize_ref(qload(q));
    float qq[4];
    qstore(qq, qn);
    orc_dq_from_quat_trans(qq, t, out);
}

/* :99-117 */
void orc_dq_add(const float a[8], const float b[8], float out[8]) {
    qstore(out, qadd(qload(a), qload(b)));
    qstore(out + 4, qadd(qload(a + 4), qload(b + 4)));
}
void orc_dq_sub(const float a[8], const float b[8], float out[8]) {
    qstore(out, qsub(qload(a), qload(b)));
    qstore(out + 4, qsub(qload(a + 4), qload(b + 4)));
}

/* :120-125 — scalar scales the DUAL part only */
void orc_dq_scale(const float a[8], float s, float out[8]) {
    qstore(out, qload(a));
    qstore(out + 4, qscale(qload(a + 4), s));
}

/* :127-135 */
void orc_dq_mul(const float a[8], const float b[8], float out[8]) {
    quat ar = qload(a), ad = qload(a + 4), br = qload(b), bd = qload(b + 4);
    quat real = qmul(ar, br);
    quat dual = qadd(qmul(ar, bd), qmul(ad, br));
    qstore(out, real);
    qstore(out + 4, dual);
}

/* :139-144 — real part only; magnitude = sqrtf(dot(real,real)) */
void orc_dq_normalize(const float a[8], float out[8]) {
    quat r          = qload(a);
    float magnitude = sqrtf(r.w * r.w + r.x * r.x + r.y * r.y + r.z * r.z);
    qstore(out, qscale(r, 1.0f / magnitude));
    qstore(out + 4, qload(a + 4));
}

/* :94-97 */
void orc_dq_get_translation(const float a[8], float out[3]) {
    quat q = qmul(qscale(qload(a + 4), 2.0f), qconj(qload(a)));
    out[0] = q.x, out[1] = q.y, out[2] = q.z;
}

/* :204-215. cv::Vec3f arithmetic is float; evaluation order as written:
 * v + 2*(r x (r x v + w v)) + 2*(w d - d0 r + r x d) */
static inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
void orc_dq_transform_vertex(const float a[8], const float v[3], float out[3]) {
    const float w = a[0], d0 = a[4];
    const float r[3] = {a[1], a[2], a[3]}, d[3] = {a[5], a[6], a[7]};
    float rxv[3], tmp[3], t1[3], rxd[3];
    cross3(r, v, rxv);
    for (int i = 0; i < 3; ++i) tmp[i] = rxv[i] + w * v[i];
    cross3(r, tmp, t1);
    cross3(r, d, rxd);
    for (int i = 0; i < 3; ++i) {
        float t2 = (w * d[i] - d0 * r[i]) + rxd[i];
        out[i]   = (v[i] + 2.f * t1[i]) + 2.f * t2;
    }
}

/* :148-160 */
float orc_dq_roll(const float a[8]) {
    float sinr = (float)(+2.0 * (a[0] * a[1] + a[2] * a[3]));
    float cosr = (float)(+1.0 - 2.0 * (a[1] * a[1] + a[2] * a[2]));
    float roll = (float)atan2(sinr, cosr);
    if (roll > M_PI) roll -= (float)M_PI_2;
    return roll;
}
/* :162-176 */
float orc_dq_pitch(const float a[8]) {
    float sinp = (float)(+2.0 * (a[0] * a[2] - a[3] * a[1]));
    if (fabs(sinp) >= 1) return (float)copysign(M_PI / 2, sinp);
    return (float)asin(sinp);
}
/* :178-190 */
float orc_dq_yaw(const float a[8]) {
    float siny = (float)(+2.0 * (a[0] * a[3] + a[1] * a[2]));
    float cosy = (float)(+1.0 - 2.0 * (a[2] * a[2] + a[3] * a[3]));
    float yaw  = (float)atan2(siny, cosy);
    if (yaw > M_PI) yaw -= (float)M_PI_2;
    return yaw;
}
/* :194-200 */
void orc_dq_get_rodrigues(const float a[8], float out[3]) {
    double nrm   = sqrt((double)a[1] * a[1] + (double)a[2] * a[2] + (double)a[3] * a[3]);
    double theta = 2 * acos(a[0]);
    double tn    = tan(0.5 * theta);
    /* tan(..) * q / norm : (double * Vec3f) -> Vec3f, then / double */
    for (int i = 0; i < 3; ++i) out[i] = (float)((float)(tn * a[1 + i]) / nrm);
}
