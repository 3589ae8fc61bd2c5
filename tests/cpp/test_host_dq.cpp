// test_host_dq.cpp — DualQuaternion<float> of the host adaptor against the reference's 21
// known answers (test/quaternion_test.cpp; tolerance 1e-4, :40).  CPU only.
#include <sstream>

#include <dynfu/utils/dual_quaternion.hpp>

#include "minitest.hpp"

namespace {
const float RAD90 = (float)(M_PI / 2), RAD60 = (float)(M_PI / 3), RAD45 = (float)(M_PI / 4), RAD30 = (float)(M_PI / 6);
const float TOL = 0.0001f;
typedef DualQuaternion<float> DQ;
DQ dq90() { return DQ(RAD90, RAD90, RAD90, 0.f, 0.f, 0.f); }
DQ dq45() { return DQ(RAD45, RAD45, RAD45, 0.f, 0.f, 0.f); }
DQ dq0() { return DQ(0.f, 0.f, 0.f, 0.f, 0.f, 0.f); }
DQ dq30() { return DQ(0.f, RAD30, 0.f, 0.f, 0.f, 100.f); }
DQ dq30rot() { return DQ(0.f, RAD30, 0.f, 0.f, 0.f, 0.f); }
DQ dqA() { return DQ(RAD30, RAD45, RAD30, 30.f, 20.f, 10.f); }

void expect_quat(const dfa::quaternion<float>& q, double a, double b, double c, double d) {
    ASSERT_NEAR(q.R_component_1(), a, TOL);
    ASSERT_NEAR(q.R_component_2(), b, TOL);
    ASSERT_NEAR(q.R_component_3(), c, TOL);
    ASSERT_NEAR(q.R_component_4(), d, TOL);
}
void expect_same_real(const DQ& x, const DQ& y) {
    expect_quat(x.getReal(), y.getReal().a, y.getReal().b, y.getReal().c, y.getReal().d);
}
}  // namespace

TEST(DualQuaternionTest, TestReal) {  // :57-65
    expect_quat(dq45().getReal(), 0.8446231020115715, 0.19134170284356308, 0.4619399539487806, 0.19134170284356303);
}
TEST(DualQuaternionTest, TestDual) {  // :70-91
    expect_quat(dq30().getReal(), 0.9659, 0.0, 0.2588, 0.0);
    expect_quat(dq30().getDual(), 0.0, -12.9409, 0.0, 48.2962);
}
TEST(DualQuaternionTest, TestFromRodrigues) {  // :93-120
    const dfa::Vec3f t(0.f, 0.f, 0.f);
    expect_same_real(DQ(dfa::Vec3f(0.f, 0.267949192431123f, 0.f), t), dq30rot());
    expect_same_real(DQ(dfa::Vec3f(0.226540919660986f, 0.546918160678027f, 0.226540919660986f), t), dq45());
    expect_same_real(DQ(dfa::Vec3f(0.f, 1.f, 0.f), t), dq90());
}
TEST(DualQuaternionTest, TestSum) {  // :123-140
    DQ s = dq45() + dq30();
    expect_quat(s.getReal(), 1.8105, 0.1913, 0.7208, 0.1913);
    expect_quat(s.getDual(), 0.0, -12.9410, 0.0, 48.2963);
}
TEST(DualQuaternionTest, TestComposeRotations) {  // :143-157
    dfa::PointXYZ v(0, 0, 1);
    dfa::PointXYZ twice = dq90().transformVertex(dq90().transformVertex(v));
    dfa::PointXYZ comp  = (dq90() * dq90()).transformVertex(v);
    ASSERT_NEAR(twice.x, comp.x, TOL);
    ASSERT_NEAR(twice.y, comp.y, TOL);
    ASSERT_NEAR(twice.z, comp.z, TOL);
}
TEST(DualQuaternionTest, TestSumAssign) {  // :160-180
    DQ s = dqA();
    s += dq30();
    expect_quat(s.getReal(), 1.8536, 0.1353, 0.6778, 0.1353);
    expect_quat(s.getDual(), -6.8953, -0.3683, 7.5233, 57.6655);
}
TEST(DualQuaternionTest, TestDiff) {  // :183-201
    DQ d = dq45() - dq30();
    expect_quat(d.getReal(), -0.1213, 0.1913, 0.2031, 0.1913);
    expect_quat(d.getDual(), 0.0, 12.9410, 0.0, -48.2963);
}
TEST(DualQuaternionTest, TestDiffAssign) {  // :203-223
    DQ d = dqA();
    d -= dq30();
    expect_quat(d.getReal(), -0.0783, 0.1353, 0.1601, 0.1353);
    expect_quat(d.getDual(), -6.8953, 25.5137, 7.5233, -38.9271);
}
TEST(DualQuaternionTest, TestScale) {  // :226-243
    DQ s = dq30() * 0.30f;
    expect_same_real(s, dq30());
    expect_quat(s.getDual(), 0.0, -3.8823, 0.0, 14.4889);
}
TEST(DualQuaternionTest, TestScaleAssign) {  // :245-265
    DQ s = dqA();
    s *= 0.30f;
    expect_same_real(s, dqA());
    expect_quat(s.getDual(), -2.0686, 3.7718, 2.2570, 2.8108);
}
TEST(DualQuaternionTest, TestMul) {  // :268-286
    DQ m = dq30() * dq45();
    expect_quat(m.getReal(), 0.6963, 0.2343, 0.6648, 0.1353);
    expect_quat(m.getDual(), -6.7650, -33.2402, 11.7172, 34.8142);
}
TEST(DualQuaternionTest, TestMulAssign) {  // :289-308
    DQ m = dqA();
    m *= dq30();
    expect_quat(m.getReal(), 0.7490, 0.0957, 0.6344, 0.1657);
    expect_quat(m.getDual(), -13.3911, 18.4657, -2.8031, 60.5945);
}
TEST(DualQuaternionTest, TestNormalize) {  // :311-330
    DQ s = dq45() + dq30();
    DQ n = s.normalize();
    expect_quat(n.getReal(), 0.9203, 0.0973, 0.3663, 0.0973);
    expect_quat(n.getDual(), 0.0, -12.9410, 0.0, 48.2963);
}
TEST(DualQuaternionTest, TestDoNotTransform) {  // :333-340
    dfa::PointXYZ r = dq0().transformVertex(dfa::PointXYZ(0, 0, 1));
    ASSERT_NEAR(r.x, 0, TOL), ASSERT_NEAR(r.y, 0, TOL), ASSERT_NEAR(r.z, 1, TOL);
}
TEST(DualQuaternionTest, TestRotate) {  // :343-350
    dfa::PointXYZ r = dq90().transformVertex(dfa::PointXYZ(0, 0, 1));
    ASSERT_NEAR(r.x, 1, TOL), ASSERT_NEAR(r.y, 0, TOL), ASSERT_NEAR(r.z, 0, TOL);
}
TEST(DualQuaternionTest, TestTranslate) {  // :353-362
    dfa::PointXYZ r = DQ(0.f, 0.f, 0.f, 1.f, 0.f, 0.f).transformVertex(dfa::PointXYZ(0, 0, 1));
    ASSERT_NEAR(r.x, 1, TOL), ASSERT_NEAR(r.y, 0, TOL), ASSERT_NEAR(r.z, 1, TOL);
}
TEST(DualQuaternionTest, TestTranslateAndRotate) {  // :365-374
    dfa::PointXYZ r = DQ(RAD90, RAD90, RAD90, 1.f, 0.f, 0.f).transformVertex(dfa::PointXYZ(0, 0, 1));
    ASSERT_NEAR(r.x, 2, TOL), ASSERT_NEAR(r.y, 0, TOL), ASSERT_NEAR(r.z, 0, TOL);
}
TEST(DualQuaternionTest, RollTest) {  // :377-387
    ASSERT_NEAR(dq30rot().getRoll(), 0, TOL);
    ASSERT_NEAR(dq45().getRoll(), RAD45, TOL);
    ASSERT_NEAR(dq90().getRoll(), RAD90, TOL);
}
TEST(DualQuaternionTest, PitchTest) {  // :390-398
    ASSERT_NEAR(dq30().getPitch(), RAD30, TOL);
    ASSERT_NEAR(dq45().getPitch(), RAD45, TOL);
    ASSERT_NEAR(dq90().getPitch(), RAD90, TOL);
}
TEST(DualQuaternionTest, YawTest) {  // :401-411
    ASSERT_NEAR(dq30rot().getYaw(), 0, TOL);
    ASSERT_NEAR(dq45().getYaw(), RAD45, TOL);
    ASSERT_NEAR(dq90().getYaw(), RAD90, TOL);
}
TEST(DualQuaternionTest, ConvertToEulerAnglesTest) {  // :414-435
    const dfa::Vec3f e30 = dq30rot().getEulerAngles(), e45 = dq45().getEulerAngles(), e90 = dq90().getEulerAngles();
    ASSERT_NEAR(e30[0], 0, TOL), ASSERT_NEAR(e45[0], RAD45, TOL), ASSERT_NEAR(e90[0], RAD90, TOL);
    ASSERT_NEAR(e30[1], RAD30, TOL), ASSERT_NEAR(e45[1], RAD45, TOL), ASSERT_NEAR(e90[1], RAD90, TOL);
    ASSERT_NEAR(e30[2], 0, TOL), ASSERT_NEAR(e45[2], RAD45, TOL), ASSERT_NEAR(e90[2], RAD90, TOL);
}
TEST(DualQuaternionTest, ConvertToRodriguesTest) {  // :438-456
    const dfa::Vec3f r30 = dq30rot().getRodrigues(), r45 = dq45().getRodrigues(), r90 = dq90().getRodrigues();
    ASSERT_NEAR(r30[0], 0, TOL), ASSERT_NEAR(r30[1], 0.267949192431123, TOL), ASSERT_NEAR(r30[2], 0, TOL);
    ASSERT_NEAR(r45[0], 0.226540919660986, TOL), ASSERT_NEAR(r45[1], 0.546918160678027, TOL);
    ASSERT_NEAR(r45[2], 0.226540919660986, TOL);
    ASSERT_NEAR(r90[0], 0, TOL), ASSERT_NEAR(r90[1], 1, TOL), ASSERT_NEAR(r90[2], 0, TOL);
}
TEST(DualQuaternionTest, TestToString) {  // :458-462
    std::ostringstream os;
    os << dq30();
    ASSERT_EQ(os.str(), std::string("real: (0.965926,0,0.258819,0)\ndual: (0,-12.941,0,48.2963)\n"));
}

int main(int argc, char** argv) { return mt::run_all(argc, argv); }
