// minitest.hpp — a few macros so the host-adaptor tests read like the reference's gtest files
// (gtest itself is fetched from the network by the reference's build, test/CMakeLists.txt:12-18).
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace mt {
struct Case {
    std::string name;
    std::function<void()> fn;
};
inline std::vector<Case>& cases() {
    static std::vector<Case> c;
    return c;
}
struct Failure {
    std::string msg;
};
struct Reg {
    Reg(const char* n, std::function<void()> f) { cases().push_back({n, f}); }
};
inline int run_all(int argc, char** argv) {
    int failed = 0, ran = 0;
    for (auto& c : cases()) {
        if (argc > 1 && c.name.find(argv[1]) == std::string::npos) continue;
        ++ran;
        try {
            c.fn();
            std::printf("[       OK ] %s\n", c.name.c_str());
        } catch (const Failure& f) {
            ++failed;
            std::printf("[  FAILED  ] %s: %s\n", c.name.c_str(), f.msg.c_str());
        } catch (const std::exception& e) {
            ++failed;
            std::printf("[  FAILED  ] %s: exception %s\n", c.name.c_str(), e.what());
        }
    }
    std::printf("%d tests, %d failed\n", ran, failed);
    return failed ? 1 : 0;
}
}  // namespace mt

#define MT_CAT2(a, b) a##b
#define MT_CAT(a, b) MT_CAT2(a, b)
#define TEST(suite, name)                                                         \
    static void MT_CAT(suite##_##name, _body)();                                  \
    static mt::Reg MT_CAT(suite##_##name, _reg)(#suite "." #name, MT_CAT(suite##_##name, _body)); \
    static void MT_CAT(suite##_##name, _body)()
namespace mt {
inline void near(double a, double b, double tol, const char* ea, const char* eb, const char* file, int line) {
    if (!(std::fabs(a - b) <= tol)) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "%s:%d: |%s - %s| = |%g - %g| > %g", file, line, ea, eb, a, b, tol);
        throw Failure{buf};
    }
}
inline void truth(bool c, const char* e, const char* file, int line) {
    if (!c) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "%s:%d: %s is false", file, line, e);
        throw Failure{buf};
    }
}
}  // namespace mt
// expressions (usable in comma lists), throwing mt::Failure
#define ASSERT_NEAR(a, b, tol) mt::near((a), (b), (tol), #a, #b, __FILE__, __LINE__)
#define ASSERT_TRUE(c) mt::truth(static_cast<bool>(c), #c, __FILE__, __LINE__)
#define ASSERT_EQ(a, b) ASSERT_TRUE((a) == (b))
