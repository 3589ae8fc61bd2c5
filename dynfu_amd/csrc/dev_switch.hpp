// dev_switch.hpp — A/B switches exist in DEVELOPMENT builds only.
//
// The product library (dynfu_amd/libdynfu_amd.so) has no environment look-ups on its call paths and none of the
// non-default kernel variants compiled in.  -DDFA_DEV_AB (dynfu_amd/libdynfu_amd_dev.so, built by dynfu_amd/build.py and
// loaded by the tests that compare variants and by the A/B scripts under tools/) turns the DFA_* environment switches
// back on: dev_env("NAME") is getenv there and a constant nullptr here, so every `if (dev_env(...))` branch — and, behind
// `#ifdef DFA_DEV_AB`, every kernel instantiation only such a branch launches — disappears from the product.
#pragma once
#include <cstdlib>

namespace dfa {
#ifdef DFA_DEV_AB
inline const char* dev_env(const char* name) { return std::getenv(name); }
constexpr bool kDevAB = true;
#else
constexpr const char* dev_env(const char*) { return nullptr; }
constexpr bool kDevAB = false;
#endif
inline int dev_env_int(const char* name, int fallback) {
    const char* e = dev_env(name);
    return e ? std::atoi(e) : fallback;
}
}  // namespace dfa
