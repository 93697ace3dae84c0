// A few lines of GoogleTest's surface (TEST, EXPECT_EQ, EXPECT_NEAR, RUN_ALL_TESTS) so that the
// shape tests can be written the way the reference writes them; GoogleTest itself is a URL fetch
// there (tests/CMakeLists.txt:9-12) and cannot be installed offline.
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace mini_gtest {
struct Case { std::string name; std::function<void()> fn; };
inline std::vector<Case>& cases() { static std::vector<Case> c; return c; }
inline int& failures() { static int f = 0; return f; }
struct Reg { Reg(const char* s, const char* n, std::function<void()> f) { cases().push_back({std::string(s) + "." + n, f}); } };
inline int run_all() {
    for (auto& c : cases()) {
        const int before = failures();
        c.fn();
        std::printf("[%s] %s\n", failures() == before ? "  OK  " : "FAILED", c.name.c_str());
    }
    std::printf("%d test(s), %d failure(s)\n", (int)cases().size(), failures());
    return failures() ? 1 : 0;
}
}  // namespace mini_gtest

#define TEST(suite, name)                                                          \
    static void suite##_##name##_body();                                           \
    static mini_gtest::Reg suite##_##name##_reg(#suite, #name, suite##_##name##_body); \
    static void suite##_##name##_body()
#define EXPECT_EQ(a, b)                                                                                      \
    do { auto _a = (a); auto _b = (b); if (!(_a == _b)) { mini_gtest::failures()++;                          \
        std::printf("%s:%d: EXPECT_EQ(%s, %s) failed: %g vs %g\n", __FILE__, __LINE__, #a, #b, (double)_a, (double)_b); } } while (0)
#define EXPECT_NEAR(a, b, tol)                                                                               \
    do { double _a = (a), _b = (b); if (!(std::fabs(_a - _b) <= (tol))) { mini_gtest::failures()++;          \
        std::printf("%s:%d: EXPECT_NEAR(%s, %s) failed: %.9g vs %.9g\n", __FILE__, __LINE__, #a, #b, _a, _b); } } while (0)
#define RUN_ALL_TESTS() mini_gtest::run_all()
