// TEST INFRASTRUCTURE -- the slice of googletest that redtail's stereoDNN/tests/tests_main.cpp uses (TEST, EXPECT_* / ASSERT_*
// with streamed messages, ADD_FAILURE, InitGoogleTest incl. --gtest_filter, RUN_ALL_TESTS), so that the reference's own plugin
// test suite compiles UNTOUCHED in an image without googletest and runs against libnvstereo_inference.so
// (redtail_amd/build.py:build_reference_tests).  Semantics follow googletest's documentation (EXPECT_FLOAT_EQ = 4 ULPs).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace testing {

class Message {
public:
    template <typename T> Message& operator<<(const T& v) { ss_ << v; return *this; }
    Message& operator<<(std::ostream& (*f)(std::ostream&)) { ss_ << f; return *this; }
    std::string str() const { return ss_.str(); }
private:
    std::ostringstream ss_;
};

struct TestInfo {
    std::string suite, name;
    std::function<void()> body;
};
struct State {
    std::vector<TestInfo> tests;
    std::string filter = "*";
    int failures_in_current = 0;
    int reported_in_current = 0;
    static State& get() { static State s; return s; }
};

class AssertHelper {
public:
    AssertHelper(const char* file, int line, const std::string& what) : file_(file), line_(line), what_(what) {}
    void operator=(const Message& m) const {
        State& s = State::get();
        s.failures_in_current++;
        if (s.reported_in_current++ < 20) {      // a wrong tensor has thousands of wrong elements: print the first few
            std::cout << file_ << ":" << line_ << ": Failure\n" << what_ << "\n";
            const std::string extra = m.str();
            if (!extra.empty()) std::cout << extra << "\n";
        }
    }
private:
    const char* file_;
    int line_;
    std::string what_;
};

struct Registrar {
    Registrar(const char* suite, const char* name, std::function<void()> body) { State::get().tests.push_back({suite, name, std::move(body)}); }
};

namespace internal {
template <typename T> std::string show(const T& v) {
    std::ostringstream ss;
    if constexpr (std::is_pointer<T>::value || std::is_null_pointer<T>::value) ss << (const void*)v;
    else if constexpr (std::is_enum<T>::value) ss << (long long)v;
    else if constexpr (std::is_floating_point<T>::value) { ss.precision(9); ss << v; }
    else ss << v;
    return ss.str();
}
template <typename A, typename B> std::string cmp_msg(const char* op, const char* ea, const char* eb, const A& a, const B& b) {
    return std::string("Expected: (") + ea + ") " + op + " (" + eb + "), actual: " + show(a) + " vs " + show(b);
}
// googletest's AlmostEquals: within 4 units in the last place (sign-magnitude -> biased integers)
inline bool float_eq(float a, float b) {
    if (std::isnan(a) || std::isnan(b)) return false;
    auto biased = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u + 1 : u | 0x80000000u; };
    const uint32_t x = biased(a), y = biased(b);
    return (x > y ? x - y : y - x) <= 4;
}
inline bool glob(const char* p, const char* s) {          // '*' and '?' patterns, ':' separated alternatives handled by the caller
    if (!*p) return !*s;
    if (*p == '*') return glob(p + 1, s) || (*s && glob(p, s + 1));
    return *s && (*p == '?' || *p == *s) && glob(p + 1, s + 1);
}
inline bool selected(const std::string& filter, const std::string& full) {
    std::string pos = filter, neg;
    const size_t dash = filter.find('-');
    if (dash != std::string::npos) { pos = filter.substr(0, dash); neg = filter.substr(dash + 1); }
    if (pos.empty()) pos = "*";
    auto any = [&](const std::string& list) {
        std::stringstream ss(list);
        std::string pat;
        while (std::getline(ss, pat, ':'))
            if (glob(pat.c_str(), full.c_str())) return true;
        return false;
    };
    return any(pos) && !(!neg.empty() && any(neg));
}
}  // namespace internal

inline void InitGoogleTest(int* argc, char** argv) {
    int out = 1;
    for (int i = 1; i < *argc; i++) {
        if (std::strncmp(argv[i], "--gtest_filter=", 15) == 0) State::get().filter = argv[i] + 15;
        else if (std::strncmp(argv[i], "--gtest_", 8) == 0) continue;
        else argv[out++] = argv[i];
    }
    *argc = out;
}

inline int RunAllTests() {
    State& s = State::get();
    int ran = 0, failed = 0;
    std::vector<std::string> failed_names;
    for (auto& t : s.tests) {
        const std::string full = t.suite + "." + t.name;
        if (!internal::selected(s.filter, full)) continue;
        std::cout << "[ RUN      ] " << full << std::endl;
        s.failures_in_current = s.reported_in_current = 0;
        t.body();
        ran++;
        if (s.failures_in_current) {
            failed++;
            failed_names.push_back(full);
            std::cout << "[  FAILED  ] " << full << " (" << s.failures_in_current << " failed checks)" << std::endl;
        } else {
            std::cout << "[       OK ] " << full << std::endl;
        }
    }
    std::cout << "[==========] " << ran << " tests ran." << std::endl;
    std::cout << "[  PASSED  ] " << ran - failed << " tests." << std::endl;
    for (auto& n : failed_names) std::cout << "[  FAILED  ] " << n << std::endl;
    return failed ? 1 : 0;
}

}  // namespace testing

#define RUN_ALL_TESTS() ::testing::RunAllTests()

#define TEST(suite, name)                                                                             \
    static void suite##_##name##_body();                                                              \
    static ::testing::Registrar suite##_##name##_reg(#suite, #name, suite##_##name##_body);           \
    static void suite##_##name##_body()

// non-fatal / fatal forms; both accept `<< message`
#define RT_GTEST_CHECK_(ok, what, fatal)                                                               \
    switch (0) case 0: default:                                                                        \
        if (ok) ;                                                                                      \
        else fatal ::testing::AssertHelper(__FILE__, __LINE__, what) = ::testing::Message()
#define RT_GTEST_NONFATAL_
#define RT_GTEST_CMP_(a, b, op, fatal)                                                                 \
    RT_GTEST_CHECK_(((a) op (b)), ::testing::internal::cmp_msg(#op, #a, #b, (a), (b)), fatal)

#define EXPECT_EQ(a, b) RT_GTEST_CMP_(a, b, ==, RT_GTEST_NONFATAL_)
#define EXPECT_NE(a, b) RT_GTEST_CMP_(a, b, !=, RT_GTEST_NONFATAL_)
#define EXPECT_GT(a, b) RT_GTEST_CMP_(a, b, >, RT_GTEST_NONFATAL_)
#define EXPECT_GE(a, b) RT_GTEST_CMP_(a, b, >=, RT_GTEST_NONFATAL_)
#define EXPECT_LT(a, b) RT_GTEST_CMP_(a, b, <, RT_GTEST_NONFATAL_)
#define EXPECT_LE(a, b) RT_GTEST_CMP_(a, b, <=, RT_GTEST_NONFATAL_)
#define EXPECT_TRUE(c) RT_GTEST_CHECK_((c), std::string("Value of: " #c "\n  Actual: false\nExpected: true"), RT_GTEST_NONFATAL_)
#define EXPECT_FALSE(c) RT_GTEST_CHECK_(!(c), std::string("Value of: " #c "\n  Actual: true\nExpected: false"), RT_GTEST_NONFATAL_)
#define EXPECT_FLOAT_EQ(a, b)                                                                          \
    RT_GTEST_CHECK_(::testing::internal::float_eq((a), (b)), ::testing::internal::cmp_msg("~= (4 ULP)", #a, #b, (float)(a), (float)(b)), RT_GTEST_NONFATAL_)
#define EXPECT_NEAR(a, b, tol)                                                                         \
    RT_GTEST_CHECK_((std::fabs((double)(a) - (double)(b)) <= (double)(tol)),                           \
                    ::testing::internal::cmp_msg("within " #tol " of", #a, #b, (double)(a), (double)(b)), RT_GTEST_NONFATAL_)
#define ASSERT_EQ(a, b) RT_GTEST_CMP_(a, b, ==, return)
#define ASSERT_NE(a, b) RT_GTEST_CMP_(a, b, !=, return)
#define ASSERT_GT(a, b) RT_GTEST_CMP_(a, b, >, return)
#define ASSERT_GE(a, b) RT_GTEST_CMP_(a, b, >=, return)
#define ASSERT_LE(a, b) RT_GTEST_CMP_(a, b, <=, return)
#define ASSERT_TRUE(c) RT_GTEST_CHECK_((c), std::string("Value of: " #c "\n  Actual: false\nExpected: true"), return)
#define ADD_FAILURE() ::testing::AssertHelper(__FILE__, __LINE__, "Failed") = ::testing::Message()
