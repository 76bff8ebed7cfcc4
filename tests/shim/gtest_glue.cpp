// TEST INFRASTRUCTURE: the one googletest-internal symbol tests_main.cpp declares itself and calls (tests_main.cpp:22-35, 997)
#include <cstdarg>
#include <cstdio>
namespace testing { namespace internal {
enum GTestColor { COLOR_DEFAULT, COLOR_RED, COLOR_GREEN, COLOR_YELLOW };
void ColoredPrintf(GTestColor, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vprintf(fmt, ap);
    va_end(ap);
}
} }
