// TEST INFRASTRUCTURE: tests_main.cpp includes the reference library's PRIVATE header "internal_utils.h" for two things --
// DimsUtils::getTensorSize and the CHECKL macro (stereoDNN/lib/internal_utils.h:41-55, internal_macros.h:14-19).  This header
// provides exactly those on top of our public headers, so that the reference's test source compiles untouched.
#ifndef REDTAIL_AMD_TEST_INTERNAL_UTILS_SHIM_H
#define REDTAIL_AMD_TEST_INTERNAL_UTILS_SHIM_H
#include <cassert>
#include <cuda_runtime_api.h>
#include <memory>
#include <string>

#include "NvInfer.h"
#include "redtail_tensorrt_plugins.h"

namespace redtail { namespace tensorrt {
using namespace nvinfer1;
class DimsUtils {
public:
    static size_t getTensorSize(Dims dims) {
        size_t n = 1;
        for (int i = 0; i < dims.nbDims; i++) n *= (size_t)dims.d[i];
        return n;
    }
};
inline void reportError(int status, const char* file, int line, const char* func, ILogger& log) {
    log.log(ILogger::Severity::kERROR, (std::string(file) + ":" + std::to_string(line) + ": " + func + ": error " + std::to_string(status) +
                                        " (" + rt_last_error_string() + ")").c_str());
}
} }
#undef CHECKL
#define CHECKL(status, log) do { auto res = (status); if ((int)res != 0) redtail::tensorrt::reportError((int)res, __FILE__, __LINE__, __FUNCTION__, log); } while (false)
#endif
