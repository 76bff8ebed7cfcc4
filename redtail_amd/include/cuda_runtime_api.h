// cuda_runtime_api.h -- the handful of CUDA runtime names redtail's Stereo DNN host code uses
// (sample_app/main.cpp:295-315, tests/tests_main.cpp:190-245), mapped onto the C ABI of
// include/rt_stereo.h so that such code runs on ROCm without a CUDA toolkit.  Streams and events are
// opaque HIP handles.
#ifndef REDTAIL_AMD_CUDA_RUNTIME_API_SHIM_H
#define REDTAIL_AMD_CUDA_RUNTIME_API_SHIM_H

#include <stddef.h>

#include "rt_stereo.h"

typedef struct ihipStream_t* cudaStream_t;
typedef struct ihipEvent_t* cudaEvent_t;
typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2,
                      cudaMemcpyDeviceToDevice = 3 };

static inline cudaError_t cudaMalloc(void** p, size_t n) { return rt_malloc(p, n); }
static inline cudaError_t cudaFree(void* p) { return rt_free(p); }
static inline cudaError_t cudaMemcpy(void* dst, const void* src, size_t n, enum cudaMemcpyKind kind) {
    int rc = kind == cudaMemcpyHostToDevice   ? rt_memcpy_h2d(dst, src, n, NULL)
             : kind == cudaMemcpyDeviceToHost ? rt_memcpy_d2h(dst, src, n, NULL)
                                              : rt_memcpy_d2d(dst, src, n, NULL);
    return rc ? rc : rt_stream_sync(NULL);
}
static inline cudaError_t cudaMemcpyAsync(void* dst, const void* src, size_t n, enum cudaMemcpyKind kind, cudaStream_t s) {
    return kind == cudaMemcpyHostToDevice   ? rt_memcpy_h2d(dst, src, n, s)
           : kind == cudaMemcpyDeviceToHost ? rt_memcpy_d2h(dst, src, n, s)
                                            : rt_memcpy_d2d(dst, src, n, s);
}
static inline cudaError_t cudaMemsetAsync(void* dst, int v, size_t n, cudaStream_t s) { return rt_memset(dst, v, n, s); }
static inline cudaError_t cudaStreamSynchronize(cudaStream_t s) { return rt_stream_sync(s); }
static inline cudaError_t cudaDeviceSynchronize(void) { return rt_stream_sync(NULL); }
static inline const char* cudaGetErrorString(cudaError_t) { return rt_last_error_string(); }
static inline const char* cudaGetErrorName(cudaError_t e) { return e == cudaSuccess ? "cudaSuccess" : "rtError"; }

#endif
