// Host-side SIMT emulator for the HIP kernels in redtail_amd/csrc  --  TEST INFRASTRUCTURE ONLY.
//
// There is no GPU in the development container, so the *unmodified* kernel sources
// (pure HIP, `#include <hip/hip_runtime.h>`) are additionally compiled for x86 with
//     amdclang++ -x c++ -I tests/emu ...
// which makes this file shadow the real <hip/hip_runtime.h>.  Every GPU thread becomes a
// fiber; __syncthreads(), wave shuffles and MFMA builtins are rendezvous points whose
// semantics (64-lane wavefront, gfx950 MFMA fragment maps) are modelled in hip_emu.cpp.
// This lets `pytest -m "not gpu"` execute the real tiling/indexing code of every kernel
// against the oracle.  It is never linked into the product libraries: those are built by
// hipcc for gfx950 only and fail loudly without a device.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __constant__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_smem());

struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));
typedef int int4 __attribute__((ext_vector_type(4)));
typedef unsigned uint2 __attribute__((ext_vector_type(2)));
typedef unsigned uint4 __attribute__((ext_vector_type(4)));
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100,
       hipErrorNotSupported = 801 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                     hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };

namespace hipemu {
struct Thread;
extern Thread* cur;                 // fiber currently running
extern uint3 g_tid, g_bid;
extern dim3 g_bdim, g_gdim;
void* dyn_smem();
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem);
void barrier();
float shfl(float v, int src_lane, int width);
unsigned long long ballot(int pred);
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
f32x16 mfma_32x32x2f32(float a, float b, f32x16 c);
f32x4 mfma_16x16x4f32(float a, float b, f32x4 c);
f32x16 mfma_32x32x16f16(f16x8 a, f16x8 b, f32x16 c);
f32x4 mfma_16x16x32f16(f16x8 a, f16x8 b, f32x4 c);
int lane_id();
}  // namespace hipemu

#define threadIdx (hipemu::g_tid)
#define blockIdx (hipemu::g_bid)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::barrier(); }
static inline float __shfl(float v, int src, int width = 64) { return hipemu::shfl(v, src, width); }
static inline float __shfl_xor(float v, int mask, int width = 64) {
    return hipemu::shfl(v, (hipemu::lane_id() % width) ^ mask, width);
}
static inline float __shfl_down(float v, unsigned d, int width = 64) {
    int l = hipemu::lane_id() % width;
    return hipemu::shfl(v, (l + (int)d < width) ? l + (int)d : l, width);
}
static inline float __shfl_up(float v, unsigned d, int width = 64) {
    int l = hipemu::lane_id() % width;
    return hipemu::shfl(v, (l - (int)d >= 0) ? l - (int)d : l, width);
}
static inline int __shfl(int v, int src, int width = 64) {
    float f; std::memcpy(&f, &v, 4); f = hipemu::shfl(f, src, width); std::memcpy(&v, &f, 4); return v;
}
// v_permlane32_swap (gfx950): lanes 32..63 of `a` are exchanged with lanes 0..31 of `b`; returns {a', b'}
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
static inline hipemu_u32x2 hipemu_permlane32_swap(unsigned a, unsigned b) {
    const int l = hipemu::lane_id();
    float fa, fb;
    std::memcpy(&fa, &a, 4); std::memcpy(&fb, &b, 4);
    const float b_from_low = hipemu::shfl(fb, l & 31, 64);       // upper lanes of a' take b's lower half
    const float a_from_high = hipemu::shfl(fa, 32 + (l & 31), 64);   // lower lanes of b' take a's upper half
    unsigned ub, ua;
    std::memcpy(&ub, &b_from_low, 4); std::memcpy(&ua, &a_from_high, 4);
    return hipemu_u32x2{l < 32 ? a : ub, l < 32 ? ua : b};
}
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipemu_permlane32_swap(a, b)
static inline unsigned long long __ballot(int p) { return hipemu::ballot(p); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }

#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu::mfma_32x32x2f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu::mfma_16x16x4f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu::mfma_32x32x16f16(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) hipemu::mfma_16x16x32f16(a, b, c)
// raw buffer resources: base + num_records, range check on voffset + soffset (out of range reads 0, stores vanish)
namespace hipemu {
struct buffer_rsrc { char* base; unsigned num_records; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
static inline buffer_rsrc make_buffer_rsrc(void* p, int, unsigned num, int) { return buffer_rsrc{static_cast<char*>(p), num}; }
static inline bool buffer_ok(const buffer_rsrc& r, unsigned voff, unsigned soff, unsigned bytes) {
    return (unsigned long long)voff + soff + bytes <= r.num_records;
}
static inline unsigned buffer_load_b32(const buffer_rsrc& r, unsigned voff, unsigned soff) {
    unsigned v = 0;
    if (buffer_ok(r, voff, soff, 4)) std::memcpy(&v, r.base + (size_t)voff + soff, 4);
    return v;
}
static inline u32x4 buffer_load_b128(const buffer_rsrc& r, unsigned voff, unsigned soff) {
    u32x4 v = {0, 0, 0, 0};
    if (buffer_ok(r, voff, soff, 16)) std::memcpy(&v, r.base + (size_t)voff + soff, 16);
    return v;
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// multi-dword accesses are range-checked per dword, like the hardware does for raw buffers
static inline u32x2 buffer_load_b64(const buffer_rsrc& r, unsigned voff, unsigned soff) {
    u32x2 v = {0, 0};
    for (int i = 0; i < 2; i++) {
        unsigned w = 0;
        if (buffer_ok(r, voff + 4 * i, soff, 4)) std::memcpy(&w, r.base + (size_t)voff + 4 * i + soff, 4);
        v[i] = w;
    }
    return v;
}
static inline void buffer_store_b64(u32x2 v, const buffer_rsrc& r, unsigned voff, unsigned soff) {
    for (int i = 0; i < 2; i++) {
        unsigned w = v[i];
        if (buffer_ok(r, voff + 4 * i, soff, 4)) std::memcpy(r.base + (size_t)voff + 4 * i + soff, &w, 4);
    }
}
static inline void buffer_store_b128(u32x4 v, const buffer_rsrc& r, unsigned voff, unsigned soff) {
    for (int i = 0; i < 4; i++) {
        unsigned w = v[i];
        if (buffer_ok(r, voff + 4 * i, soff, 4)) std::memcpy(r.base + (size_t)voff + 4 * i + soff, &w, 4);
    }
}
static inline unsigned short buffer_load_b16(const buffer_rsrc& r, unsigned voff, unsigned soff) {
    unsigned short v = 0;
    if (buffer_ok(r, voff, soff, 2)) std::memcpy(&v, r.base + (size_t)voff + soff, 2);
    return v;
}
static inline void buffer_store_b16(unsigned short v, const buffer_rsrc& r, unsigned voff, unsigned soff) {
    if (buffer_ok(r, voff, soff, 2)) std::memcpy(r.base + (size_t)voff + soff, &v, 2);
}
static inline void buffer_store_b32(unsigned v, const buffer_rsrc& r, unsigned voff, unsigned soff) {
    if (buffer_ok(r, voff, soff, 4)) std::memcpy(r.base + (size_t)voff + soff, &v, 4);
}
}  // namespace hipemu
#define __amdgpu_buffer_rsrc_t hipemu::buffer_rsrc
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) hipemu::make_buffer_rsrc(p, stride, num, flags)
#define __builtin_amdgcn_raw_buffer_load_b32(r, v, s, aux) hipemu::buffer_load_b32(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b128(r, v, s, aux) hipemu::buffer_load_b128(r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b32(d, r, v, s, aux) hipemu::buffer_store_b32(d, r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b64(r, v, s, aux) hipemu::buffer_load_b64(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b16(r, v, s, aux) hipemu::buffer_load_b16(r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b16(d, r, v, s, aux) hipemu::buffer_store_b16(d, r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b64(d, r, v, s, aux) hipemu::buffer_store_b64(d, r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b128(d, r, v, s, aux) hipemu::buffer_store_b128(d, r, v, s)
#define RT_WAVES_PER_EU(n)
#define RT_LDS_PTR(p) ((void*)(p))
// buffer_load ... lds: lane l of the wave writes `size` bytes at lds + size*l (range check per dword, like the loads)
static inline void hipemu_buffer_load_lds(const hipemu::buffer_rsrc& r, void* lds, unsigned size, unsigned voff, unsigned soff) {
    char* dst = static_cast<char*>(lds) + (size_t)hipemu::lane_id() * size;
    for (unsigned i = 0; i < size; i += 4) {
        unsigned w = 0;
        if (hipemu::buffer_ok(r, voff + i, soff, 4)) std::memcpy(&w, r.base + (size_t)voff + i + soff, 4);
        std::memcpy(dst + i, &w, 4);
    }
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, size, voff, soff, off, aux) hipemu_buffer_load_lds(r, lds, size, (voff) + (off), soff)
// atomics: fibers run one at a time, plain read-modify-write is atomic
static inline unsigned atomicMax(unsigned* p, unsigned v) { const unsigned o = *p; if (v > o) *p = v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)

// ---- runtime API subset used by rt_capi.hip ------------------------------------------------
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipStreamCreate(hipStream_t* s);
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipEventCreate(hipEvent_t* e);
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
const char* hipGetErrorString(hipError_t e);
const char* hipGetErrorName(hipError_t e);

template <typename K, typename... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    hipemu::launch([=]() { kernel(args...); }, grid, block, shmem);
}
