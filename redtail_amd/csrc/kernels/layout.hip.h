// Layout glue kernels of the 3-D models (Transform / Padding / Slice plugins, channel concat).
// The fusing executor elides most of these; they exist so that the plugin API keeps its meaning.
#pragma once
#include "common.hip.h"

namespace rt {

// Copy `rows` segments of `len` elements: dst[r*dstride + i] = src[r*sstride + i]; optional zero tail
// of `ztail` elements after each segment (PaddingPlugin).  16-byte path when everything is aligned.
template <typename T>
__global__ void __launch_bounds__(256)
copy_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t len, int64_t sstride, int64_t dstride,
                 int64_t ztail) {
    const int64_t row = blockIdx.y;
    const T* s = src + row * sstride;
    T* d = dst + row * dstride;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len + ztail; i += stride)
        d[i] = i < len ? s[i] : (T)0;
}

// Generic 4-D permute (plus batch): y dims = x dims permuted by `order`; thread per output element,
// coalesced on the output side.
template <typename T>
__global__ void __launch_bounds__(256)
permute4d_kernel(const T* __restrict__ x, T* __restrict__ y, int o0, int o1, int o2, int o3, int64_t s0, int64_t s1,
                 int64_t s2, int64_t s3, int64_t total) {
    // o* = output dims, s* = input stride (elements) of the input dim that feeds output dim *
    const int64_t n = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        int64_t t = i;
        const int i3 = (int)(t % o3); t /= o3;
        const int i2 = (int)(t % o2); t /= o2;
        const int i1 = (int)(t % o1); t /= o1;
        const int i0 = (int)t;
        y[n * total + i] = x[n * total + i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
    }
}

// ... and the permutations that leave the innermost dimension(s) in place -- the Transform plugin of the 3-D models swaps the two OUTER
// dimensions, (K, D, H, W) <-> (D, K, H, W) (reference lib/transform_plugin.cpp:60-75, cudnnTransformTensor): contiguous runs of `inner`
// elements move as they are.  The element-per-thread kernel above spends three 64-bit divisions per element (NVSmall's volume: 2.8 TB/s of
// read + write); here a block decodes its run once and every thread moves kPermU elements with their loads in flight together.
// grid.x = chunks * o0 * o1 * o2, grid.y = samples; T = the moved unit (float, _Float16, or 16 bytes when everything is aligned).
constexpr int kPermU = 8;
template <typename T>
__global__ void __launch_bounds__(256)
permute_runs_kernel(const T* __restrict__ x, T* __restrict__ y, int o1, int o2, int64_t s0, int64_t s1, int64_t s2, int64_t inner, int chunks,
                    int64_t total) {
    const int chunk = blockIdx.x % chunks, run = blockIdx.x / chunks;       // run = (i0 * o1 + i1) * o2 + i2 in OUTPUT order
    const int i2 = run % o2, t = run / o2, i1 = t % o1, i0 = t / o1;
    const T* __restrict__ src = x + (int64_t)blockIdx.y * total + i0 * s0 + i1 * s1 + i2 * s2;
    T* __restrict__ dst = y + (int64_t)blockIdx.y * total + (int64_t)run * inner;
    const int64_t i = (int64_t)chunk * (256 * kPermU) + threadIdx.x;
    T v[kPermU];
#pragma unroll
    for (int u = 0; u < kPermU; u++)
        if (i + 256 * u < inner) v[u] = src[i + 256 * u];
#pragma unroll
    for (int u = 0; u < kPermU; u++)
        if (i + 256 * u < inner) dst[i + 256 * u] = v[u];
}

// Tensor format conversion at a plugin boundary -- what TensorRT inserts ("reformat" layers) between an fp32 tensor and
// an IPluginExt that asked for kHALF in kNCHW or kNC2HW2 (reference tests_main.cpp:301-321, 988-1026): kinds
// 0 = fp32 NCHW, 1 = fp16 NCHW, 2 = fp16 NC2HW2 (channel pairs (2i, 2i+1) of a pixel in one 4-byte slot, odd C zero padded).
// One thread per (sample, channel pair, pixel).
__device__ static __forceinline__ float cvt_load(const void* p, int kind, int64_t n, int C, int64_t inner, int c, int64_t i) {
    if (c >= C) return 0.f;
    if (kind == 0) return static_cast<const float*>(p)[(n * C + c) * inner + i];
    if (kind == 1) return (float)static_cast<const _Float16*>(p)[(n * C + c) * inner + i];
    return (float)static_cast<const _Float16*>(p)[((n * ((C + 1) / 2) + c / 2) * inner + i) * 2 + (c & 1)];
}
__device__ static __forceinline__ void cvt_store(void* p, int kind, int64_t n, int C, int64_t inner, int c, int64_t i, float v) {
    if (kind == 2) { static_cast<_Float16*>(p)[((n * ((C + 1) / 2) + c / 2) * inner + i) * 2 + (c & 1)] = (_Float16)(c < C ? v : 0.f); return; }
    if (c >= C) return;
    if (kind == 0) static_cast<float*>(p)[(n * C + c) * inner + i] = v;
    else static_cast<_Float16*>(p)[(n * C + c) * inner + i] = (_Float16)v;
}
__global__ void __launch_bounds__(256)
convert_format_kernel(const void* __restrict__ src, void* __restrict__ dst, int C, int64_t inner, int src_kind, int dst_kind) {
    const int64_t n = blockIdx.z;
    const int cp = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < inner; i += stride) {
#pragma unroll
        for (int h = 0; h < 2; h++) cvt_store(dst, dst_kind, n, C, inner, 2 * cp + h, i, cvt_load(src, src_kind, n, C, inner, 2 * cp + h, i));
    }
}

}  // namespace rt
