// Element-wise kernels: ELU, add(+activation), sigmoid.  HBM-bound: 16-byte accesses per lane,
// grid-stride, 2 loads + 1 store in flight per iteration.
#pragma once
#include "common.hip.h"

namespace rt {

// y = act(a [+ b]);  VEC floats per lane per iteration (4 => dwordx4).  Requires 16-B aligned
// pointers when VEC == 4; the launcher falls back to VEC == 1 otherwise.
template <int VEC, bool ADD>
__global__ void __launch_bounds__(256) ew_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ y, int64_t n, int act) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    const int64_t stride = (int64_t)gridDim.x * 256 * VEC;
    if (VEC == 4) {
        for (; i + 3 < n; i += stride) {
            f32x4 va = *reinterpret_cast<const f32x4*>(a + i);
            if (ADD) va += *reinterpret_cast<const f32x4*>(b + i);
            f32x4 r;
            for (int j = 0; j < 4; j++) r[j] = apply_act_rt(va[j], act);
            *reinterpret_cast<f32x4*>(y + i) = r;
        }
        if (i < n)   // the single lane that owns the ragged tail (n % 4 elements)
            for (int64_t j = i; j < n; j++) y[j] = apply_act_rt(ADD ? a[j] + b[j] : a[j], act);
    } else {
        for (; i < n; i += stride) y[i] = apply_act_rt(ADD ? a[i] + b[i] : a[i], act);
    }
}

// fp16 storage, fp32 math; 8 halfs (16 B) per lane.
template <int VEC, bool ADD>
__global__ void __launch_bounds__(256) ew_f16_kernel(const _Float16* __restrict__ a, const _Float16* __restrict__ b,
                                                     _Float16* __restrict__ y, int64_t n, int act) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    const int64_t stride = (int64_t)gridDim.x * 256 * VEC;
    if (VEC == 8) {
        for (; i + 7 < n; i += stride) {
            f16x8 va = *reinterpret_cast<const f16x8*>(a + i);
            f16x8 vb = va;
            if (ADD) vb = *reinterpret_cast<const f16x8*>(b + i);
            f16x8 r;
            for (int j = 0; j < 8; j++) {
                float v = (float)va[j];
                if (ADD) v += (float)vb[j];
                r[j] = (_Float16)apply_act_rt(v, act);
            }
            *reinterpret_cast<f16x8*>(y + i) = r;
        }
        if (i < n)
            for (int64_t j = i; j < n; j++)
                y[j] = (_Float16)apply_act_rt(ADD ? (float)a[j] + (float)b[j] : (float)a[j], act);
    } else {
        for (; i < n; i += stride)
            y[i] = (_Float16)apply_act_rt(ADD ? (float)a[i] + (float)b[i] : (float)a[i], act);
    }
}

// max |x| and the count of elements outside [-limit, limit) or non-finite, over `rows` rows of `valid` leading elements (row pitch
// `pitch`): the debug range check of the fp16-split domain (rt_check_range).  out[0] = bit pattern of max |x| (non-negative floats
// order like unsigned integers; NaN / inf count as violations and as 0x7f800000), out[1] = violations (low 32 bits), out[2] = high.
template <typename T>
__global__ void __launch_bounds__(256) range_check_kernel(const T* __restrict__ x, int64_t rows, int64_t valid, int64_t pitch, float limit, unsigned* out) {
    const int64_t n = rows * valid;
    unsigned mx = 0;
    unsigned long long bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / valid, c = i - r * valid;
        const float v = fabsf((float)x[r * pitch + c]);
        const bool finite = v < __builtin_inff();                  // false for inf and NaN
        if (!finite || v >= limit) bad++;
        const unsigned bits = finite ? __builtin_bit_cast(unsigned, v) : 0x7f800000u;
        mx = bits > mx ? bits : mx;
    }
    if (mx) atomicMax(out, mx);
    if (bad) atomicAdd(reinterpret_cast<unsigned long long*>(out + 2), bad);
}

// Order-independent 64-bit checksum of a device buffer (rt_hash_buffer): sum over the 4-byte words of mix(word, index) modulo 2^64.
// Bit-exact comparison of tensors without copying them to the host: the executor's launch trace (IExecutionContext::setLaunchTrace)
// hashes every launch's output on the launch's own stream, so two passes over the same input can be compared launch by launch.
__global__ void __launch_bounds__(256) hash_words_kernel(const unsigned* __restrict__ x, int64_t words, unsigned long long* out) {
    unsigned long long h = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long v = ((unsigned long long)x[i] + 0x9e3779b97f4a7c15ull) * (2ull * (unsigned long long)i + 1ull);
        v ^= v >> 29;
        h += v * 0xbf58476d1ce4e5b9ull;
    }
    if (h) atomicAdd(out, h);
}

}  // namespace rt
