// Do VALU instructions overlap with MFMA on a gfx950 SIMD?  (a) waves that only run an MFMA chain share the SIMD
// with waves that only run VALU FMAs; (b) one wave interleaves K VALU ops after every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bit0: even waves run MFMA; bit1: odd waves run VALU.  blockDim = 512 -> 2 waves per SIMD
__global__ void __launch_bounds__(512) split_kernel(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = ((wave >> 2) & 1) == 0;      // waves 0-3 -> SIMD 0-3 (MFMA), waves 4-7 -> SIMD 0-3 (VALU)
    float s = 0.f;
    if (mfma_wave) {
        if (mode & 1) {
            f32x16 acc;
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f;
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            for (int r = 0; r < 16; r++) s += acc[r];
        }
    } else if (mode & 2) {
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
        const float m = 1.0001f, c = 0.5f;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int u = 0; u < 32; u++) {       // 256 VALU FMAs = 1024 cycles  vs 16 MFMA = 1024 cycles
                x0 = fmaf(x0, m, c); x1 = fmaf(x1, m, c); x2 = fmaf(x2, m, c); x3 = fmaf(x3, m, c);
                x4 = fmaf(x4, m, c); x5 = fmaf(x5, m, c); x6 = fmaf(x6, m, c); x7 = fmaf(x7, m, c);
            }
        s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int K>
__global__ void __launch_bounds__(256) interleave_kernel(float* out, int iters) {
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f;
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = j + threadIdx.x;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int u = 0; u < 16; u++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; j++) x[j & 7] = fmaf(x[j & 7], m, c);
        }
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += acc[r];
    for (int j = 0; j < 8; j++) s += x[j];
    if (s == 12345.f) out[threadIdx.x] = s;
}

float* out;
template <typename F>
float time_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    (void)hipMalloc(&out, 4096);
    const int iters = 2000;
    for (int mode = 1; mode <= 3; mode++) {
        float ms = time_ms([&] { hipLaunchKernelGGL(split_kernel, dim3(256), dim3(512), 0, 0, out, iters, mode); });
        printf("split mode %d (%s): %.3f ms  (ideal MFMA-only or VALU-only: %.3f ms)\n", mode,
               mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both", ms, iters * 1024.0 / 2.4e6);
    }
    float t0 = time_ms([&] { hipLaunchKernelGGL(interleave_kernel<0>, dim3(256), dim3(256), 0, 0, out, iters); });
    float t4 = time_ms([&] { hipLaunchKernelGGL(interleave_kernel<4>, dim3(256), dim3(256), 0, 0, out, iters); });
    float t8 = time_ms([&] { hipLaunchKernelGGL(interleave_kernel<8>, dim3(256), dim3(256), 0, 0, out, iters); });
    float t12 = time_ms([&] { hipLaunchKernelGGL(interleave_kernel<12>, dim3(256), dim3(256), 0, 0, out, iters); });
    float t16 = time_ms([&] { hipLaunchKernelGGL(interleave_kernel<16>, dim3(256), dim3(256), 0, 0, out, iters); });
    printf("interleave, 1 wave/SIMD, K VALU per MFMA (64 cycles): K=0 %.3f  K=4 %.3f  K=8 %.3f  K=12 %.3f  K=16 %.3f ms\n", t0, t4, t8, t12, t16);
    float u0 = time_ms([&] { hipLaunchKernelGGL(interleave_kernel<0>, dim3(1024), dim3(256), 0, 0, out, iters); });
    float u8 = time_ms([&] { hipLaunchKernelGGL(interleave_kernel<8>, dim3(1024), dim3(256), 0, 0, out, iters); });
    float u16 = time_ms([&] { hipLaunchKernelGGL(interleave_kernel<16>, dim3(1024), dim3(256), 0, 0, out, iters); });
    printf("interleave, 4 waves/SIMD: K=0 %.3f  K=8 %.3f  K=16 %.3f ms\n", u0, u8, u16);
    return 0;
}
