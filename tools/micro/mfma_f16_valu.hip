// Do VALU instructions of ONE wave overlap with the fp16 MFMAs (v_mfma_f32_32x32x16_f16, 8 passes = 32 cycles) of ANOTHER wave on
// the same gfx950 SIMD?  The streaming residual-block kernel (conv_rbs.hip.h) puts one MFMA-heavy and one VALU-heavy wave on every
// SIMD; DESIGN.md 4.5 prices a step as MFMA cycles + VALU cycles.   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f16_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// mode bit0: waves 0-3 (one per SIMD) run an MFMA chain; bit1: waves 4-7 (the second wave of each SIMD) run VALU FMAs;
// bit2: ... run v_exp_f32 (transcendental, quarter rate) instead
__global__ void __launch_bounds__(512) split_kernel(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc;
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            f16x8 a, b;
            for (int j = 0; j < 8; j++) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(blockIdx.x * 0.002f + j); }
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);      // 16 x 32 = 512 cycles
            for (int r = 0; r < 16; r++) s += acc[r];
        }
    } else if (mode & 2) {
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
        const float m = 1.0001f, c = 0.5f;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int u = 0; u < 16; u++) {       // 128 VALU FMAs x 4 cycles = 512 cycles
                x0 = fmaf(x0, m, c); x1 = fmaf(x1, m, c); x2 = fmaf(x2, m, c); x3 = fmaf(x3, m, c);
                x4 = fmaf(x4, m, c); x5 = fmaf(x5, m, c); x6 = fmaf(x6, m, c); x7 = fmaf(x7, m, c);
            }
        s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if (mode & 4) {
        float x0 = threadIdx.x * 1e-3f, x1 = .1f, x2 = .2f, x3 = .3f;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int u = 0; u < 8; u++) {        // 32 v_exp_f32: 512 cycles if they are quarter rate
                x0 = __builtin_amdgcn_exp2f(x0) * 0.25f; x1 = __builtin_amdgcn_exp2f(x1) * 0.25f;
                x2 = __builtin_amdgcn_exp2f(x2) * 0.25f; x3 = __builtin_amdgcn_exp2f(x3) * 0.25f;
            }
        s = x0 + x1 + x2 + x3;
    }
    if (s == 12345.f) out[threadIdx.x] = s;
}

float* out;
template <typename F>
float time_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    (void)hipMalloc(&out, 4096);
    const int iters = 4000;
    const char* names[8] = {"", "MFMA waves only", "FMA waves only", "MFMA + FMA waves", "exp waves only", "MFMA + exp waves", "", ""};
    for (int mode : {1, 2, 3, 4, 5}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(split_kernel, dim3(256), dim3(512), 0, 0, out, iters, mode); });
        printf("mode %d (%-18s): %.3f ms   (512 cycles x %d iterations at 2.4 GHz = %.3f ms)\n", mode, names[mode], ms, iters, iters * 512.0 / 2.4e6);
    }
    return 0;
}
