// Shader clock actually delivered while a kernel runs: s_memtime (core clock) vs s_memrealtime (100 MHz).
// Workload mix selectable: MFMA only / MFMA + LDS traffic / MFMA + LDS + global loads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) mix_kernel(const f32x4* g, unsigned long long* stamps, float* out, int iters, int mode) {
    __shared__ f32x4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f32x4{1.f, 2.f, 3.f, (float)i};
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    f32x4 a = lds[threadIdx.x], b = lds[threadIdx.x + 256];
    f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
    const size_t gbase = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; it++) {
        if (mode >= 1) { a = lds[(threadIdx.x + it) & 1023]; b = lds[(threadIdx.x + 2 * it + 256) & 1023]; }
        if (mode >= 2) gsum += g[(gbase + (size_t)it * 65536) & ((1u << 22) - 1)];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[e], a[e], acc, 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = gsum[0] + gsum[1] + gsum[2] + gsum[3];
    for (int r = 0; r < 16; r++) s += acc[r];
    if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    f32x4* g; unsigned long long* st; float* out;
    const int grid = 256 * 8;
    hipMalloc(&g, (size_t)(1u << 22) * 16); hipMemset(g, 0, (size_t)(1u << 22) * 16);
    hipMalloc(&st, grid * 16); hipMalloc(&out, 4096);
    unsigned long long* h = new unsigned long long[grid * 2];
    for (int mode = 0; mode < 3; mode++)
        for (int iters : {2000, 20000}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(256), 0, 0, g, st, out, iters, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(256), 0, 0, g, st, out, iters, mode);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, st, grid * 16, hipMemcpyDeviceToHost);
            double c = 0, r = 0;
            for (int i = 0; i < grid; i++) { c += h[2 * i]; r += h[2 * i + 1]; }
            printf("mode %d (%s) iters %5d: %8.3f ms  %6.1f TFLOP/s  shader clock %.0f MHz\n", mode,
                   mode == 0 ? "MFMA" : mode == 1 ? "MFMA+LDS" : "MFMA+LDS+global", iters, ms,
                   (double)grid * 4 * iters * 8 * 4096.0 / ms / 1e9, c / r * 100.0);
        }
    return 0;
}
