// Achievable v_mfma_f32_32x32x2_f32 rate on gfx950 as a function of independent accumulators per wave and
// waves per SIMD (calibration for the roofline of conv_mfma_f32_kernel).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_chain(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NACC>
void run(int wgs_per_cu, int iters) {
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(mfma_chain<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_chain<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("acc/wave %d  waves/SIMD %d  iters %d: %8.3f ms  %7.1f TFLOP/s\n", NACC, wgs_per_cu, iters, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<1>(w, 4000 / w);
        run<2>(w, 2000 / w);
        run<4>(w, 1000 / w);
    }
    // long run: does the rate hold for ~50 ms?
    run<4>(2, 20000);
    return 0;
}
