// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs?  (round 6: an unscaled low part of the fp16 split would be subnormal for |w| < 0.25)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float av, float bv) {
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float cases[][2] = {{1e-6f, 1.0f}, {3e-5f, 1.0f}, {6.2e-5f, 1.0f}, {1e-6f, 1e-6f}, {5.96e-8f, 1.0f}, {1e-6f, 1000.0f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        const float a16 = (float)(_Float16)c[0], b16 = (float)(_Float16)c[1];
        printf("a = %g (fp16 %g, %s)  b = %g: mfma sum of 16 products = %g, exact %g\n", c[0], a16, a16 < 6.1035e-5f ? "subnormal" : "normal", c[1], h, 16.0 * a16 * b16);
    }
    return 0;
}
