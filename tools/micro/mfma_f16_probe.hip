// What v_mfma_f32_32x32x16_f16 does with fp16 subnormals and how it adds its 16 products to the accumulator --
// the two hardware facts the 3-term fp16 split of conv_split.hip.h (fp32 convolutions on the fp16 matrix pipe)
// depends on.   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f16_probe.hip -o tools/build/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A[32][16], B[16][32] row-major halves; C/D[32][32] floats
__global__ void __launch_bounds__(64) mfma_once(const _Float16* A, const _Float16* B, const float* C, float* D) {
    const int l = threadIdx.x, m = l & 31, kg = l >> 5;
    f16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = A[m * 16 + 8 * kg + e]; b[e] = B[(8 * kg + e) * 32 + m]; }
    f32x16 c;
    for (int r = 0; r < 16; r++) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + m];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + m] = c[r];
}

static std::vector<float> run(const std::vector<_Float16>& A, const std::vector<_Float16>& B, const std::vector<float>& C) {
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    return D;
}

int main() {
    std::vector<_Float16> A(512), B(512);
    std::vector<float> C(1024, 0.f);
    // 1. subnormal fp16 inputs: a = 2^-20 (subnormal, min normal is 2^-14), b = 1024 -> 16 products of 2^-10
    for (auto& v : A) v = (_Float16)ldexpf(1.f, -20);
    for (auto& v : B) v = (_Float16)1024.f;
    auto D = run(A, B, C);
    printf("subnormal A input : D[0][0] = %.9g (kept: %.9g, flushed: 0)\n", D[0], 16 * ldexp(1.0, -10));
    for (auto& v : A) v = (_Float16)1024.f;
    for (auto& v : B) v = (_Float16)ldexpf(3.f, -24);
    D = run(A, B, C);
    printf("subnormal B input : D[0][0] = %.9g (kept: %.9g)\n", D[0], 16 * 1024 * ldexp(3.0, -24));
    // 2. one product of 2^10 and fifteen of 1.5 ulp(2^10) = 3 * 2^-14: exact sum = 1024 + 22.5 ulp
    for (int m = 0; m < 32; m++)
        for (int k = 0; k < 16; k++) { A[m * 16 + k] = (_Float16)(k == 0 ? 32.f : ldexpf(3.f, -7)); }
    for (int k = 0; k < 16; k++)
        for (int n = 0; n < 32; n++) B[k * 32 + n] = (_Float16)(k == 0 ? 32.f : ldexpf(1.f, -7));
    D = run(A, B, C);
    printf("1 big + 15 small  : (D - 1024) / ulp = %.3f   (exact 22.5; small terms truncated to the big one's ulp: 15)\n", (D[0] - 1024.0) / ldexp(1.0, -13));
    // ... the same with the big value in the accumulator instead
    for (int m = 0; m < 32; m++) A[m * 16] = (_Float16)ldexpf(3.f, -7);
    for (int n = 0; n < 32; n++) B[n] = (_Float16)ldexpf(1.f, -7);
    for (auto& v : C) v = 1024.f;
    D = run(A, B, C);
    printf("big C + 16 small  : (D - 1024) / ulp = %.3f   (exact 24.0)\n", (D[0] - 1024.0) / ldexp(1.0, -13));
    // 3. random operands: error of one MFMA against the exact sum, in units of eps * sum|a*b|
    srand(1);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    double worst = 0, rms = 0;
    for (int rep = 0; rep < 20; rep++) {
        for (auto& v : A) v = (_Float16)(rnd() * (rep % 2 ? 8.f : 1.f));
        for (auto& v : B) v = (_Float16)rnd();
        for (auto& v : C) v = rnd() * 4.f;
        D = run(A, B, C);
        for (int m = 0; m < 32; m++)
            for (int n = 0; n < 32; n++) {
                double ex = C[m * 32 + n], mag = fabs(C[m * 32 + n]);
                for (int k = 0; k < 16; k++) { const double p = (double)(float)A[m * 16 + k] * (double)(float)B[k * 32 + n]; ex += p; mag += fabs(p); }
                const double e = fabs(D[m * 32 + n] - ex) / (mag * ldexp(1.0, -24));
                worst = e > worst ? e : worst; rms += e * e;
            }
    }
    printf("random operands   : |D - exact| / (2^-24 * sum|a*b|): worst %.3f, rms %.3f  (one correctly rounded fp32 result: <= ~1)\n", worst, sqrt(rms / (20 * 1024)));
    return 0;
}
