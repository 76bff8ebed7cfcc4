// Development harness of corr_mfma_planar_kernel (corr_mfma.hip.h): the stand-alone planar correlation at ResNet-18 2D's shape
// (C = 32, D = 48, 185 x 629), timed back to back at batch 1 and 8, checked against an fp64 host evaluation (the ablations and variants
// it was used for are in profiles/r06_corr_dev.txt).
// Never part of the product; built and run by tools/r06/corr_harness.sh.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../redtail_amd/csrc/kernels/corr_mfma.hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int C = 32, D = 48, H = 185, W = 629, BMAX = 8;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> l((size_t)BMAX * C * H * W), r(l.size());
    for (auto& v : l) v = nd(rng);
    for (auto& v : r) v = nd(rng);
    float *dl, *dr, *dcv;
    CK(hipMalloc(&dl, l.size() * 4)); CK(hipMalloc(&dr, r.size() * 4)); CK(hipMalloc(&dcv, (size_t)BMAX * D * H * W * 4));
    CK(hipMemcpy(dl, l.data(), l.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr, r.data(), r.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dcv, 0xff, (size_t)BMAX * D * H * W * 4));
    int tpw = 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int b) {
        rt::CorrPlanarArgs a;
        a.left = dl; a.right = dr; a.out = dcv; a.C = C; a.H = H; a.W = W; a.D = D; a.blocks_x = (W + 31) / 32; a.batch = b;
        const unsigned nwg = (unsigned)(((long)a.blocks_x * H * b + 3) / 4);
        hipLaunchKernelGGL(rt::corr_mfma_planar_kernel, dim3((nwg + 7u) / 8u * 8u), dim3(256), 0, 0, a);
    };
    for (int tp : {1, 1, 1}) {
    tpw = tp;
    CK(hipMemset(dcv, 0xff, (size_t)BMAX * D * H * W * 4));
    launch(1);
    CK(hipDeviceSynchronize());
    std::vector<float> cv((size_t)D * H * W);
    CK(hipMemcpy(cv.data(), dcv, cv.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0; long bad = 0;
    for (int d = 0; d < D; d++)
        for (int y = 0; y < H; y += 7)
            for (int x = 0; x < W; x++) {
                double ref = 0, mag = 0;
                if (x >= d)
                    for (int c = 0; c < C; c++) {
                        const double a = l[((size_t)c * H + y) * W + x], b = r[((size_t)c * H + y) * W + x - d];
                        ref += a * b; mag += std::fabs(a * b);
                    }
                const double e = std::fabs(cv[((size_t)d * H + y) * W + x] - ref);
                if (!(e <= mag * 4.8e-7 + 1e-30)) bad++;
                if (mag > 0 && e / mag > worst) worst = e / mag;
            }
    printf("tpw %d check: %ld bad, worst |err| / sum|ab| = %.3g\n", tpw, bad, worst);
    for (int b : {1, 8}) {
        for (int i = 0; i < 20; i++) launch(b);
        const int N = 300;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < N; i++) launch(b);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / N, bytes = 4.0 * b * (2 * C + D) * H * W;
        printf("batch %d: %.2f us  %.2f TB/s  %.3f of 8 TB/s\n", b, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
    }
    }
    return 0;
}
