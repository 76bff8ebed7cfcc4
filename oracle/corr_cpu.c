/* Plain single-thread C restatement of the reference's correlation cost-volume kernel -- TEST / BENCH INFRASTRUCTURE ONLY
 * (oracle/: never linked into the product; tools/bench_ops.py times it as the op-level CPU baseline BASELINE.md section 2 promises,
 * tests/test_oracle_golden.py pins it to the reference's golden tensors).
 *
 * Follows lib/kernels.cu:168-200 (corrCostVolumeKernel<float>): one output per (disparity d = blockIdx.z, row, column),
 *     dst[d, y, x] = sum_c left[c, y, x] * right[c, y, x - d]     for x >= d,   0 otherwise,
 * channels accumulated in order c = 0 .. C-1 in fp32 (the kernel's `val += *pl * (*pr)`), output planes ordered from d = 0 up
 * ("Disparity feature maps are arranged from min to max").  Batch: the plugin launches the kernel once per sample
 * (lib/cost_volume_plugin.cpp:99,124), hence the outer loop.
 *
 *   gcc -O2 -shared -fPIC oracle/corr_cpu.c -o oracle/_ref/libcorr_cpu.so */
#include <stddef.h>
#include <stdint.h>

void corr_cost_volume_cpu(const float* left, const float* right, int32_t n, int32_t c, int32_t h, int32_t w, int32_t disp, float* dst) {
    const size_t stride = (size_t)h * w;
    for (int32_t b = 0; b < n; b++) {
        const float* l0 = left + (size_t)b * c * stride;
        const float* r0 = right + (size_t)b * c * stride;
        float* d0 = dst + (size_t)b * disp * stride;
        for (int32_t pad = 0; pad < disp; pad++)
            for (int32_t iy = 0; iy < h; iy++)
                for (int32_t ix = 0; ix < w; ix++) {
                    float val = 0.f;
                    if (ix >= pad) {
                        const float* pl = l0 + (size_t)iy * w + ix;
                        const float* pr = r0 + (size_t)iy * w + ix - pad;
                        for (int32_t i = 0; i < c; i++) {
                            val += *pl * (*pr);
                            pl += stride;
                            pr += stride;
                        }
                    }
                    d0[(size_t)pad * stride + (size_t)iy * w + ix] = val;
                }
    }
}
