// Image front-end / back-end of the sample application on the GPU:
//   preprocess_bgr8_kernel   replaces readImgFile()'s cv::Mat pipeline (reference sample_app/main.cpp:83-98):
//                            u8 BGR HWC -> float, cv::resize(INTER_AREA), BGR -> RGB, HWC -> CHW, / 255
//   disparity_u16_kernel     replaces the result path (main.cpp:324-330): disparity * 256 (* width for the sigmoid
//                            output of ResNet-18 2D) -> saturating round-to-nearest-even conversion to 16 bit (KITTI PNG)
// so a camera frame crosses PCIe as 3 bytes per pixel instead of 12 and the host does no per-pixel work.
// INTER_AREA for down-scaling is the area-weighted box average of OpenCV's computeResizeAreaTab: destination pixel dx
// covers source interval [dx*s, (dx+1)*s), every source pixel contributes its overlap / cell width.
#pragma once
#include "common.hip.h"

namespace rt {

// overlap weights of destination index `d` along one axis (scale s >= 1, source size n): first source index and up
// to kMaxTaps weights, exactly as OpenCV builds its table (fractional head, whole pixels, fractional tail).
constexpr int kAreaMaxTaps = 8;       // supports scale factors up to 6
// (positions in double like OpenCV's table builder -- d * s in fp32 is off by 1e-4 pixels at x = 1000 -- weights float)
__device__ static __forceinline__ int area_taps(int d, double s, int n, float* wgt) {
    const double f1 = d * s, f2 = f1 + s;
    const double cell = fmin(s, (double)n - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = s2 < n ? s2 : n;
    s1 = s1 < s2 ? s1 : s2;
    int first = s1, k = 0;
    if (s1 - f1 > 1e-3) { first = s1 - 1; wgt[k++] = (float)((s1 - f1) / cell); }
    for (int sx = s1; sx < s2 && k < kAreaMaxTaps; sx++) wgt[k++] = (float)(1.0 / cell);
    if (f2 - s2 > 1e-3 && k < kAreaMaxTaps && s2 < n) wgt[k++] = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
    for (int i = k; i < kAreaMaxTaps; i++) wgt[i] = 0.f;
    return first;
}

// grid = (ceil(dw/256), dh, batch)
__global__ void __launch_bounds__(256)
preprocess_bgr8_kernel(const unsigned char* __restrict__ src, int sh, int sw, float* __restrict__ dst, int dh, int dw) {
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y, n = blockIdx.z;
    if (dx >= dw) return;
    const unsigned char* s = src + (int64_t)n * sh * sw * 3;
    float* d = dst + (int64_t)n * 3 * dh * dw;
    float b = 0.f, g = 0.f, r = 0.f;
    if (sh == dh && sw == dw) {
        const unsigned char* px = s + ((int64_t)dy * sw + dx) * 3;
        b = px[0]; g = px[1]; r = px[2];
    } else {
        float wx[kAreaMaxTaps], wy[kAreaMaxTaps];
        const int x0 = area_taps(dx, (double)sw / dw, sw, wx);
        const int y0 = area_taps(dy, (double)sh / dh, sh, wy);
        for (int j = 0; j < kAreaMaxTaps; j++) {
            if (wy[j] == 0.f) continue;
            const unsigned char* row = s + (int64_t)(y0 + j) * sw * 3;
            float rb = 0.f, rg = 0.f, rr = 0.f;
            for (int i = 0; i < kAreaMaxTaps; i++) {
                if (wx[i] == 0.f) continue;
                const unsigned char* px = row + (x0 + i) * 3;
                rb += wx[i] * px[0]; rg += wx[i] * px[1]; rr += wx[i] * px[2];
            }
            b += wy[j] * rb; g += wy[j] * rg; r += wy[j] * rr;
        }
    }
    const int64_t plane = (int64_t)dh * dw, o = (int64_t)dy * dw + dx;
    d[o] = r / 255.f;                 // RGB planes
    d[plane + o] = g / 255.f;
    d[2 * plane + o] = b / 255.f;
}

__global__ void __launch_bounds__(256)
disparity_u16_kernel(const float* __restrict__ disp, unsigned short* __restrict__ out, int64_t n, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = rintf(disp[i] * scale);           // cv::saturate_cast<ushort>(cvRound(x)): nearest even
    out[i] = (unsigned short)(v < 0.f ? 0.f : (v > 65535.f ? 65535.f : v));
}

}  // namespace rt
