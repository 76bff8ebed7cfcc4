// The first Conv3D of the 3-D models over the DEFAULT cost volume, factored (round 4).
//
// The reference builds cv[d, 0:F] = L, cv[d, F:2F, y, x] = R[:, y, x - d] (0 for x < d) for d = 0 .. D-1 (lib/kernels.cu:50-97) and runs a
// 3x3x3 Conv3D over it (nvsmall_1025x321_net.cpp: conv3D_1; 64 -> 32 channels, 48 x 161 x 513: 438 GFLOP).  Rounds 2-4 never built the
// volume (rtConv3dDesc::cv_fold: the convolution gathers it from the two feature maps) but still multiplied all of it.  The volume is D
// shifted copies of TWO images, and a convolution commutes with a shift:
//
//   out[d, k, y, x] = b[k] + sum_{j in J(d)} ( (W_j^L * L)[k, y, x] + (W_j^R * R)~[k, y, x - d - j + 1] ),   J(d) = { j : 0 <= d + j - 1 < D }
//
// with W_j the depth tap j of the kernel split into its left / right channel halves, `*` a 2-D 3x3 convolution (zero padding) and ~ the
// same convolution evaluated one column beyond the image on the left (column -1 sees R[., 0] through the tap dx = +1).  So:
//   A_v  = conv3x3(L, sum_{j in v} W_j^L),   v = first / middle / last depth slice       one 2-D convolution F -> 3K
//   C'_j = conv3x3([0 | R], W_j^R)           (R with one zero column in front)           one 2-D convolution F -> 3K, width W + 1
//                                            (round 5: R itself with a left pad of 2 -- no copy; the plan's output is W + 2 wide, its last
//                                             column is never read)
//   T_v(u) = sum_{j in v} C'_j(u - j + 2),   u = x - d in [-2, W-1]                     one small pass (the three taps' shifts folded)
//   out[d](x) = act( b + A_v(d)(x) + T_v(d)(x - d) - [x = W-1] E[d] )
// Both convolutions run on the split-fp16 kernels of conv_split.hip.h (fp32-class accuracy); what is left per output voxel is one load
// and two adds.  NVSmall: 438 GFLOP -> 9 GFLOP + one pass that writes the 254 MB (fp16) / 507 MB (fp32) output.
//
// E: the one place where shift and convolution do not commute.  The volume ends at x = W-1, so at the last column the tap dx = +1 reads the
// zero padding of the VOLUME -- while C'_j there still sees the pixel R[., W + 1 - d - j] of the (longer) image.  E[d, k, y] is that
// contribution (a 3 x 1 convolution of one image column per depth tap), subtracted at x = W-1.
#pragma once
#include "common.hip.h"

namespace rt {

// tools/iso_conv3d.py fold: instrumented builds of the combining pass (-DRT_FOLD_ABL=<mask>: 1 no T loads, 2 no activation, 4 no stores)
#ifndef RT_FOLD_ABL
#define RT_FOLD_ABL 0
#endif
constexpr int kFoldAbl = RT_FOLD_ABL;
// (-DRT_FOLD_T_LDS=0: the combining pass reads every T slot from memory, as before round 5's last change)
#ifndef RT_FOLD_T_LDS
#define RT_FOLD_T_LDS 1
#endif
constexpr bool kFoldTLds = RT_FOLD_T_LDS != 0;
constexpr int kFoldLdsD = 128;          // depth slices up to which a block's T window (256 + D slots of 16 bytes per channel quad) is kept in LDS

struct FoldFactorArgs {
    const float* x;        // (N, 2F, H, W) fp32 planar: [left | right] feature maps
    const float* a;        // (N, 3K, H, W): A_first, A_middle, A_last
    const float* c;        // (N, 3K, H, W + 2): C'_0, C'_1, C'_2 (column W + 1 is not used)
    float* t;              // (N, 3K/4, H, W + 2, 4): T_first, T_middle, T_last at index u + 2, channel-interleaved in groups of 4 (16-byte slots)
    float* e;              // (N, D, K, H)
    const float* wedge;    // [j 3][dy 3][c F][k K]: w[k, j, F + c, dy, dx = 2]
    const float* bias;     // [K]
    void* y;               // output, depth-major: (N, D, K, H, W) planar or (N, D, K/G, H, W, G) interleaved (G = 4 fp32, 8 fp16)
    int F, K, D, H, W;
    int act, batch;
    int64_t x_bstride, a_bstride, c_bstride, t_bstride, e_bstride, y_bstride;    // elements
};

// T_v[n, v K + k, y, i] = sum_{j in v} C'_j[k, y, i - j]  (i = u + 2 in [0, W + 1]; C' indices outside [0, W] contribute nothing), written
// as (3K/4, H, W + 2, 4): one thread per (i, y, G channels) reads 3 G planar values (coalesced along i) and stores 3 x G / 4 16-byte slots
template <int G>
__global__ void __launch_bounds__(256) fold_t_kernel(FoldFactorArgs p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int KG = p.K / G;
    const int y = blockIdx.y, kg = blockIdx.z % KG, n = blockIdx.z / KG;
    if (i > p.W + 1) return;
    const int64_t cplane = (int64_t)p.H * (p.W + 2);
    const float* __restrict__ c = p.c + (int64_t)n * p.c_bstride + (int64_t)y * (p.W + 2);
    float tv[3][G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        float cj[3];
#pragma unroll
        for (int j = 0; j < 3; j++) cj[j] = (i - j >= 0 && i - j <= p.W) ? c[(int64_t)(j * p.K + kg * G + g) * cplane + i - j] : 0.f;
        // (summed in tap order 0, 1, 2)
        tv[0][g] = cj[1] + cj[2];
        tv[1][g] = (cj[0] + cj[1]) + cj[2];
        tv[2][g] = cj[0] + cj[1];
    }
#pragma unroll
    for (int v = 0; v < 3; v++)
#pragma unroll
        for (int q = 0; q < G / 4; q++) {
            float* __restrict__ t = p.t + (int64_t)n * p.t_bstride + ((((int64_t)v * (p.K / 4) + kg * (G / 4) + q) * p.H + y) * (p.W + 2) + i) * 4;
            *reinterpret_cast<f32x4*>(t) = f32x4{tv[v][4 * q], tv[v][4 * q + 1], tv[v][4 * q + 2], tv[v][4 * q + 3]};
        }
}

// E[n, d, k, y]: FOUR lanes per (4 output channels, y), each over a quarter of the input channels, summed with two DPP-free lane
// exchanges -- the loop is a chain of dependent loads and was latency-bound at 288 trips per thread (0.14 ms per batch of 8 for 71 MFLOP);
// grid (ceil(H * K / 256), D, N): 256 threads = 64 (k-quad, y) items x 4 channel quarters.
__global__ void __launch_bounds__(256) fold_edge_kernel(FoldFactorArgs p) {
    const int t = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    const int d = blockIdx.y, n = blockIdx.z;
    const int K4 = p.K / 4;
    const bool live = t < p.H * K4;
    const int k = live ? (t % K4) * 4 : 0, y = live ? t / K4 : 0;
    const float* __restrict__ r = p.x + (int64_t)n * p.x_bstride + (int64_t)p.F * p.H * p.W;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};       // two chains per output over this lane's channels (F is a multiple of 4)
    const int c0 = part * (p.F / 4), c1 = c0 + p.F / 4;
    for (int j = 0; j < 3; j++) {
        const int dz = d + j - 1, col = p.W + 1 - d - j;
        if (dz < 0 || dz >= p.D || d + j < 2 || col < 0) continue;                    // depth padding / the pixel lies outside the image as well
        for (int dy = 0; dy < 3; dy++) {
            const int iy = y + dy - 1;
            if (iy < 0 || iy >= p.H) continue;
            const float* __restrict__ wv = p.wedge + ((int64_t)(j * 3 + dy) * p.F) * p.K + k;
            const float* __restrict__ rv = r + (int64_t)iy * p.W + col;
            const int64_t cstep = (int64_t)p.H * p.W;
            for (int c = c0; c < c1; c += 2) {
                f32x4 w4[2];
                float rr[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int cc = c + u < c1 ? c + u : c;                              // (an odd quarter repeats its last channel with a zero factor)
                    w4[u] = *reinterpret_cast<const f32x4*>(wv + (int64_t)cc * p.K);
                    rr[u] = c + u < c1 ? rv[cc * cstep] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[u][q] = fmaf(w4[u][q], rr[u], acc[u][q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float v = acc[0][q] + acc[1][q];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        if (live && part == 0) p.e[(int64_t)n * p.e_bstride + ((int64_t)d * p.K + k + q) * p.H + y] = v;
    }
}

// out[d]: one thread per pixel and group of G output channels, all depths.  TOUT = float / _Float16; IL: (D, K/G, H, W, G) output.
// Round 4: bound by its vector-memory instructions (G 4-byte loads of T per 16 bytes stored: 2.2 TB/s).  Round 5: T is interleaved in
// 16-byte slots that are contiguous across the lanes of a wave -- a voxel group is G / 4 loads of 1 KB per wave (lane offset (x - d + 2) * 16
// bytes, out of range = masked = 0 by the buffer's own bounds check; the row of each 4-channel group as a wave-uniform scalar offset) --
// and the bias sits in the A registers.  U depth slices per trip have their loads issued together.  LASTCOL: the same arithmetic for the
// volume's last column only (x = W - 1, where the edge term E is subtracted), one thread per (row, depth slice, group, sample) -- the main launch covers
// x < W - 1, so its blocks of 256 pixels carry no edge logic and, for the image widths of the reference (W - 1 a multiple of 256), no idle
// lanes (the third block of a 513-pixel row had one live lane in 256).
template <typename TOUT, bool IL, int U, bool LASTCOL>
__global__ void __launch_bounds__(256) fold_combine_kernel(FoldFactorArgs p) {
    constexpr int G = sizeof(TOUT) == 2 ? 8 : 4;
    const int KG = p.K / G;
    int x, y, kg, n;
    int dl0 = 0, dl1 = p.D;                          // depth slices of this thread (LASTCOL: one)
    if constexpr (LASTCOL) {
        const int t = blockIdx.x * 256 + threadIdx.x;
        x = p.W - 1; y = t % p.H; dl0 = (t / p.H) % p.D; kg = (t / (p.H * p.D)) % KG; n = t / (p.H * p.D * KG);
        if (n >= p.batch) return;
        dl1 = dl0 + 1;
    } else {
        x = blockIdx.x * 256 + threadIdx.x;
        y = blockIdx.y; kg = blockIdx.z % KG; n = blockIdx.z / KG;
    }
    const bool live = LASTCOL || x < p.W - 1;
    const int64_t plane = (int64_t)p.H * p.W;
    const float* __restrict__ a = p.a + (int64_t)n * p.a_bstride + (int64_t)y * p.W + (live ? x : 0);
    const float* __restrict__ e = p.e + (int64_t)n * p.e_bstride + y;
    const buf_rsrc rs_t = make_buf(p.t + (int64_t)n * p.t_bstride);
    float av[3][G];                                  // A_v + bias
#pragma unroll
    for (int g = 0; g < G; g++) {
        const float b = p.bias[kg * G + g];
#pragma unroll
        for (int v = 0; v < 3; v++) av[v][g] = a[(int64_t)(v * p.K + kg * G + g) * plane] + b;
    }
    const unsigned trow = (unsigned)(p.W + 2) * 16u;                          // bytes of one row of a 4-channel T group (3 K planes of one sample: < 4 GB)
    // The middle slices (v = 1: all but the first and the last) read T(x - d) from ONE row per channel group, one slot further left per
    // slice: the block's window of that row -- 256 + D slots -- sits in LDS for the whole walk.  Read from memory instead, a wave asks for
    // the same kilobyte 47 times, shifted by 16 bytes, and with 32 waves on a CU the 32 KB L1 does not hold their rows: the loads went to the
    // L2 (2 x the bytes of the output) and were a third of the pass (`no T loads`: 0.68 -> 0.43 ms per batch of 8).
    constexpr int kTWin = 256 + kFoldLdsD;
    __shared__ f32x4 t_lds[LASTCOL ? 1 : (G / 4) * kTWin];
    const bool t_in_lds = !LASTCOL && kFoldTLds && p.D <= kFoldLdsD;           // uniform
    if constexpr (!LASTCOL) {
        if (t_in_lds) {
            const int i0 = blockIdx.x * 256 - p.D + 2;                        // slot j of the window = T index i0 + j; j = tid + D - d at slice d
#pragma unroll
            for (int q = 0; q < G / 4; q++) {
                const unsigned so = (unsigned)((1 * (p.K / 4) + kg * (G / 4) + q) * p.H + y) * trow;
                for (int j = threadIdx.x; j < 256 + p.D; j += 256) {
                    const int i = i0 + j;
                    t_lds[q * kTWin + j] = buf_load4(rs_t, (i >= 0 && i < p.W + 2) ? (unsigned)i * 16u : kBufOOB, so);
                }
            }
            __syncthreads();
        }
    }
    char* __restrict__ yb = static_cast<char*>(p.y) + (int64_t)n * p.y_bstride * sizeof(TOUT);
    const bool elu = p.act == 1;
    const int act = p.act;
    for (int d0 = dl0; d0 < dl1; d0 += U) {
        f32x4 tv[U][G / 4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int d = d0 + u < p.D ? d0 + u : p.D - 1;                                  // (a ragged last trip repeats the last slice)
            const int v = d == 0 ? 0 : (d == p.D - 1 ? 2 : 1);                             // wave-uniform
            const int i = x - d + 2;                                                        // index of T(x - d); < 0: the whole right half is masked (x < d - 2)
            const unsigned vo = (live && i >= 0) ? (unsigned)i * 16u : kBufOOB;
#pragma unroll
            for (int q = 0; q < G / 4; q++) {
                if (kFoldAbl & 1) tv[u][q] = f32x4{(float)d, 1.f, 2.f, (float)x};
                else if (!LASTCOL && t_in_lds && v == 1) tv[u][q] = t_lds[q * kTWin + (int)threadIdx.x + p.D - d];
                else tv[u][q] = buf_load4(rs_t, vo, (unsigned)((v * (p.K / 4) + kg * (G / 4) + q) * p.H + y) * trow);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int d = d0 + u;
            if (U > 1 && d >= dl1) break;                                                   // uniform
            const int v = d == 0 ? 0 : (d == p.D - 1 ? 2 : 1);
            float o[G];
#pragma unroll
            for (int q = 0; q < G / 4; q++)
#pragma unroll
                for (int g = 0; g < 4; g++) o[4 * q + g] = tv[u][q][g] + av[v][4 * q + g];
            if constexpr (LASTCOL) {
#pragma unroll
                for (int g = 0; g < G; g++) o[g] -= e[((int64_t)d * p.K + kg * G + g) * p.H];
            }
            if (kFoldAbl & 2) {
            } else if (elu) {
#pragma unroll
                for (int g = 0; g < G; g++) o[g] = elu_fast(o[g]);
            } else {
#pragma unroll
                for (int g = 0; g < G; g++) o[g] = apply_act_fast(o[g], act);
            }
            if (!live) continue;
            if constexpr (IL) {
                TOUT* dst = reinterpret_cast<TOUT*>(yb) + (((int64_t)d * KG + kg) * plane + (int64_t)y * p.W + x) * G;
                if constexpr (G == 8) {
                    u32x4_t w;
#pragma unroll
                    for (int q = 0; q < 4; q++) w[q] = pack_f16(o[2 * q], o[2 * q + 1]);
                    if (!(kFoldAbl & 4) || w[0] == 0x12345678u) *reinterpret_cast<u32x4_t*>(dst) = w;
                } else {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                }
            } else {
                TOUT* dst = reinterpret_cast<TOUT*>(yb) + ((int64_t)d * p.K + kg * G) * plane + (int64_t)y * p.W + x;
#pragma unroll
                for (int g = 0; g < G; g++) dst[(int64_t)g * plane] = (TOUT)o[g];
            }
        }
    }
}

}  // namespace rt
