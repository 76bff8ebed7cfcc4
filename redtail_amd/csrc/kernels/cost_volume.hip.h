// Cost-volume and soft-argmax kernels (HBM-bound part of the hot path).
//   corr_f32_kernel        replaces corrCostVolumeKernel            (reference lib/kernels.cu:168-200)
//   corr_f32_kernel<FUSED> = correlation + soft-argmax in one pass  (volume never written to HBM)
//   cost_volume_f32_kernel replaces costVolumeCopy{,Pad}Kernel      (reference lib/kernels.cu:50-97)
//   softargmax_f32_kernel  replaces SoftargmaxPlugin::enqueue's 5 passes (lib/softargmax_plugin.cpp:167-205)
#pragma once
#include "common.hip.h"

namespace rt {

// ------------------------------------------------------------------------------------------------
// Correlation cost volume, fp32 NCHW.
//
// Workgroup = 4 waves = one 4-row x 128-pixel tile of one sample, all disparities of one 4*DT block.
// Wave w owns the disparity group d in [d_base + w*DT, d_base + (w+1)*DT); lane = (row r = lane>>4,
// pixel group xg = lane&15) owns 8 consecutive pixels -> an 8 x DT register tile of accumulators.
// Per channel a lane needs L[8] and the sliding window R[8 + DT - 1]: both come from LDS as
// ds_read_b128.  The tiles are stored de-interleaved by 16-byte chunk parity (E = even chunks,
// O = odd chunks): a lane's window starts at chunk 2*xg + const, so within one instruction all
// lanes of a row read consecutive 16-byte slots of ONE parity array, and row regions are 512 B
// apart, i.e. every ds_read_b128 lane group is bank-conflict free (MI355X_MICROARCH.md, LDS table).
// Each L / R element is fetched from HBM once per tile (R has a D-wide halo), instead of D times
// as in the reference kernel ("not optimized", kernels.cu:45).
// ------------------------------------------------------------------------------------------------
constexpr int kCorrRY = 4;      // rows per tile
constexpr int kCorrTX = 128;    // pixels per tile row
constexpr int kCorrCC = 8;      // channels per LDS stage
constexpr int kCorrRS = 32;     // 16-byte slots per (channel, parity, row) region of the R tile

// H2 = fp16 tensors in TensorRT's NC2HW2 packing (reference corrCostVolumeFP16NC2HW2Kernel, lib/kernels.cu:203-250):
// a 4-byte slot holds channels (2i, 2i+1) of one pixel -- on output disparities (2i, 2i+1) -- so the pointers keep
// their 4-byte element type; arithmetic is fp32 as in the reference (:219-221), results are rounded once.
__device__ static __forceinline__ float half_of_slot(float slot, int hi) {
    const unsigned u = __builtin_bit_cast(unsigned, slot);
    const unsigned short h = (unsigned short)(hi ? (u >> 16) : (u & 0xffffu));
    return (float)__builtin_bit_cast(_Float16, h);
}
__device__ static __forceinline__ float pack_half2(float lo, float hi) {
    const unsigned a = __builtin_bit_cast(unsigned short, (_Float16)lo), b = __builtin_bit_cast(unsigned short, (_Float16)hi);
    return __builtin_bit_cast(float, a | (b << 16));
}

// T = storage type of the NCHW feature maps and of the fused soft-argmax output (float, or _Float16 when the
// executor runs in half2 mode); the NC2HW2 form (H2) always travels as 4-byte slots, T = float.
template <int DT, bool FUSED, bool ISMIN, bool H2 = false, typename T = float>
__global__ void __launch_bounds__(256)
corr_f32_kernel(const float* __restrict__ left, const float* __restrict__ right, float* __restrict__ out, int C,
                int H, int W, int D, int d_base, int64_t out_bstride, int in_pitch, int out_pitch) {
    constexpr int RY = kCorrRY, TX = kCorrTX, CC = kCorrCC, RS = kCorrRS;
    constexpr int DPAD = 4 * DT;          // disparities covered by the 4 waves of this workgroup
    constexpr int RW = DPAD + TX;         // R tile row width (floats), starts at x0 - d_base - DPAD + 1 - 1
    constexpr int NW = DT / 4 + 2;        // 16-byte chunks in a lane's R window (DT + 8 floats)
    static_assert(DT % 4 == 0 && DT >= 4 && DT <= 16, "DT must be 4, 8, 12 or 16");
    static_assert(RW / 8 <= RS, "R region too small");

    __shared__ __attribute__((aligned(16))) float sR[CC * 2 * RY * RS * 4];   // [c][parity][row][slot][4]
    __shared__ __attribute__((aligned(16))) float sL[CC * 2 * RY * 16 * 4];   // [c][parity][row][slot][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, dg = tid >> 6;
    const int r = lane >> 4, xg = lane & 15;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * RY, n = blockIdx.z;
    const int64_t plane = (int64_t)H * in_pitch;         // row pitch >= W (dense: == W)
    const int cslots = H2 ? (C + 1) / 2 : C;              // planes per sample
    const T* __restrict__ lb = reinterpret_cast<const T*>(left) + (int64_t)n * cslots * plane;
    const T* __restrict__ rb = reinterpret_cast<const T*>(right) + (int64_t)n * cslots * plane;

    float acc[8][DT];
#pragma unroll
    for (int pq = 0; pq < 8; pq++)
#pragma unroll
        for (int q = 0; q < DT; q++) acc[pq][q] = 0.f;

    // R tile float index i in [0, RW) <-> right-image column x0 - d_base - DPAD + i.
    // Lane window = floats [8*xg + DT*(3-dg), +DT+8)  ->  first chunk g0 = 2*xg + q0.
    const int q0 = (DT / 4) * (3 - dg);

    // Staging: wave dg stages channels {2*dg, 2*dg+1} of each chunk (channel = wave-uniform), so the
    // in-plane gather offsets depend on the lane only and are computed once; the loads of chunk i+1
    // are issued before the FMAs of chunk i and land in LDS after them (register prefetch).
    constexpr int CPW = CC / 4;
    constexpr int NKR = (RY * RW + 63) / 64, NKL = (RY * TX) / 64;
    const int wv = __builtin_amdgcn_readfirstlane(dg);
    int roff[NKR], loff[NKL];
#pragma unroll
    for (int k = 0; k < NKR; k++) {
        const int idx = lane + 64 * k;
        const int rr = idx / RW, i = idx - rr * RW;
        const int gx = x0 - d_base - DPAD + i, gyy = y0 + rr;
        roff[k] = (idx < RY * RW && gyy < H && gx >= 0 && gx < W) ? gyy * in_pitch + gx : -1;
    }
#pragma unroll
    for (int k = 0; k < NKL; k++) {
        const int idx = lane + 64 * k;
        const int rr = idx / TX, i = idx - rr * TX;
        const int gx = x0 + i, gyy = y0 + rr;
        loff[k] = (gyy < H && gx < W) ? gyy * in_pitch + gx : -1;
    }
    float pr_[CPW][NKR], pl_[CPW][NKL];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int j = 0; j < CPW; j++) {
            const int c = c0 + wv * CPW + j;
            const bool cok = c < C;
            const int cp = cok ? (H2 ? c >> 1 : c) : 0;
            const T* rp = rb + (int64_t)cp * plane;
            const T* lp = lb + (int64_t)cp * plane;
#pragma unroll
            for (int k = 0; k < NKR; k++) {
                const bool ok = cok & (roff[k] >= 0);
                float v = (float)rp[ok ? roff[k] : 0];
                if (H2) v = half_of_slot(v, c & 1);
                pr_[j][k] = ok ? v : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NKL; k++) {
                const bool ok = cok & (loff[k] >= 0);
                float v = (float)lp[ok ? loff[k] : 0];
                if (H2) v = half_of_slot(v, c & 1);
                pl_[j][k] = ok ? v : 0.f;
            }
        }
    };

    prefetch(0);
    for (int c0 = 0; c0 < C; c0 += CC) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CPW; j++) {
            const int c = wv * CPW + j;
#pragma unroll
            for (int k = 0; k < NKR; k++) {
                const int idx = lane + 64 * k;
                const int rr = idx / RW, i = idx - rr * RW;
                if (idx < RY * RW) sR[(((c * 2 + ((i >> 2) & 1)) * RY + rr) * RS + (i >> 3)) * 4 + (i & 3)] = pr_[j][k];
            }
#pragma unroll
            for (int k = 0; k < NKL; k++) {
                const int idx = lane + 64 * k;
                const int rr = idx / TX, i = idx - rr * TX;
                sL[(((c * 2 + ((i >> 2) & 1)) * RY + rr) * 16 + (i >> 3)) * 4 + (i & 3)] = pl_[j][k];
            }
        }
        __syncthreads();
        if (c0 + CC < C) prefetch(c0 + CC);
#pragma unroll 2
        for (int c = 0; c < CC; c++) {
            float lv[8], rw[DT + 8];
            const f32x4 l0 = *reinterpret_cast<const f32x4*>(&sL[(((c * 2 + 0) * RY + r) * 16 + xg) * 4]);
            const f32x4 l1 = *reinterpret_cast<const f32x4*>(&sL[(((c * 2 + 1) * RY + r) * 16 + xg) * 4]);
#pragma unroll
            for (int e = 0; e < 4; e++) { lv[e] = l0[e]; lv[4 + e] = l1[e]; }
#pragma unroll
            for (int t = 0; t < NW; t++) {
                const int g = q0 + t;   // wave-uniform
                const f32x4 v = *reinterpret_cast<const f32x4*>(&sR[(((c * 2 + (g & 1)) * RY + r) * RS + xg + (g >> 1)) * 4]);
#pragma unroll
                for (int e = 0; e < 4; e++) rw[4 * t + e] = v[e];
            }
            // pixel pq, disparity q (d = d_base + DT*dg + q): R column = window float pq - q + DT
#pragma unroll
            for (int pq = 0; pq < 8; pq++)
#pragma unroll
                for (int q = 0; q < DT; q++) acc[pq][q] = fmaf(lv[pq], rw[pq - q + DT], acc[pq][q]);
        }
    }

    const int gy = y0 + r;
    const int gx0 = x0 + 8 * xg;
    if (!FUSED) {
        // cv[n][d][y][x]; x < d comes out as 0 because the R tile is zero-filled left of column 0.
        if (gy < H) {
#pragma unroll
            for (int q = 0; q < DT; q += H2 ? 2 : 1) {
                const int d = d_base + DT * dg + q;
                if (d < D) {
                    float* o = out + (int64_t)n * out_bstride + (int64_t)(H2 ? d >> 1 : d) * H * out_pitch + (int64_t)gy * out_pitch + gx0;
#pragma unroll
                    for (int pq = 0; pq < 8; pq++)
                        if (gx0 + pq < W) o[pq] = H2 ? pack_half2(acc[pq][q], acc[pq][q + 1 < DT ? q + 1 : q]) : acc[pq][q];
                }
            }
        }
    } else {
        // per-lane soft-argmax partials over this wave's DT disparities, then combine the 4 waves via LDS
        __syncthreads();                       // all waves are done with sR
        float* red = sR;                       // [3][4 waves][RY*TX]
        constexpr int NPX = RY * TX;
#pragma unroll
        for (int pq = 0; pq < 8; pq++) {
            float m = -INFINITY;
#pragma unroll
            for (int q = 0; q < DT; q++) {
                const int d = d_base + DT * dg + q;
                const float v = ISMIN ? -acc[pq][q] : acc[pq][q];
                if (d < D) m = fmaxf(m, v);
            }
            float s = 0.f, ws = 0.f;
#pragma unroll
            for (int q = 0; q < DT; q++) {
                const int d = d_base + DT * dg + q;
                const float v = ISMIN ? -acc[pq][q] : acc[pq][q];
                if (d < D) {
                    const float e = expf(v - m);
                    s += e;
                    ws += e * (float)d;
                }
            }
            const int px = r * TX + 8 * xg + pq;
            red[(0 * 4 + dg) * NPX + px] = m;
            red[(1 * 4 + dg) * NPX + px] = s;
            red[(2 * 4 + dg) * NPX + px] = ws;
        }
        __syncthreads();
        for (int px = tid; px < NPX; px += 256) {
            float m = -INFINITY;
#pragma unroll
            for (int w4 = 0; w4 < 4; w4++) m = fmaxf(m, red[(0 * 4 + w4) * NPX + px]);
            float s = 0.f, ws = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; w4++) {
                const float mw = red[(0 * 4 + w4) * NPX + px];
                const float sc = mw == -INFINITY ? 0.f : expf(mw - m);
                s += red[(1 * 4 + w4) * NPX + px] * sc;
                ws += red[(2 * 4 + w4) * NPX + px] * sc;
            }
            const int yy = y0 + px / TX, xx = x0 + px % TX;
            if (yy < H && xx < W) reinterpret_cast<T*>(out)[(int64_t)n * out_bstride + (int64_t)yy * out_pitch + xx] = (T)(ws / s);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Default (concatenation) cost volume: (N,C,H,W) x2 -> (N,D,2C,H,W).  Pure store bandwidth:
// every lane reads its L value once and streams D + D coalesced stores.
// grid = (ceil(W/256), H, N*C)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
cost_volume_f32_kernel(const float* __restrict__ left, const float* __restrict__ right, float* __restrict__ out,
                       int C, int H, int W, int D) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int c = blockIdx.z % C, n = blockIdx.z / C;
    if (x >= W) return;
    const int64_t plane = (int64_t)H * W;
    const int64_t src = ((int64_t)n * C + c) * plane + (int64_t)y * W + x;
    const float lv = left[src];
    const float* rrow = right + src;
    float* ol = out + (((int64_t)n * D) * 2 * C + c) * plane + (int64_t)y * W + x;
    float* orr = ol + (int64_t)C * plane;
    const int64_t dstride = 2 * (int64_t)C * plane;
    for (int d = 0; d < D; d++) {
        ol[d * dstride] = lv;
        orr[d * dstride] = x >= d ? rrow[-d] : 0.f;
    }
}

// ... with 16-byte stores (round 5).  The kernel above issues 2 D four-byte stores per pixel and, for the odd image widths of the reference
// (513, 1025), leaves a third block of every row with one live lane: 3.5 TB/s of store traffic on NVSmall's volume.  Here a thread owns
// FOUR consecutive elements of a (sample, channel) plane, flattened over rows, positioned so that their address in the OUTPUT is 16-byte
// aligned: with C a multiple of 4 the offset of plane (d, channel) is a multiple of 4 elements plus (channel * plane) mod 4, the same for
// every disparity and for both halves -- `a` below.  The right half's four values at disparity d + 1 are those at d moved up by one
// element plus one new load; a pixel with x < d stores zero (kernels.cu:85-96).  Requires C % 4 == 0 and a 16-byte aligned output.
// grid = (ceil((H*W + 3) / 4 / 256), C, N)
__global__ void __launch_bounds__(256)
cost_volume_f32x4_kernel(const float* __restrict__ left, const float* __restrict__ right, float* __restrict__ out,
                         int C, int H, int W, int D) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int plane = H * W;                                                  // (< 2^31: checked by the host)
    const int a = (int)(((int64_t)c * plane) & 3);
    const int p0 = 4 * (int)(blockIdx.x * 256 + threadIdx.x) - a;             // first of this thread's four elements; may be < 0 (a > 0, first group)
    if (p0 >= plane) return;
    const int64_t src = ((int64_t)n * C + c) * plane;
    const float* __restrict__ lp = left + src;
    const float* __restrict__ rp = right + src;
    bool ok[4];
    int xs[4];
    f32x4 lv, w;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = p0 + j;
        ok[j] = p >= 0 && p < plane;
        xs[j] = ok[j] ? p % W : -1;
        lv[j] = ok[j] ? lp[p] : 0.f;
        w[j] = ok[j] ? rp[p] : 0.f;
    }
    const bool full = ok[0] && ok[3];
    const int64_t dstride = 2 * (int64_t)C * plane;
    float* ol = out + (((int64_t)n * D) * 2 * C + c) * plane + p0;           // (p0 < 0: only elements >= -p0 are touched)
    float* orr = ol + (int64_t)C * plane;
    for (int d = 0; d < D; d++) {
        // the element that enters the window at disparity d + 1: R[p0 - d - 1], needed only where x0 >= d + 1 (then the index is >= 0)
        const int pn = p0 - d - 1;
        const float nx = (pn >= 0 && pn < plane) ? rp[pn] : 0.f;
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = xs[j] >= d ? w[j] : 0.f;
        if (full) {
            *reinterpret_cast<f32x4*>(ol + d * dstride) = lv;
            *reinterpret_cast<f32x4*>(orr + d * dstride) = r;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ok[j]) { ol[d * dstride + j] = lv[j]; orr[d * dstride + j] = r[j]; }
        }
        w = f32x4{nx, w[0], w[1], w[2]};
    }
}

// ------------------------------------------------------------------------------------------------
// Soft-argmax / soft-argmin over D: one pass, online softmax in registers, coalesced along W.
// grid = (ceil(H*W/256), N)
// ------------------------------------------------------------------------------------------------
// T = float or _Float16 storage (the reference's plugin takes kHALF volumes in NCHW, softargmax_plugin.cpp:51-54,
// and widens them to fp32 around its cuDNN passes, :116-160); arithmetic is fp32 either way.
template <bool ISMIN, typename T = float>
__global__ void __launch_bounds__(256)
softargmax_f32_kernel(const T* __restrict__ vol, T* __restrict__ out, int D, int64_t HW) {
    const int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (px >= HW) return;
    const T* v = vol + (int64_t)n * D * HW + px;
    float m = -INFINITY, s = 0.f, ws = 0.f;
    for (int d0 = 0; d0 < D; d0 += 8) {
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float t = d0 + j < D ? (float)v[(int64_t)(d0 + j) * HW] : 0.f;
            xv[j] = d0 + j < D ? (ISMIN ? -t : t) : -INFINITY;
        }
        float cm = xv[0];
#pragma unroll
        for (int j = 1; j < 8; j++) cm = fmaxf(cm, xv[j]);
        const float mn = fmaxf(m, cm);
        const float sc = m == -INFINITY ? 0.f : expf(m - mn);
        s *= sc;
        ws *= sc;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float e = expf(xv[j] - mn);      // exp(-inf) = 0 for the masked tail
            s += e;
            ws += e * (float)(d0 + j);
        }
        m = mn;
    }
    out[(int64_t)n * HW + px] = (T)(ws / s);
}

}  // namespace rt
