// Correlation cost volume + soft-argmax in one pass ON THE MATRIX CORES: CostVolumePlugin(kCorrelation) followed by
// SoftargmaxPlugin of ResNet-18 2D (reference lib/kernels.cu:168-200 corrCostVolumeKernel,
// lib/softargmax_plugin.cpp:161-205; resnet18_2D_513x257_net.cpp:600-610).
//
//     cv[d, y, x] = sum_c L[c, y, x] * R[c, y, x - d]        (0 for x < d),   d in [0, D)
// is, for a block of 32 output pixels x0 .. x0+31 of one row, the band d = x - x' of the Gram matrix
//     G[x', x] = sum_c R[c, x'] * L[c, x],        x' in [x0 - 64, x0 + 31]:
// three 32 x 32 blocks, K = C <= 32.  With the 3-term fp16 split of conv_split.hip.h (fp32-class accuracy on the fp16
// pipe) that is 3 blocks x 2 chunks x 3 terms = 18 v_mfma_f32_32x32x16_f16 per 32 pixels instead of 32 x 48 x 32 fp32 FMAs
// on the vector ALU (what bounded the round-1 kernel).  Rows of the MFMA are R pixels, columns are L (= output)
// pixels, so a lane holds, for ITS output pixel, 16 of the 32 x' of each block: the soft-argmax runs per lane over its
// registers (online, block after block), the two half-waves that share an output pixel are combined with one
// cross-half shuffle, and nothing but the disparity is written.
//
// No LDS: the feature maps are channel-interleaved (C/4, H, pitch, 4) fp32 (the layout conv_s3_kernel writes), so the
// MFMA operand of a lane -- 8 consecutive channels of one pixel -- is two 16-byte loads.  One wave per (row, 32-pixel
// block); R blocks are shared with the neighbouring waves through L1/L2 only.
#pragma once
#include <type_traits>
#include "common.hip.h"
#include "conv_split.hip.h"

namespace rt {

struct CorrMfmaArgs {
    const float* left;      // (C/4, H, pitch, 4) per sample
    const float* right;
    float* out;             // (H, out_pitch) per sample
    int C, H, W, D;
    int in_pitch, out_pitch;
    int64_t in_bstride, out_bstride;     // elements
    int blocks_x;                        // ceil(W / 32)
    int batch;
    int out_slot;                        // 1: out is a plane (H, out_pitch); 4: lane 0 of the 16-byte pixel slots of a channel-interleaved
                                         // group (H, out_pitch, 4) -- the other three lanes are written as zeros
};

// half2 mode (round 6): fp16 feature maps, channel-interleaved (C/8, H, pitch, 8) -- the stored values ARE the operands: one 16-byte
// load per pixel and 16-channel half-chunk, ONE MFMA per block and chunk (fp32 accumulation of exact fp16 products, what the planar fp16
// kernel computes on the vector ALU), the map written as an fp16 plane.  Same Gram-band formulation and online soft-argmax as below.
template <bool ISMIN>
__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(4) corr_softargmax_mfma_f16_kernel(CorrMfmaArgs p) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t task = (int64_t)blockIdx.x * 4 + wv;
    const int per_row = p.blocks_x;
    const int64_t total = (int64_t)per_row * p.H * p.batch;
    if (task >= total) return;
    const int bx = (int)(task % per_row);
    const int y = (int)((task / per_row) % p.H);
    const int n = (int)(task / ((int64_t)per_row * p.H));
    const int x0 = bx * 32;
    const buf_rsrc rs_l = make_buf(elem_ptr(p.left, (int64_t)n * p.in_bstride, 2));
    const buf_rsrc rs_r = make_buf(elem_ptr(p.right, (int64_t)n * p.in_bstride, 2));
    const unsigned gstride = (unsigned)(8 * p.H * p.in_pitch) * 2u;        // bytes between groups of 8 channels
    const int nchunks = (p.C + 15) / 16;
    // operand of lane (pixel px, k-group kg), chunk c: channel group 2 c + kg of that pixel
    auto load_px = [&](const buf_rsrc& rs, int px, f16x8 (&v)[2]) {
        const bool ok = px >= 0 && px < p.W;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int g = 2 * c + kg;
            v[c] = __builtin_bit_cast(f16x8, buf_load4(rs, (ok && c < nchunks && 8 * g < p.C) ? (unsigned)g * gstride + (unsigned)(y * p.in_pitch + px) * 16u : kBufOOB, 0u));
        }
    };
    f16x8 lh[2], rh[3][2];
    load_px(rs_l, x0 + l31, lh);
#pragma unroll
    for (int j = 0; j < 3; j++) load_px(rs_r, x0 - 64 + 32 * j + l31, rh[j]);
    constexpr float kL2E = 1.44269504088896341f;
    const bool d_ge32 = p.D >= 32;
    float m_run = -1e30f, s_run = 0.f, w_run = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; c++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[j][c], lh[c], acc, 0, 0, 0);
        const int dj = l31 + 64 - 32 * j - 4 * kg;
        float v[16];
        float bm = -1e30f;
        auto gather = [&](auto lo_check, auto hi_check) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int cr = (r & 3) + 8 * (r >> 2);
                const float val = ISMIN ? -acc[r] : acc[r];
                bool ok = true;
                if constexpr (decltype(lo_check)::value) ok = ok && cr <= dj;             // d >= 0
                if constexpr (decltype(hi_check)::value) ok = ok && dj - cr < p.D;        // d < D
                v[r] = ok ? val : -1e30f;
                bm = fmaxf(bm, v[r]);
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (j == 2) {
            if (d_ge32) gather(T_{}, F_{}); else gather(T_{}, T_{});
        } else {
            gather(F_{}, T_{});
        }
        const float m_new = fmaxf(m_run, bm);
        const float sc = __builtin_amdgcn_exp2f((m_run - m_new) * kL2E);
        const float mL = m_new < -1e29f ? 0.f : -m_new * kL2E;
        float s_blk = 0.f, c_blk = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float cr = (float)((r & 3) + 8 * (r >> 2));
            const float e = __builtin_amdgcn_exp2f(fmaf(v[r], kL2E, mL));
            s_blk += e;
            c_blk = fmaf(e, cr, c_blk);
        }
        s_run = fmaf(s_run, sc, s_blk);
        w_run = fmaf(w_run, sc, fmaf((float)dj, s_blk, -c_blk));
        m_run = m_new;
    }
    const float m2 = __shfl_xor(m_run, 32), s2 = __shfl_xor(s_run, 32), w2 = __shfl_xor(w_run, 32);
    const float M = fmaxf(m_run, m2);
    const float e1 = fast_exp(m_run - M), e2 = fast_exp(m2 - M);
    const float s = s_run * e1 + s2 * e2, w = w_run * e1 + w2 * e2;
    const int x = x0 + l31;
    if (kg == 0 && x < p.W) {
        _Float16* o = reinterpret_cast<_Float16*>(p.out) + (int64_t)n * p.out_bstride;
        const _Float16 dv = (_Float16)(w / s);
        // out_slot 8: the map as lane 0 of the 16-byte pixel slots of a group of 8 fp16 channels (the 33rd channel of conv2D_1's input
        // when its concatenation stays interleaved, resnet18_2D_513x257_net.cpp:601-615), zeros in lanes 1 .. 7
        if (p.out_slot == 8) *reinterpret_cast<u32x4_t*>(o + ((int64_t)y * p.out_pitch + x) * 8) = u32x4_t{(unsigned)__builtin_bit_cast(unsigned short, dv), 0u, 0u, 0u};
        else o[(int64_t)y * p.out_pitch + x] = dv;
    }
}

__device__ static __forceinline__ void corr_split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
    const S3Split s0 = s3_split(a), s1 = s3_split(b);
#pragma unroll
    for (int j = 0; j < 4; j++) { hi[j] = s0.hi[j]; hi[4 + j] = s1.hi[j]; lo[j] = s0.lo[j]; lo[4 + j] = s1.lo[j]; }
}

// PLANAR (round 6): the feature maps are planar (C, H, pitch) fp32 -- rt_corr_softargmax on a client's NCHW maps; the operand of a lane is
// gathered as 8 dword loads, 128 contiguous bytes per half wave each (corr_mfma_planar_kernel below has the same gather).
template <bool ISMIN, bool PLANAR = false>
__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(4) corr_softargmax_mfma_kernel(CorrMfmaArgs p) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // task = (sample, row, 32-pixel block); a workgroup's 4 waves take 4 consecutive blocks of a row
    const int64_t task = (int64_t)blockIdx.x * 4 + wv;
    const int per_row = p.blocks_x;
    const int64_t total = (int64_t)per_row * p.H * p.batch;
    if (task >= total) return;
    const int bx = (int)(task % per_row);
    const int y = (int)((task / per_row) % p.H);
    const int n = (int)(task / ((int64_t)per_row * p.H));
    const int x0 = bx * 32;

    const buf_rsrc rs_l = make_buf(elem_ptr(p.left, (int64_t)n * p.in_bstride, 4));
    const buf_rsrc rs_r = make_buf(elem_ptr(p.right, (int64_t)n * p.in_bstride, 4));
    const unsigned gstride = (unsigned)(4 * p.H * p.in_pitch) * 4u;        // bytes between channel groups
    const int nchunks = (p.C + 15) / 16;

    // operand of lane (pixel px, k-group kg), chunk c: channel groups 4c + 2kg, 4c + 2kg + 1 of that pixel
    auto load_px = [&](const buf_rsrc& rs, int px, f32x4 (&v)[2][2]) {
        const bool ok = px >= 0 && px < p.W;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int g = 4 * c + 2 * kg + h;
                if constexpr (PLANAR) {
                    // channel 4 g + e: plane offset 4 g x (H x pitch) elements = g x gstride / 4 bytes ... gstride counts 4 planes
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        v[c][h][e] = buf_load(rs, (ok && 4 * g + e < p.C) ? (unsigned)g * gstride + (unsigned)e * (gstride >> 2) + (unsigned)(y * p.in_pitch + px) * 4u : kBufOOB, 0u);
                } else {
                    v[c][h] = buf_load4(rs, (ok && c < nchunks && 4 * g < p.C) ? (unsigned)g * gstride + (unsigned)(y * p.in_pitch + px) * 16u : kBufOOB, 0u);
                }
            }
    };

    f32x4 lraw[2][2], rraw[3][2][2];
    load_px(rs_l, x0 + l31, lraw);
#pragma unroll
    for (int j = 0; j < 3; j++) load_px(rs_r, x0 - 64 + 32 * j + l31, rraw[j]);

    f16x8 lh[2], ll[2];
#pragma unroll
    for (int c = 0; c < 2; c++) corr_split8(lraw[c][0], lraw[c][1], lh[c], ll[c]);

    // online soft-argmax state of this lane (its 16 of the 32 R pixels of each block), in base 2: e = 2^((v - m) log2 e).
    // VALU is what bounds this kernel (round 3: 55 VALU instructions per MFMA), so per element there is one fma for the split's
    // correction term, one compare + select for the disparity range, one fma + v_exp_f32 for the weight and two accumulations:
    //   * accumulator register r holds R pixel m = c_r + 4 kg, c_r = (r & 3) + 8 (r >> 2), i.e. disparity d = dj - c_r with the LANE
    //     value dj = l31 + 64 - 32 j - 4 kg: sum e d = dj sum e - sum e c_r, c_r a literal -- no int -> float conversion per element;
    //   * a masked element is -1e30 BEFORE the exponential, which makes its weight exactly 0 (2^-huge) without a second select;
    //   * d >= 0 can only fail in the last block (j = 2: d <= 31), d < D only in the first two when D >= 32 (j = 0: d >= 33).
    constexpr float kL2E = 1.44269504088896341f;
    const bool d_ge32 = p.D >= 32;                                  // wave-uniform
    float m_run = -1e30f, s_run = 0.f, w_run = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        f32x16 acc_m, acc_c;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc_m[r] = 0.f; acc_c[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            f16x8 rh, rl;
            corr_split8(rraw[j][c][0], rraw[j][c][1], rh, rl);
            acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, lh[c], acc_m, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl, lh[c], acc_c, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, ll[c], acc_c, 0, 0, 0);
        }
        const int dj = l31 + 64 - 32 * j - 4 * kg;
        float v[16];
        float bm = -1e30f;
        auto gather = [&](auto lo_check, auto hi_check) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int cr = (r & 3) + 8 * (r >> 2);
                const float val = fmaf(acc_c[r], ISMIN ? -kSplitInv : kSplitInv, ISMIN ? -acc_m[r] : acc_m[r]);
                bool ok = true;
                if constexpr (decltype(lo_check)::value) ok = ok && cr <= dj;             // d >= 0
                if constexpr (decltype(hi_check)::value) ok = ok && dj - cr < p.D;        // d < D
                v[r] = ok ? val : -1e30f;
                bm = fmaxf(bm, v[r]);
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (j == 2) {
            if (d_ge32) gather(T_{}, F_{}); else gather(T_{}, T_{});
        } else {
            gather(F_{}, T_{});
        }
        const float m_new = fmaxf(m_run, bm);
        const float sc = __builtin_amdgcn_exp2f((m_run - m_new) * kL2E);
        // (nothing valid yet: m_new is the mask value, and the ROUNDED product -m_new log2 e differs from the exact one inside the fma by
        // up to 1e23 -- an exponent of 0 instead keeps every masked weight at exactly 0)
        const float mL = m_new < -1e29f ? 0.f : -m_new * kL2E;
        float s_blk = 0.f, c_blk = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float cr = (float)((r & 3) + 8 * (r >> 2));
            const float e = __builtin_amdgcn_exp2f(fmaf(v[r], kL2E, mL));
            s_blk += e;
            c_blk = fmaf(e, cr, c_blk);
        }
        s_run = fmaf(s_run, sc, s_blk);
        w_run = fmaf(w_run, sc, fmaf((float)dj, s_blk, -c_blk));
        m_run = m_new;
    }
    // the other half-wave holds the other 16 R pixels of every block for the same output pixel
    const float m2 = __shfl_xor(m_run, 32), s2 = __shfl_xor(s_run, 32), w2 = __shfl_xor(w_run, 32);
    const float M = fmaxf(m_run, m2);
    const float e1 = fast_exp(m_run - M), e2 = fast_exp(m2 - M);
    const float s = s_run * e1 + s2 * e2, w = w_run * e1 + w2 * e2;
    const int x = x0 + l31;
    if (kg == 0 && x < p.W) {
        float* o = p.out + (int64_t)n * p.out_bstride;
        if (p.out_slot == 4) *reinterpret_cast<f32x4*>(o + ((int64_t)y * p.out_pitch + x) * 4) = f32x4{w / s, 0.f, 0.f, 0.f};
        else o[(int64_t)y * p.out_pitch + x] = w / s;
    }
}

// Stand-alone CostVolumePlugin(kCorrelation) on PLANAR fp32 NCHW maps (the reference plugin's own tensors, lib/kernels.cu:168-200;
// round 6): the same Gram band on the matrix cores, with
//   * the operand of a lane -- 8 consecutive channels of one pixel -- gathered as 8 dword loads, each of them 128 contiguous bytes per
//     half wave (a channel plane's row segment): 64 loads in flight per lane for C = 32, no LDS staging, no workgroup barrier;
//   * the band written as a volume: accumulator register r of lane (x, kg) is cv[d = dj - c_r, y, x] -- 32 planes per instruction, so
//     the wave transposes through ITS OWN part of the LDS (d, x) and stores plane rows, two per instruction (LDS operations of one
//     wave complete in order: no barrier);
//   * workgroups renumbered so that each XCD (its L2) gets a contiguous eighth of the rows: the cache lines that neighbouring blocks
//     and rows share are completed in one L2 instead of being evicted half written from two (measured: 14.6 -> 13.0 us).
// One wave per (row, 32 output pixels) -- the op is a single round of waves at batch 1, so occupancy is what it runs on: a wave that
// walks several blocks and keeps the older R blocks in registers (half the loads) measured 18-22 us (profiles/r06_corr_dev.txt).
// D <= 64 (three R blocks), C <= 32.  fp32 maps through the 3-term fp16 split: the product the engines' correlation runs on; small maps
// and RT_CONV_EXACT_FP32 keep corr_f32_kernel (cost_volume.hip.h).
struct CorrPlanarArgs {
    const float* left;      // (C, H, W) per sample
    const float* right;
    float* out;             // (D, H, W) per sample
    int C, H, W, D;
    int blocks_x;           // ceil(W / 32)
    int batch;
};

__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(4) corr_mfma_planar_kernel(CorrPlanarArgs p) {
    __shared__ __attribute__((aligned(16))) float sm_all[4][64 * 32];     // per wave: (D, 32)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned nb8 = gridDim.x >> 3;                                   // (the grid is a multiple of 8)
    const int64_t vb = (int64_t)(blockIdx.x & 7u) * nb8 + (blockIdx.x >> 3);
    const int64_t task = vb * 4 + wv;
    const int per_row = p.blocks_x;
    const int64_t total = (int64_t)per_row * p.H * p.batch;
    if (task >= total) return;
    const int bx = (int)(task % per_row);
    const int y = (int)((task / per_row) % p.H);
    const int n = (int)(task / ((int64_t)per_row * p.H));
    const int x0 = bx * 32;
    float* sm = sm_all[wv];

    const unsigned plane = (unsigned)(p.H * p.W) * 4u;                      // bytes per channel / disparity plane
    const buf_rsrc rs_l = make_buf(elem_ptr(p.left, (int64_t)n * p.C * p.H * p.W, 4));
    const buf_rsrc rs_r = make_buf(elem_ptr(p.right, (int64_t)n * p.C * p.H * p.W, 4));
    // operand of lane (pixel px, k-group kg), chunk c: channels 16 c + 8 kg .. + 7 of that pixel
    auto load_px = [&](const buf_rsrc& rs, int px, f32x4 (&v)[2][2]) {
        const unsigned at = (px >= 0 && px < p.W) ? (unsigned)(y * p.W + px) * 4u : kBufOOB;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int ch = 16 * c + 8 * kg + 4 * h + e;
                    v[c][h][e] = buf_load(rs, ch < p.C ? at + (unsigned)ch * plane : kBufOOB, 0u);
                }
    };
    f32x4 lraw[2][2], rraw[3][2][2];
    const bool blk0 = p.D > 33;                                            // block 0 holds d >= 33 only (wave-uniform)
    load_px(rs_l, x0 + l31, lraw);
#pragma unroll
    for (int j = 2; j >= 0; j--)
        if (j > 0 || blk0) load_px(rs_r, x0 - 64 + 32 * j + l31, rraw[j]);

    f16x8 lh[2], ll[2];
#pragma unroll
    for (int c = 0; c < 2; c++) corr_split8(lraw[c][0], lraw[c][1], lh[c], ll[c]);
#pragma unroll
    for (int j = 2; j >= 0; j--) {
        if (j == 0 && !blk0) break;
        f32x16 acc_m, acc_c;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc_m[r] = 0.f; acc_c[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            f16x8 rh, rl;
            corr_split8(rraw[j][c][0], rraw[j][c][1], rh, rl);
            acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, lh[c], acc_m, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl, lh[c], acc_c, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, ll[c], acc_c, 0, 0, 0);
        }
        // register r: R pixel c_r + 4 kg of the block, i.e. disparity d = dj - c_r of output pixel l31
        const int dj = l31 + 64 - 32 * j - 4 * kg;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int d = dj - ((r & 3) + 8 * (r >> 2));
            if (d >= 0 && d < p.D) sm[d * 32 + l31] = fmaf(acc_c[r], kSplitInv, acc_m[r]);
        }
    }
    wave_lds_sync();
    // plane rows out: lane (x, kg) stores disparities kg, kg + 2, ...
    const buf_rsrc rs_o = make_buf(elem_ptr(p.out, (int64_t)n * p.D * p.H * p.W, 4));
    const int x = x0 + l31;
    const unsigned oat = x < p.W ? (unsigned)(y * p.W + x) * 4u : kBufOOB;
    for (int d = kg; d < p.D; d += 2)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sm[d * 32 + l31]), rs_o, oat + (unsigned)d * plane, 0u, 0);
}

}  // namespace rt
