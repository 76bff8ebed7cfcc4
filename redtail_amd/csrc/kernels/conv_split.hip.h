// fp32 convolutions on the fp16 matrix pipe of gfx950: 3-term split ("fp16x3"), fp32 tensors in and out.
//
// Why.  v_mfma_f32_32x32x2_f32 / 16x16x4_f32 run at the fp32 VECTOR rate (157 TFLOP/s, and every other VALU
// instruction of the wave is paid in the same issue slots); v_mfma_f32_32x32x16_f16 runs 16x faster (2.5 PFLOP/s).
// An fp32 value is the exact sum of two fp16 values to 22 bits:
//     x = xh + xl * 2^-11,   xh = fp16(x),   xl = fp16((x - xh) * 2^11)         (same for a weight w = wh + wl * 2^-11)
// so      x * w = xh*wh + (xh*wl + xl*wh) * 2^-11 + O(2^-22 |x w|)
// and each of the three products of fp16 operands is EXACT in the fp32 accumulator of the matrix core (11 x 11
// bits).  Three MFMAs at the fp16 rate replace one at the fp32 rate (5.3x the fp32 matrix peak), the error of a
// whole convolution against an fp64 evaluation is the one of an fp32 fmaf chain (measured on the reference's trained
// ResNet-18 2D: 1.1e-7 on the disparity against 2.8e-7 for the torch fp32 oracle; tests/test_split_parity.py), and the
// layer turns from MFMA/VALU-issue-bound into HBM-bound.  The cross terms go to their own accumulator (they are 2^11
// smaller than the main term), which is scaled and added once, in the epilogue.  The low parts are pre-scaled by 2^11
// so that they are normal fp16 numbers whenever the value itself is (|x| >= 6.1e-5): nothing depends on fp16
// subnormals.  Domain: |x| < 65504 (activations of the Stereo DNN graphs are O(1..100)); larger inputs give inf/NaN
// -- loudly -- and RT_CONV_EXACT_FP32=1 keeps the plans on the fp32 kernels (conv_wino.hip.h, conv_mfma.hip.h).
//
// conv_s3p_kernel: PERSISTENT kernel for the layer that dominates ResNet-18 2D (reference
// resnet18_2D_513x257_net.cpp:48-719: 3x3, stride 1, Cin <= 32, Cout <= 32; 36 of the 49 launches, 83 % of the FLOPs).
//   * grid = one workgroup (8 waves, two per SIMD) per CU; a workgroup walks a contiguous range of 8-row x 32-pixel tiles.
//   * the split weights of the whole layer (9 taps x 32 ci x 32 co x {hi, lo} fp16 = 36 KB) are loaded into LDS once.
//   * tile i+1 is gathered from global memory into registers while tile i is multiplied out of LDS (double-buffered
//     patch image, ONE barrier per tile), its residual tile is requested before the MFMAs and consumed after them.
//   * the patch image holds the split input: per pixel 32 x fp16 hi | 32 x fp16 lo | 16 B pad = 144 B, a pixel stride
//     of 36 banks, so every ds_read_b128 lane group of a B-operand fetch touches all 64 banks exactly once.
//   * per tap and 16-channel chunk: 2 B fetches + 2 A fetches (16 B per lane each) feed 3 MFMAs.
// Tensors: channel-interleaved (C/4, H, pitch, 4) or planar, per tensor (conv_wino.hip.h describes the layout).
#pragma once
#include <type_traits>
#include "common.hip.h"
#include "conv_mfma.hip.h"

namespace rt {

constexpr float kSplitScale = 2048.f;            // 2^11: low parts are stored scaled
constexpr float kSplitInv = 1.f / 2048.f;

struct S3Split {
    f16x4 hi, lo;
};
// 4 fp32 values (4 channels of a pixel) -> 4 fp16 high parts + 4 scaled fp16 low parts
//     hi = fp16(v),   lo = fp16((v - float(hi)) * 2^11)        (both conversions round to nearest even; v - float(hi) is exact)
// On the device: 5 VALU instructions per PAIR of values instead of the 11 the compiler makes of the definition -- one packed conversion,
// v_fma_mix_f32 (reads the fp16 half directly: v - float(hi) = fma(hi, -1, v)) and v_fma_mixlo/hi_f16 (fp32 product rounded once into
// a half of the destination: the scaling by a power of two is exact, so it is the same single rounding).  Checked bit for bit against
// the definition over ALL 2^32 inputs: tools/micro/split_mix.hip.
__device__ static __forceinline__ S3Split s3_split(f32x4 v) {
    S3Split s;
#ifdef HIPEMU
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const _Float16 h = (_Float16)v[j];
        s.hi[j] = h;
        s.lo[j] = (_Float16)((v[j] - (float)h) * kSplitScale);
    }
#else
    unsigned h[2], l[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[j]) : "v"(v[2 * j]), "v"(v[2 * j + 1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h[j]), "v"(v[2 * j]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h[j]), "v"(v[2 * j + 1]));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l[j]) : "v"(r0), "v"(kSplitScale));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l[j]) : "v"(r1), "v"(kSplitScale));
    }
    s.hi = __builtin_bit_cast(f16x4, u32x2_t{h[0], h[1]});
    s.lo = __builtin_bit_cast(f16x4, u32x2_t{l[0], l[1]});
#endif
    return s;
}

template <int NW>
struct S3PCfg {
    static constexpr int TY = NW, TX = 32;                 // one output row of 32 pixels per wave
    static constexpr int PR = TY + 2, PC = TX + 2, NPIX = PR * PC;
    static constexpr int PXB = 144;                        // bytes per patch pixel in LDS (64 hi + 64 lo + 16 pad)
    static constexpr int NT = 64 * NW;
    static constexpr int NSLOT = NPIX * 8;                 // (pixel, group of 4 channels) gather slots per tile
    static constexpr int NKX = (NSLOT + NT - 1) / NT;      // ... per thread
    static constexpr int W_SLOTS = 9 * 2 * 2 * 2 * 32;     // 16-byte slots: [tap][chunk][hi/lo][k-group][co]
    static constexpr int NK_W = (W_SLOTS + NT - 1) / NT;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
};

// XIL / YIL: input / output tensor channel-interleaved (C/4, H, pitch, 4); the residual's layout is p.r_il8.
template <int NW, bool XIL, bool YIL>
__global__ void __launch_bounds__(64 * NW) RT_WAVES_PER_EU(NW / 4) conv_s3p_kernel(ConvArgs p) {
    using Cfg = S3PCfg<NW>;
    constexpr int TY = Cfg::TY, TX = Cfg::TX, PC = Cfg::PC, NPIX = Cfg::NPIX, PXB = Cfg::PXB, NT = Cfg::NT;
    constexpr int NKX = Cfg::NKX, NK_W = Cfg::NK_W;

    __shared__ __attribute__((aligned(16))) f32x4 sW[Cfg::W_SLOTS];
    __shared__ __attribute__((aligned(16))) char sX[2][NPIX * PXB];

    const int tid = threadIdx.x;
#ifdef RT_KERNEL_TIMING
    unsigned long long* dbgp = p.dbg ? p.dbg + (size_t)blockIdx.x * 16 : nullptr;
    int dbi = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    RT_TSTAMP();                                  // 0: start
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's tiles: a contiguous range, contiguous per XCD (workgroup b runs on XCD b % 8) ----------------
    const int G = gridDim.x;
    int v = blockIdx.x;
    if ((G & 7) == 0) v = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int tiles_y = (p.Ho + TY - 1) / TY;
    const int per_img = p.tiles_x * tiles_y;
    const int T = per_img * p.batch;
    const int t_begin = (int)((int64_t)v * T / G), t_end = (int)((int64_t)(v + 1) * T / G);
    if (t_begin >= t_end) return;

    // ---- weights: the whole layer, once ---------------------------------------------------------------------------------
    {
        const buf_rsrc rs_w = make_buf(p.w);
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + NT * k;
            if (idx < Cfg::W_SLOTS) sW[idx] = buf_load4(rs_w, (unsigned)idx * 16u, 0u);
        }
    }

    // ---- gather slots of this thread: (patch pixel, group of 4 channels), fixed for the whole kernel -------------------
    int s_pr[NKX], s_pc[NKX], s_lds[NKX], s_g[NKX];
    unsigned s_goff[NKX];                    // byte offset of the group's first channel plane, kBufOOB = channels do not exist
    const unsigned cs_x = (unsigned)p.x_cstride;
#pragma unroll
    for (int k = 0; k < NKX; k++) {
        const int idx = tid + NT * k;
        const int g = idx / NPIX, pix = idx - g * NPIX;
        const int pr = pix / PC;
        s_pr[k] = pr;
        s_pc[k] = pix - pr * PC;
        s_g[k] = g;
        const bool own = idx < Cfg::NSLOT && 4 * g < p.cin_real;
        s_goff[k] = own ? (unsigned)(4 * g) * cs_x * 4u : kBufOOB;
        s_lds[k] = idx < Cfg::NSLOT ? pix * PXB + g * 8 : -1;
    }

    f32x4 rin[NKX];
    auto gather = [&](int tile) {
        const int n = tile / per_img, rem = tile - n * per_img;
        const int ty0 = (rem / p.tiles_x) * TY, tx0 = (rem % p.tiles_x) * TX;
        const buf_rsrc rs = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, 4));
#pragma unroll
        for (int k = 0; k < NKX; k++) {
            const int iy = ty0 - p.pad_y + s_pr[k], ix = tx0 - p.pad_x + s_pc[k];
            const bool in = s_goff[k] != kBufOOB && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            if constexpr (XIL) {
                rin[k] = buf_load4(rs, in ? s_goff[k] + (unsigned)(iy * p.x_pitch + ix) * 16u : kBufOOB, 0u);
            } else {
                const unsigned vo = in ? s_goff[k] + (unsigned)(iy * p.x_pitch + ix) * 4u : kBufOOB;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // channels beyond cin_real inside the last group: the plan's zero-padded weights multiply whatever is
                    // read, so they must read zeros -> out of range
                    const bool cj = in && 4 * s_g[k] + j < p.cin_real;
                    rin[k][j] = buf_load(rs, cj ? vo + (unsigned)j * cs_x * 4u : kBufOOB, 0u);
                }
            }
        }
    };
    auto stage = [&](int buf) {
        char* base = sX[buf];
#pragma unroll
        for (int k = 0; k < NKX; k++) {
            if (s_lds[k] < 0) continue;
            const S3Split s = s3_split(rin[k]);
            *reinterpret_cast<f16x4*>(base + s_lds[k]) = s.hi;
            *reinterpret_cast<f16x4*>(base + s_lds[k] + 64) = s.lo;
        }
    };

    // ---- per-lane constants of the contraction and of the epilogue ------------------------------------------------------
    const int b_off = (wv * PC + l31) * PXB + kg * 16;       // B operand: pixel (row wv + r, col l31 + s), k-group kg
    const int a_off = kg * 32 + l31;                         // A operand: output channel l31, k-group kg (16-byte slots)
    f32x4 bias4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bias4[q] = *reinterpret_cast<const f32x4*>(p.bias + 8 * q + 4 * kg);
    const int cs32 = (int)p.y_cstride, rs32 = (int)p.r_cstride;
    const bool r_il = p.r_il8 != 0, has_r = p.resid != nullptr;
    const int act = p.act;

    gather(t_begin);
    RT_TSTAMP();                                  // 1: weights in LDS, first gather issued
    for (int tile = t_begin; tile < t_end; tile++) {
        const int buf = (tile - t_begin) & 1;
        stage(buf);
        RT_TSTAMP();                              // 2, 6, 10: tile gathered, split and written to LDS
        __syncthreads();       // tile staged; every wave is done with the other buffer (it finished the previous tile)
        RT_TSTAMP();                              // 3, 7, 11: barrier passed
        if (tile + 1 < t_end) gather(tile + 1);

        const int n = tile / per_img, rem = tile - n * per_img;
        const int oy = (rem / p.tiles_x) * TY + wv, ox = (rem % p.tiles_x) * TX + l31;
        const bool inb = oy < p.Ho && ox < p.Wo;
        // residual tile: requested now, consumed after the MFMAs
        f32x4 rr[4];
        if (has_r) {
            const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, (int64_t)n * p.r_bstride + p.y_off, 4));
            if (r_il) {
                const unsigned vo = inb ? (unsigned)((oy * p.y_ystride + ox) * 4 + 4 * kg * rs32) * 4u : kBufOOB;
#pragma unroll
                for (int q = 0; q < 4; q++) rr[q] = buf_load4(rs_r, (8 * q + 4 * kg < p.Cout) ? vo : kBufOOB, (unsigned)(8 * q * rs32) * 4u);
            } else {
                const unsigned vo = inb ? (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * kg * rs32) * 4u : kBufOOB;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        rr[q][e] = buf_load(rs_r, (8 * q + 4 * kg + e < p.Cout) ? vo : kBufOOB, (unsigned)((8 * q + e) * rs32) * 4u);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) rr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        // ---- 9 taps x 2 chunks x 3 MFMAs -----------------------------------------------------------------------------------
        f32x16 acc_m, acc_c;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc_m[r] = 0.f; acc_c[r] = 0.f; }
        const char* xb = sX[buf] + b_off;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int r = t / 3, s = t % 3;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const char* bp = xb + (r * PC + s) * PXB + c * 32;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(bp);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + 64);
                const f16x8 ah = __builtin_bit_cast(f16x8, sW[a_off + ((t * 2 + c) * 2 + 0) * 64]);
                const f16x8 al = __builtin_bit_cast(f16x8, sW[a_off + ((t * 2 + c) * 2 + 1) * 64]);
                acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc_m, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc_c, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc_c, 0, 0, 0);
            }
        }

        RT_TSTAMP();                              // 4, 8, 12: MFMAs issued
        // ---- epilogue: y = main + cross * 2^-11 + bias + residual, activation, stores --------------------------------------
        // accumulator register 4q + e of a lane = channel 8q + 4kg + e of pixel l31
        const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride + p.y_off, 4));
        auto epilogue = [&](auto ACT) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    o[e] = apply_act_fast(fmaf(acc_c[4 * q + e], kSplitInv, acc_m[4 * q + e]) + (bias4[q][e] + rr[q][e]), decltype(ACT)::value);
                if constexpr (YIL) {
                    const unsigned vo = (inb && 8 * q + 4 * kg < p.Cout) ? (unsigned)((oy * p.y_ystride + ox) * 4 + 4 * kg * cs32) * 4u : kBufOOB;
                    buf_store4(o, rs_y, vo, (unsigned)(8 * q * cs32) * 4u);
                } else {
                    const unsigned vo = inb ? (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * kg * cs32) * 4u : kBufOOB;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        buf_store(o[e], rs_y, (8 * q + 4 * kg + e < p.Cout) ? vo : kBufOOB, (unsigned)((8 * q + e) * cs32) * 4u);
                }
            }
        };
        if (act == 1) epilogue(std::integral_constant<int, 1>{});
        else if (act == 2) epilogue(std::integral_constant<int, 2>{});
        else epilogue(std::integral_constant<int, 0>{});
        RT_TSTAMP();                              // 5, 9, 13: stores issued
    }
#ifdef RT_KERNEL_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    if (dbgp && tid == 0) { dbgp[14] = __builtin_amdgcn_s_memtime(); dbgp[15] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
}

}  // namespace rt

namespace rt {

// -----------------------------------------------------------------------------------------------------------------------
// conv_s3_kernel: the general form -- every window / gather description conv_mfma_f32_kernel serves (3x3 stride 1 / 2,
// the 1x1 .. 2x2 windows of merged transposed-convolution phases with their ZSlice tables, the (D*C)-merged gather
// table of Conv3DPlugin / Conv3DTransposePlugin; reference lib/conv3d_plugin.cpp:187-216,
// conv3d_transpose_plugin.cpp:205-243) on the same 3-term fp16 split.  One workgroup = 4 rows x 32 pixels x 32 output
// channels, 4 waves; per chunk of 16 gathered channels:
//   global (fp32) --registers: loads of chunk i+1 fly under the MFMAs of chunk i--> split --> LDS patch image
//       [row][col] pixels x (16 x fp16 hi | 16 x fp16 lo | 16 B pad = 80 B: 20 banks, conflict-free ds_read_b128)
//   weights (split on the host) [chunk][tap][hi/lo][k-group][co][8 halfs]: a straight 16-byte copy into LDS
//   per tap 2 + 2 LDS fetches, 3 MFMAs.
// Staging: wave w gathers channel group w (4 channels) of the chunk for every patch pixel, so the plane offsets come
// from the gather table as scalars.  Stride 2: even and odd patch columns are stored apart, so that the lanes of a B
// fetch (pixels 2l + s) still read consecutive LDS pixels.
// -----------------------------------------------------------------------------------------------------------------------
template <int KH, int KW, int S, int NW_ = 4>
struct S3Cfg {
    static constexpr int NW = NW_, TY = NW_, TX = 32, CC = 16, TAPS = KH * KW, NT = 64 * NW_;       // one output row per wave
    static constexpr int PR = (TY - 1) * S + KH, PC = (TX - 1) * S + KW;
    static constexpr int PCH = (PC + 1) / 2;                       // stride 2: columns per parity
    static constexpr int PCL = S == 2 ? 2 * PCH : PC;              // LDS columns per patch row
    // staging: wave w gathers channel group w & 3 (4 channels of the chunk) for part w >> 2 of the patch pixels
    static constexpr int NPART = NW / 4;
    static constexpr int NPIX = PR * PC, NKP = ((NPIX + NPART - 1) / NPART + 63) / 64;   // patch pixels per lane
    static constexpr int PXB = 80;
    static constexpr int W_SLOTS = TAPS * 2 * 2 * 32;              // 16-byte slots of one chunk's weights
    static constexpr int NK_W = (W_SLOTS + NT - 1) / NT;
    static constexpr int IN_BYTES = PR * PCL * PXB, GRP_BYTES = IN_BYTES + W_SLOTS * 16;   // LDS of one contraction group (dynamic)
    // split-K: at most KS_MAX groups of NW waves per workgroup (the register budget below: 4 resp. 2 waves per SIMD)
    static constexpr int KS_MAX = NW_ != 4 ? 1 : (S == 2 ? 2 : 4);
    static constexpr int max_groups(int want) {
        int ks = want < 1 ? 1 : (want > KS_MAX ? KS_MAX : want);
        while (ks > 1 && (ks == 3 || ks * GRP_BYTES < (ks - 1) * NW * 64 * 16 * 4)) ks--;   // partial sums travel through the operand images
        return ks;
    }
    static_assert(IN_BYTES % 16 == 0, "weight image alignment");
    static_assert(S == 1 || S == 2, "stride 1 or 2");
};

// register budget: 4 waves per SIMD (4 workgroups per CU) for the stride-1 windows, whose patch + weight images are
// <= 35 KB; the stride-2 patch (66 KB) allows two workgroups per CU anyway.  Occupancy is what this kernel lives on --
// measured on the 32->32 @629x185 layer (round 2): fetching the operands of tap t+1 before the MFMAs of tap t (two operand sets,
// +16 registers -> 3 waves per SIMD) ran 16.3 us against 14.5 us, 1978 against 2110 pairs/s; at 4 waves it spilled 14 registers.
#ifndef RT_S3_WAVES
#define RT_S3_WAVES(S) ((S) == 2 ? 2 : 4)
#endif
// NW = 8 (3x3 stride 1 only): 8-row tiles, 8 waves -- the chunk's weights are fetched and staged once per 8 rows instead of
// once per 4 and the patch halo shrinks from 6/4 to 10/8 rows: 29 % fewer bytes through the CU's vector-memory path, which
// is what bounds the layer (profiles/README.md, round 2)
// TIN / TOUT (planar tensors only): storage type of the input and of the output + residual -- float, or _Float16 for the
// 3-D tensors of half2 mode (TensorRT setHalf2Mode with an fp16 weight file, reference sample_app/main.cpp:256-262; the
// reference's Conv3D plugins convert fp32 <-> fp16 around cuDNN, lib/conv3d_plugin.cpp:247-274).  An fp16 input IS its own
// high part (low part 0) and fp16 weights are theirs (p.w_exact), so the corresponding MFMAs are skipped: 3 -> 2 -> 1
// matrix instructions per tap, fp32 accumulation throughout.
// Split-K (the low-resolution layers): a launch with KS * 64 * NW threads per workgroup (and KS * GRP_BYTES of dynamic LDS) runs KS
// groups of NW waves on the workgroup's tile; group g contracts chunks g, g + KS, ... from its own LDS images and the groups' fp32
// partial sums are added in a fixed order (1, 2, .. after group 0's) before the epilogue.  A 128->128 layer at 158 x 47 is 240
// workgroups of 8 chunks: with ONE wave per SIMD every LDS fetch and every dependent MFMA pair of a tap is exposed (3.3 k cycles
// per chunk for 27 MFMAs, tools/time_tail.py); four groups put four independent chunk streams on each SIMD.  KS is a launch
// parameter, not a template parameter: a second instantiation would start with cold instruction caches once per pair (+6 us on
// its first launch, tools/sync_trace.py), which is what the split saves.  KS follows from the per-sample geometry only
// (rt_capi.hip), never from the batch, so a result does not depend on how many samples a launch carries.
template <int KH, int KW, int S, bool XIL, bool YIL, int NW = 4, typename TIN = float, typename TOUT = float>
__global__ void __launch_bounds__(64 * NW * (S3Cfg<KH, KW, S, NW>::KS_MAX)) RT_WAVES_PER_EU(RT_S3_WAVES(S)) conv_s3_kernel(ConvArgs p) {
    using Cfg = S3Cfg<KH, KW, S, NW>;
    constexpr bool XF16 = std::is_same<TIN, _Float16>::value, YF16 = std::is_same<TOUT, _Float16>::value;
    constexpr unsigned ESX = Io<TIN>::ES, ESY = Io<TOUT>::ES;
    // interleaved INPUT tensors are fp32 here (fp16 ones go through conv_f16mma_kernel); an interleaved OUTPUT may also be fp16,
    // (K/8, H, W, 8): the last layer of a feature tower of a 3-D model in half2 mode (fp32 tower tensors in, fp16 feature map out)
    // and the first Conv3D when the feature maps stay fp32
    static_assert(!XIL || !XF16, "interleaved fp16 input: conv_f16mma_kernel");
    constexpr int TY = Cfg::TY, TX = Cfg::TX, CC = Cfg::CC, TAPS = Cfg::TAPS, PC = Cfg::PC, PCL = Cfg::PCL, PXB = Cfg::PXB;
    constexpr int NKP = Cfg::NKP, NK_W = Cfg::NK_W, NT = Cfg::NT;

    constexpr int IN_BYTES = Cfg::IN_BYTES, GRP_BYTES = Cfg::GRP_BYTES;
#ifdef HIPEMU
    HIP_DYNAMIC_SHARED(char, smem)
#else
    extern __shared__ __attribute__((aligned(16))) char smem[];                       // KS * GRP_BYTES (rt_capi.hip: launch_s3)
#endif

    const int tid_wg = threadIdx.x;
    const int KS = Cfg::KS_MAX == 1 ? 1 : __builtin_amdgcn_readfirstlane((int)blockDim.x / NT);
    const int kgi = Cfg::KS_MAX == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid_wg / NT);       // contraction group of this wave
    const int tid = tid_wg - kgi * NT;                                                // thread index inside the group
    char* const sIn = smem + kgi * GRP_BYTES;
    f32x4* const sW = reinterpret_cast<f32x4*>(smem + kgi * GRP_BYTES + IN_BYTES);
#ifdef RT_KERNEL_TIMING
    unsigned long long* dbgp = (p.dbg && kgi == 0) ? p.dbg + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 : nullptr;
    int dbi = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    RT_TSTAMP();                                  // 0: start
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    RT_WG_TILE_NB(p, tile, zi, n, nblk)
    const int tx0 = (tile % p.tiles_x) * TX;
    const int ty0 = (tile / p.tiles_x) * TY;
    const char* __restrict__ xb = elem_ptr(p.x, (int64_t)n * p.x_bstride, ESX);
    const int nchunks = p.CinPad / CC;
    int pad_y = p.pad_y, pad_x = p.pad_x, Ho = p.Ho, Wo = p.Wo, ch_row = zi;
    int64_t y_off = p.y_off + (int64_t)zi * p.y_zstride, w_off = 0, r_off = y_off, r_off_il8 = y_off;
    // interleaved fp32 tensors (groups of 4 channels, 16-byte pixel slots): the PIXEL part of an offset counts 4 elements, the plane part
    // (depth slice of a 3-D tensor) is what it is.  p.y_off is a pixel offset (2-D plans), zi * y_zstride a plane offset (Conv3D).
    int64_t y_off_il4 = 4 * p.y_off + (int64_t)zi * p.y_zstride, r_off_il4 = y_off_il4;
    unsigned tap_mask = ~0u;
    if (p.zs) {
        const ZSlice z = p.zs[zi];
        pad_y = z.pad_y; pad_x = z.pad_x; Ho = z.Ho; Wo = z.Wo; ch_row = z.ch_row; y_off = z.y_off; w_off = z.w_off;
        r_off = z.r_off; r_off_il8 = z.r_off_il8;
        y_off_il4 = 4 * z.y_off;                   // phases of a 2-D transposed convolution (interleaved 3-D outputs of phase launches: fp16 only)
        r_off_il4 = z.r_off_il4;
        tap_mask = z.tap_mask;
    }
    const int act = p.act;

    // ---- staging: wave w gathers channels 4w .. 4w+3 of each chunk for every patch pixel -----------------------------
    const int sg = wv & 3, spart = wv >> 2;                         // channel group / patch part of this wave
    const int* __restrict__ tab = p.ch_off + (int64_t)ch_row * p.CinPad + 4 * sg;
    unsigned voff[NKP];
    int lidx[NKP], xcol[NKP];
    const int* __restrict__ shtab = p.ch_shift ? p.ch_shift + (int64_t)ch_row * p.CinPad + 4 * sg : nullptr;
#pragma unroll
    for (int k = 0; k < NKP; k++) {
        const int pidx = spart * (NKP * 64) + lane + 64 * k;
        const int pr = pidx / PC, pc = pidx - pr * PC;
        const int iy = ty0 * S - pad_y + pr, ix = tx0 * S - pad_x + pc;
        const bool own = pidx < Cfg::NPIX && lane + 64 * k < NKP * 64;
        xcol[k] = ix;
        voff[k] = (own && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) ? (unsigned)(iy * p.x_pitch + ix) * (XIL ? 16u : ESX) : kBufOOB;
        const int col = S == 2 ? (pc & 1) * Cfg::PCH + (pc >> 1) : pc;
        lidx[k] = own ? (pr * PCL + col) * PXB + sg * 8 : -1;
    }
    // w_off and the slab size count 16-byte slots
    const char* __restrict__ wsrc = reinterpret_cast<const char*>(p.w) + (w_off + ((int64_t)nblk * nchunks) * Cfg::W_SLOTS) * 16;
    const buf_rsrc rs_w = make_buf(wsrc);

    f32x4 rin[NKP];
    f32x4 rw[NK_W];
    auto prefetch = [&](int ch) {
        if constexpr (XIL) {
            const int off = tab[ch * CC];                          // group offset == planar offset of its first channel
            const int sh = shtab ? shtab[ch * CC] : 0;             // folded cost volume: 16-byte slots, sh pixels to the left
            const buf_rsrc rs = make_buf(xb, off >= 0);
#pragma unroll
            for (int k = 0; k < NKP; k++) {
                const unsigned vo = sh == 0 ? voff[k] : ((voff[k] != kBufOOB && xcol[k] >= sh) ? voff[k] - (unsigned)sh * 16u : kBufOOB);
                rin[k] = buf_load4(rs, vo, (unsigned)off * 4u);
            }
        } else {
            // folded cost volume: the 4 channels of the group share one shift (right-image channels of depth slice d: d)
            const int sh = shtab ? shtab[ch * CC] : 0;             // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int off = tab[ch * CC + j];                  // wave-uniform scalar load
                const buf_rsrc rs = make_buf(xb, off >= 0);
#pragma unroll
                for (int k = 0; k < NKP; k++) {
                    const unsigned vo = sh == 0 ? voff[k] : ((voff[k] != kBufOOB && xcol[k] >= sh) ? voff[k] - (unsigned)sh * ESX : kBufOOB);
                    rin[k][j] = Io<TIN>::load(rs, vo, (unsigned)off * ESX);
                }
            }
        }
        const unsigned so = (unsigned)ch * (unsigned)(Cfg::W_SLOTS * 16);
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + NT * k;
            rw[k] = buf_load4(rs_w, idx < Cfg::W_SLOTS ? (unsigned)idx * 16u : kBufOOB, so);
        }
    };
    auto stage_to_lds = [&]() {
#pragma unroll
        for (int k = 0; k < NKP; k++) {
            if (lidx[k] < 0) continue;
            if constexpr (XF16) {                                   // the stored fp16 values are the operands; no low part
                f16x4 h;
#pragma unroll
                for (int j = 0; j < 4; j++) h[j] = (_Float16)rin[k][j];
                *reinterpret_cast<f16x4*>(sIn + lidx[k]) = h;
            } else {
                const S3Split s = s3_split(rin[k]);
                *reinterpret_cast<f16x4*>(sIn + lidx[k]) = s.hi;
                *reinterpret_cast<f16x4*>(sIn + lidx[k] + 32) = s.lo;
            }
        }
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + NT * k;
            if (idx < Cfg::W_SLOTS) sW[idx] = rw[k];
        }
    };

    // ---- output addressing; the residual tile is requested first and consumed in the epilogue ----------------------------
    const int64_t ybase = (int64_t)n * p.y_bstride + y_off;
    const int64_t rbase = (int64_t)n * p.r_bstride + r_off;
    const int cs32 = (int)p.y_cstride, rs32 = (int)p.r_cstride;
    const int oy = ty0 + wv, ox = tx0 + l31;
    const bool inb = oy < Ho && ox < Wo;
    const bool r_il = p.r_il8 != 0;
    const int cb = nblk * 32;                                       // first channel of this block
    f32x4 rr[4];
    if (p.resid != nullptr && kgi == 0) {
        // interleaved tensors (2-D plans only): pixel offsets -- r_off / y_off of a transposed-convolution phase -- count
        // 16-byte slots, i.e. 4 elements
        const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, r_il ? (int64_t)n * p.r_bstride + (YF16 ? r_off_il8 : r_off_il4) : rbase, ESY));
        if (r_il && YF16) {
            // fp16 residual, (C/8, H, W, 8): the lane's 4 consecutive channels of a pixel are 8 bytes (the skip tensor of a fused
            // Conv3DTranspose + skip + ELU launch in half2 mode, written by conv_f16mma_kernel)
            const unsigned vo = inb ? (unsigned)((oy * p.y_ystride + ox * p.y_xstride) * 8 + 4 * kg) * 2u : kBufOOB;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(rs_r, (cb + 8 * q + 4 * kg < p.Cout) ? vo : kBufOOB, (unsigned)((cb + 8 * q) * rs32) * 2u, 0);
#pragma unroll
                for (int e = 0; e < 4; e++) rr[q][e] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u[e >> 1] >> (16 * (e & 1))));
            }
        } else if (r_il) {
            const unsigned vo = inb ? (unsigned)((oy * p.y_ystride + ox * p.y_xstride) * 4 + 4 * kg * rs32) * 4u : kBufOOB;
#pragma unroll
            for (int q = 0; q < 4; q++) rr[q] = buf_load4(rs_r, (cb + 8 * q + 4 * kg < p.Cout) ? vo : kBufOOB, (unsigned)((cb + 8 * q) * rs32) * 4u);
        } else {
            const unsigned vo = inb ? (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * kg * rs32) * ESY : kBufOOB;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    rr[q][e] = Io<TOUT>::load(rs_r, (cb + 8 * q + 4 * kg + e < p.Cout) ? vo : kBufOOB, (unsigned)((cb + 8 * q + e) * rs32) * ESY);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) rr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    f32x16 acc_m, acc_c;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc_m[r] = 0.f; acc_c[r] = 0.f; }
    const int a_off = kg * 32 + l31;
    const bool w_exact = p.w_exact != 0;
    // B operand of tap (r, s): pixel (wv*S + r, l31*S + s); stride 2: column parity (s & 1), index l31 + (s >> 1)
    const int b_off = (wv * S * PCL + l31) * PXB + kg * 16;
    auto compute = [&]() {
#pragma unroll
        for (int t = 0; t < TAPS; t++) {
            if (!((tap_mask >> t) & 1u)) continue;                  // wave-uniform
            const int r = t / KW, s = t % KW;
            const int col = S == 2 ? (s & 1) * Cfg::PCH + (s >> 1) : s;
            const char* bp = sIn + b_off + (r * PCL + col) * PXB;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(bp);
            const f16x8 ah = __builtin_bit_cast(f16x8, sW[a_off + (t * 2 + 0) * 64]);
            acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc_m, 0, 0, 0);
            if (!w_exact) {                                         // wave-uniform: fp16 weights have no low part
                const f16x8 al = __builtin_bit_cast(f16x8, sW[a_off + (t * 2 + 1) * 64]);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc_c, 0, 0, 0);
            }
            if constexpr (!XF16) {
                const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + 32);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc_c, 0, 0, 0);
            }
        }
    };

    if (kgi < nchunks) prefetch(kgi);
    RT_TSTAMP();                                  // 1: residual + first gathers issued
    for (int ch = kgi; ch < nchunks + kgi; ch += KS) {      // every group runs the same number of rounds (barriers are workgroup-wide)
        const bool mine = KS == 1 || ch < nchunks;          // wave-uniform
        if (ch != kgi) __syncthreads();       // everyone finished reading the previous chunk from LDS
        if (mine) stage_to_lds();
        __syncthreads();
        RT_TSTAMP();                              // 2, 4: chunk in LDS
        if (ch + KS < nchunks) prefetch(ch + KS);
        if (mine) compute();
        RT_TSTAMP();                              // 3, 5: MFMAs issued
    }

    // main + cross terms; split-K: the sums of groups 1 .. KS-1 travel through LDS (the operand images are dead) and are added in group order
#pragma unroll
    for (int r = 0; r < 16; r++) acc_m[r] = fmaf(acc_c[r], kSplitInv, acc_m[r]);
    if (KS > 1) {
        f32x4* const red = reinterpret_cast<f32x4*>(smem);
        __syncthreads();
        if (kgi > 0) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                red[(((kgi - 1) * NW + wv) * 4 + q) * 64 + lane] = f32x4{acc_m[4 * q], acc_m[4 * q + 1], acc_m[4 * q + 2], acc_m[4 * q + 3]};
        }
        __syncthreads();
        if (kgi > 0) return;
        for (int g = 1; g < KS; g++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = red[(((g - 1) * NW + wv) * 4 + q) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; e++) acc_m[4 * q + e] += v[e];
            }
    }

    // ---- epilogue -------------------------------------------------------------------------------------------------------
    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (YIL && !YF16) ? (int64_t)n * p.y_bstride + y_off_il4 : ybase, ESY));
    auto epilogue = [&](auto ACT) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + cb + 8 * q + 4 * kg);     // padded to 64 channels
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = apply_act_fast(acc_m[4 * q + e] + (bv[e] + rr[q][e]), decltype(ACT)::value);
            if constexpr (YIL && YF16) {
                // (K/8, H, W, 8) fp16: the lane's 4 consecutive channels of the pixel as one 8-byte store (y_off = depth-slice offset in elements)
                const unsigned vo = (inb && cb + 8 * q + 4 * kg < p.Cout) ? (unsigned)((oy * p.y_ystride + ox * p.y_xstride) * 8 + 4 * kg) * 2u : kBufOOB;
                u32x2_t w2;
#pragma unroll
                for (int e = 0; e < 2; e++)
                    w2[e] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)o[2 * e]) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)o[2 * e + 1]) << 16);
                __builtin_amdgcn_raw_buffer_store_b64(w2, rs_y, vo, (unsigned)((cb + 8 * q) * cs32) * 2u, 0);
            } else if constexpr (YIL) {
                const unsigned vo = (inb && cb + 8 * q + 4 * kg < p.Cout) ? (unsigned)((oy * p.y_ystride + ox * p.y_xstride) * 4 + 4 * kg * cs32) * 4u : kBufOOB;
                buf_store4(o, rs_y, vo, (unsigned)((cb + 8 * q) * cs32) * 4u);
            } else {
                const unsigned vo = inb ? (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * kg * cs32) * ESY : kBufOOB;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    Io<TOUT>::store(o[e], rs_y, (cb + 8 * q + 4 * kg + e < p.Cout) ? vo : kBufOOB, (unsigned)((cb + 8 * q + e) * cs32) * ESY);
            }
        }
    };
    if (act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (act == 2) epilogue(std::integral_constant<int, 2>{});
    else epilogue(std::integral_constant<int, 0>{});
    RT_TSTAMP();                                  // 2 + 2 * nchunks: stores issued
#ifdef RT_KERNEL_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    if (dbgp && tid == 0) { dbgp[14] = __builtin_amdgcn_s_memtime(); dbgp[15] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
}

}  // namespace rt

namespace rt {

// -----------------------------------------------------------------------------------------------------------------------
// conv_s3_first_kernel: the networks' first layer -- 5x5, stride 2, on the 3-channel fp32 image (reference
// resnet18_2D_513x257_net.cpp:48-63, nvsmall_1025x321_net.cpp:36-51) -- in the row-as-contraction form of
// conv_f16_first.hip.h on the 3-term split: with 3 input channels a channel-chunked contraction is 81 % padding, so a ROW
// of the window is the contraction index,  k = 3*s + c  (s = window column, c = channel; 15 values, the 16th has zero
// weights), contiguous in LDS patches stored [row][col][channel] (one of fp16 high parts, one of scaled low parts).
// 5 window rows x 3 split terms = 15 MFMAs per 32 pixels x 32 output channels (the fp32 direct form: 50 fp32 MFMAs = 25x
// the matrix-pipe time); the 10 A operands (hi / lo weights [r][k-half][co][8]) stay in registers, one barrier.
// Bound by data movement: 11 x 67 x 3 fp32 in, 4 x 32 x 32 fp32 out per tile.
// -----------------------------------------------------------------------------------------------------------------------
struct S3FirstCfg {
    static constexpr int KH = 5, KW = 5, S = 2, TY = 4, TX = 32, CMAX = 3;
    static constexpr int PR = (TY - 1) * S + KH, PC = (TX - 1) * S + KW;      // 11 x 67 input pixels per channel
    static constexpr int NPAIR = (PC + 1) / 2;                                // column pairs per row: 34 (68 columns)
    static constexpr int RS = 208;                                            // LDS row stride in halfs >= 68*3 = 204, 16-byte multiple
    static constexpr int NTASK = CMAX * PR * NPAIR, NK = (NTASK + 255) / 256; // (channel, row, pair) gathers per lane
    static constexpr int W_SLOTS = KH * 2 * 2 * 32;                           // 16-byte slots per 32-channel block: [r][hi/lo][k-half][co]
};

template <bool YIL>
__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(4) conv_s3_first_kernel(ConvArgs p) {
    using Cfg = S3FirstCfg;
    constexpr int KH = Cfg::KH, S = Cfg::S, TY = Cfg::TY, TX = Cfg::TX, PR = Cfg::PR, NPAIR = Cfg::NPAIR, RS = Cfg::RS;

    __shared__ __attribute__((aligned(16))) _Float16 sHi[PR * RS];
    __shared__ __attribute__((aligned(16))) _Float16 sLo[PR * RS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    int tile = blockIdx.x;
    if (p.xcd_order) {                            // contiguous tile range per XCD (see conv_mfma.hip.h)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx0 = (tile % p.tiles_x) * TX;
    const int ty0 = (tile / p.tiles_x) * TY;
    const int nblk = blockIdx.y;
    const int n = blockIdx.z;

    // ---- weights: the 10 A operands of this lane (co = l31, k-half = kg), straight from global memory --------------------
    f32x4 wa[KH][2];
    {
        const buf_rsrc rs_w = make_buf(reinterpret_cast<const char*>(p.w) + (int64_t)nblk * Cfg::W_SLOTS * 16);
#pragma unroll
        for (int r = 0; r < KH; r++)
#pragma unroll
            for (int part = 0; part < 2; part++) wa[r][part] = buf_load4(rs_w, (unsigned)(((r * 2 + part) * 2 + kg) * 32 + l31) * 16u, 0);
    }

    // ---- gather: fp32 image -> split patches [row][col][channel]; a lane takes a column pair of one channel --------------
    const buf_rsrc rs_x = make_buf((p.x2 && n >= p.x2_from) ? p.x2 + (int64_t)(n - p.x2_from) * p.x_bstride : p.x + (int64_t)n * p.x_bstride);
    const int ix0 = tx0 * S - p.pad_x, iy0 = ty0 * S - p.pad_y;
    const int cin = p.cin_real;
#pragma unroll
    for (int k = 0; k < Cfg::NK; k++) {
        const int t = tid + 256 * k;
        const int c = t / (PR * NPAIR), rem = t - c * (PR * NPAIR);
        const int pr = rem / NPAIR, pc = 2 * (rem - pr * NPAIR);
        const int iy = iy0 + pr, ix = ix0 + pc;
        const bool own = t < Cfg::NTASK && c < cin;
        const bool row_ok = own && iy >= 0 && iy < p.Hi;
        // the pair may straddle either edge of a dense row: each element is masked on its own
        const bool ok0 = row_ok && ix >= 0 && ix < p.Wi, ok1 = row_ok && ix + 1 >= 0 && ix + 1 < p.Wi && pc + 1 < Cfg::PC;
        const unsigned vo = (unsigned)((c * p.Hi + iy) * p.x_pitch + ix) * 4u;
        const float v0 = buf_load(rs_x, ok0 ? vo : kBufOOB, 0);
        const float v1 = buf_load(rs_x, ok1 ? vo + 4u : kBufOOB, 0);
        if (t < Cfg::NTASK) {                     // channels >= cin are written as zeros
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const int o = pr * RS + pc * Cfg::CMAX + c;
            sHi[o] = h0;
            sHi[o + Cfg::CMAX] = h1;
            sLo[o] = (_Float16)((v0 - (float)h0) * kSplitScale);
            sLo[o + Cfg::CMAX] = (_Float16)((v1 - (float)h1) * kSplitScale);
        }
    }
    __syncthreads();

    // ---- 5 window rows x 3 MFMAs: B operand = 8 consecutive (column, channel) halfs of patch row 2*wv + r ---------------------
    f32x16 acc_m, acc_c;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc_m[r] = 0.f; acc_c[r] = 0.f; }
    const unsigned* __restrict__ hi32 = reinterpret_cast<const unsigned*>(sHi);      // 6*x + 8*h halfs is an even offset
    const unsigned* __restrict__ lo32 = reinterpret_cast<const unsigned*>(sLo);
#pragma unroll
    for (int r = 0; r < KH; r++) {
        const int base = ((wv * S + r) * RS + 6 * l31 + 8 * kg) >> 1;
        u32x4_t bh, bl;
#pragma unroll
        for (int e = 0; e < 4; e++) { bh[e] = hi32[base + e]; bl[e] = lo32[base + e]; }
        const f16x8 ah = __builtin_bit_cast(f16x8, wa[r][0]), al = __builtin_bit_cast(f16x8, wa[r][1]);
        acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, bh), acc_m, 0, 0, 0);
        acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(f16x8, bh), acc_c, 0, 0, 0);
        acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, bl), acc_c, 0, 0, 0);
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------------
    const int oy = ty0 + wv, ox = tx0 + l31;
    const bool inb = oy < p.Ho && ox < p.Wo;
    const int cs32 = (int)p.y_cstride;
    const int cb = nblk * 32;
    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride + p.y_off, 4));
    const int act = p.act;
    auto epilogue = [&](auto ACT) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + cb + 8 * q + 4 * kg);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = apply_act_fast(fmaf(acc_c[4 * q + e], kSplitInv, acc_m[4 * q + e]) + bv[e], decltype(ACT)::value);
            if constexpr (YIL) {
                const unsigned vo = (inb && cb + 8 * q + 4 * kg < p.Cout) ? (unsigned)((oy * p.y_ystride + ox) * 4 + 4 * kg * cs32) * 4u : kBufOOB;
                buf_store4(o, rs_y, vo, (unsigned)((cb + 8 * q) * cs32) * 4u);
            } else {
                const unsigned vo = inb ? (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * kg * cs32) * 4u : kBufOOB;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    buf_store(o[e], rs_y, (cb + 8 * q + 4 * kg + e < p.Cout) ? vo : kBufOOB, (unsigned)((cb + 8 * q + e) * cs32) * 4u);
            }
        }
    };
    if (act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (act == 2) epilogue(std::integral_constant<int, 2>{});
    else epilogue(std::integral_constant<int, 0>{});
}

}  // namespace rt

namespace rt {

// -----------------------------------------------------------------------------------------------------------------------
// conv_s3rb_kernel: a whole residual block in one launch,
//       y = act2( conv3x3( act1( conv3x3(x) + b1 ) ) + b2 + x )          (Cin = Cmid = Cout <= 32, stride 1)
// the unit the feature towers of ResNet-18 2D / ResNet-18 3D are made of (reference resnet18_2D_513x257_net.cpp:66-575:
// resblockN_conv1 -> ELU -> resblockN_conv2 -> add -> ELU, 8 blocks per side = 32 of the network's 49 convolutions).
// Layer by layer the block moves x, t, t, x, y through HBM (5 tensor passes); fused, the intermediate t lives in LDS and
// the block moves x (+ halo) and y.  Workgroup = 4 waves = a 4-row x 32-pixel output tile, ~76 KB of LDS, two workgroups
// per CU so that one's gathers / stores overlap the other's MFMAs:
//   1. conv1 on the 6 x 34 region conv2 needs (204 pixels = 7 MFMA column blocks of 32: two per wave), contraction in two
//      chunks of 16 input channels staged like conv_s3_kernel does (x region 8 x 36, split into fp16 hi / lo, 80 B per pixel;
//      split weights of the chunk; the next chunk's gathers fly under the MFMAs);
//      epilogue = bias, activation, ZERO outside the image (conv2's padding), split, LDS image sT[chunk][pixel]
//   2. conv2 on the 4 x 32 tile out of sT (one row per wave), its two weight chunks staged into the same buffer; the
//      residual (the fp32 x values, L2-hot) is requested before the MFMAs and added in the epilogue.
// The recomputed halo costs 7/4 of conv1's multiplies (11/8 of the block's); the matrix pipe has that slack, HBM does not.
// -----------------------------------------------------------------------------------------------------------------------
struct S3RBCfg {
    static constexpr int NW = 4, TY = 4, TX = 32, NT = 256;
    static constexpr int XR = TY + 4, XC = TX + 4, XPIX = XR * XC;          // input region 8 x 36
    static constexpr int TR = TY + 2, TC = TX + 2, TPIX = TR * TC;          // conv1 output region 6 x 34
    static constexpr int TSEG = (TPIX + 31) / 32;                            // 7 column blocks of 32 pixels
    static constexpr int SEGW = (TSEG + NW - 1) / NW;                        // 2 per wave
    static constexpr int PXB = 80;                                           // 16 channels: 32 B hi | 32 B lo | 16 B pad
    static constexpr int NSLOT = XPIX * 4, NKX = (NSLOT + NT - 1) / NT;      // (pixel, 4-channel group) gather slots per thread and chunk
    static constexpr int W_SLOTS = 9 * 2 * 2 * 32, NK_W = (W_SLOTS + NT - 1) / NT;    // one 16-channel chunk of split weights
};

struct RBArgs {
    ConvArgs c;            // geometry, x / y pointers and strides, bias = conv2's, w = conv2's split weights, act = act2
    const float* w1;       // conv1's split weights (conv_s3_kernel's slab order: [chunk][tap][hi/lo][k-group][co][8])
    const float* bias1;    // padded to 64
    int act1;
    int cmid;              // channels of the intermediate (conv1's outputs = conv2's inputs)
    int seg;               // conv_s3rbs_kernel: output rows per workgroup (a multiple of 4)
};

template <bool XIL, bool YIL>
__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(2) conv_s3rb_kernel(RBArgs a) {
    using Cfg = S3RBCfg;
    const ConvArgs& p = a.c;
    constexpr int TY = Cfg::TY, TX = Cfg::TX, XC = Cfg::XC, TC = Cfg::TC, PXB = Cfg::PXB, NT = Cfg::NT;
    constexpr int NKX = Cfg::NKX, NK_W = Cfg::NK_W, SEGW = Cfg::SEGW;

    __shared__ __attribute__((aligned(16))) f32x4 sW[Cfg::W_SLOTS];
    __shared__ __attribute__((aligned(16))) char sX[Cfg::XPIX * PXB];
    __shared__ __attribute__((aligned(16))) char sT[2][Cfg::TSEG * 32 * PXB];

    const int tid = threadIdx.x;
#ifdef RT_KERNEL_TIMING
    unsigned long long* dbgp = p.dbg ? p.dbg + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16 : nullptr;
    int dbi = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    RT_TSTAMP();                                  // 0: start
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    int tile = blockIdx.x;
    if (p.xcd_order) {                            // contiguous tile range per XCD (see conv_mfma.hip.h)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx0 = (tile % p.tiles_x) * TX;
    const int ty0 = (tile / p.tiles_x) * TY;
    const int n = blockIdx.z;
    const int H = p.Hi, W = p.Wi;                  // stride 1, same padding: input, intermediate and output grids coincide

    // ---- gather slots of this thread (the same for both chunks): patch pixel and channel group within the chunk ----------------
    const buf_rsrc rs_x = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, 4));
    const unsigned cs_x = (unsigned)p.x_cstride;
    unsigned xoff[NKX];            // in-plane byte offset of the pixel, kBufOOB outside the image
    int xg[NKX], xlds[NKX];
#pragma unroll
    for (int k = 0; k < NKX; k++) {
        const int idx = tid + NT * k;
        const int g = idx / Cfg::XPIX, pix = idx - g * Cfg::XPIX;
        const int pr = pix / XC, pc = pix - pr * XC;
        const int iy = ty0 - 2 + pr, ix = tx0 - 2 + pc;
        const bool in = idx < Cfg::NSLOT && iy >= 0 && iy < H && ix >= 0 && ix < W;
        xoff[k] = in ? (unsigned)(iy * p.x_pitch + ix) * (XIL ? 16u : 4u) : kBufOOB;
        xg[k] = g;
        xlds[k] = idx < Cfg::NSLOT ? pix * PXB + g * 8 : -1;
    }
    f32x4 rin[NKX];
    f32x4 rw[NK_W];
    auto gather_x = [&](int ch) {
#pragma unroll
        for (int k = 0; k < NKX; k++) {
            const int c0 = 16 * ch + 4 * xg[k];                          // first channel of the group
            const bool ok = xoff[k] != kBufOOB && c0 < p.cin_real;
            if constexpr (XIL) {
                rin[k] = buf_load4(rs_x, ok ? (unsigned)c0 * cs_x * 4u + xoff[k] : kBufOOB, 0u);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    rin[k][j] = buf_load(rs_x, (ok && c0 + j < p.cin_real) ? (unsigned)(c0 + j) * cs_x * 4u + xoff[k] : kBufOOB, 0u);
            }
        }
    };
    auto gather_w = [&](const float* w, int ch) {
        const buf_rsrc rs_w = make_buf(w);
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + NT * k;
            rw[k] = buf_load4(rs_w, idx < Cfg::W_SLOTS ? (unsigned)idx * 16u : kBufOOB, (unsigned)ch * (unsigned)(Cfg::W_SLOTS * 16));
        }
    };
    auto stage_x = [&]() {
#pragma unroll
        for (int k = 0; k < NKX; k++) {
            if (xlds[k] < 0) continue;
            const S3Split s = s3_split(rin[k]);
            *reinterpret_cast<f16x4*>(sX + xlds[k]) = s.hi;
            *reinterpret_cast<f16x4*>(sX + xlds[k] + 32) = s.lo;
        }
    };
    auto stage_w = [&]() {
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + NT * k;
            if (idx < Cfg::W_SLOTS) sW[idx] = rw[k];
        }
    };
    const int a_off = kg * 32 + l31;
    // 9 taps x 3 MFMAs of one 16-channel chunk: B operand of tap (r, s) at bp0 + (r * row_pitch + s) pixels
    auto contract = [&](const char* bp0, int row_pitch, f32x16& acc_m, f32x16& acc_c) {
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int r = t / 3, s = t % 3;
            const char* bp = bp0 + (r * row_pitch + s) * PXB;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(bp);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + 32);
            const f16x8 ah = __builtin_bit_cast(f16x8, sW[a_off + (t * 2 + 0) * 64]);
            const f16x8 al = __builtin_bit_cast(f16x8, sW[a_off + (t * 2 + 1) * 64]);
            acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc_m, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc_c, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc_c, 0, 0, 0);
        }
    };

    // ---- conv1: column blocks wv and wv + 4 of the 6 x 34 intermediate region -----------------------------------------------------
    int pt[SEGW], bx[SEGW];
#pragma unroll
    for (int i = 0; i < SEGW; i++) {
        pt[i] = (wv + Cfg::NW * i) * 32 + l31;                     // intermediate pixel (row-major, 34 wide); >= TPIX: padding lanes
        const int tr = pt[i] / TC, tc = pt[i] - tr * TC;
        // taps (r, s) of intermediate pixel (tr, tc) read input-region pixels (tr + r, tc + s); padding lanes read pixel 0
        bx[i] = (pt[i] < Cfg::TPIX ? (tr * XC + tc) * PXB : 0) + kg * 16;
    }
    f32x16 am[SEGW], ac[SEGW];
#pragma unroll
    for (int i = 0; i < SEGW; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) { am[i][r] = 0.f; ac[i][r] = 0.f; }
    const bool seg1 = wv + Cfg::NW < Cfg::TSEG;                     // wave-uniform: this wave has a second column block

    gather_x(0);
    gather_w(a.w1, 0);
    RT_TSTAMP();                                  // 1: first gathers issued
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        if (ch) __syncthreads();                  // everyone finished reading the previous chunk
        stage_x();
        stage_w();
        __syncthreads();
        RT_TSTAMP();                              // 2, 4: chunk in LDS
        if (ch == 0) { gather_x(1); gather_w(a.w1, 1); }
        else gather_w(p.w, 0);                    // conv2's first weight chunk flies under conv1's last MFMAs
        contract(sX + bx[0], XC, am[0], ac[0]);
        if (seg1) contract(sX + bx[1], XC, am[1], ac[1]);
        RT_TSTAMP();                              // 3, 5: MFMAs issued
    }
    // conv1 epilogue -> sT[chunk of the intermediate channel][pixel]
#pragma unroll
    for (int i = 0; i < SEGW; i++) {
        if (i == 1 && !seg1) break;
        if (pt[i] < Cfg::TPIX) {
            const int tr = pt[i] / TC, tc = pt[i] - tr * TC;
            const int gy = ty0 - 1 + tr, gx = tx0 - 1 + tc;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias1 + 8 * q + 4 * kg);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = apply_act_fast(fmaf(ac[i][4 * q + e], kSplitInv, am[i][4 * q + e]) + bv[e], a.act1);
                    o[e] = (inside && 8 * q + 4 * kg + e < a.cmid) ? v : 0.f;       // conv2 pads the intermediate with zeros
                }
                const S3Split s = s3_split(o);
                char* dst = sT[q >> 1] + pt[i] * PXB + (8 * (q & 1) + 4 * kg) * 2;
                *reinterpret_cast<f16x4*>(dst) = s.hi;
                *reinterpret_cast<f16x4*>(dst + 32) = s.lo;
            }
        }
    }
    RT_TSTAMP();                                  // 6: intermediate written

    // ---- conv2 on the 4 x 32 tile (row wv); residual requested first -----------------------------------------------------------------
    const int oy = ty0 + wv, ox = tx0 + l31;
    const bool inb = oy < p.Ho && ox < p.Wo;
    const int cs32 = (int)p.y_cstride, rs32 = (int)p.r_cstride;
    f32x4 rr[4];
    {
        const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, (int64_t)n * p.r_bstride, 4), p.resid != nullptr);
        if (p.r_il8) {
            const unsigned vo = inb ? (unsigned)((oy * p.x_pitch + ox) * 4 + 4 * kg * rs32) * 4u : kBufOOB;
#pragma unroll
            for (int q = 0; q < 4; q++) rr[q] = buf_load4(rs_r, (8 * q + 4 * kg < p.Cout) ? vo : kBufOOB, (unsigned)(8 * q * rs32) * 4u);
        } else {
            const unsigned vo = inb ? (unsigned)(oy * p.x_pitch + ox + 4 * kg * rs32) * 4u : kBufOOB;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    rr[q][e] = buf_load(rs_r, (8 * q + 4 * kg + e < p.Cout) ? vo : kBufOOB, (unsigned)((8 * q + e) * rs32) * 4u);
        }
    }
    f32x16 acc_m, acc_c;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc_m[r] = 0.f; acc_c[r] = 0.f; }
    const int b2 = (wv * TC + l31) * PXB + kg * 16;
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        __syncthreads();                          // conv1's (or the previous chunk's) weights are no longer read; sT complete
        stage_w();
        __syncthreads();
        if (ch == 0) gather_w(p.w, 1);
        contract(sT[ch] + b2, TC, acc_m, acc_c);
        RT_TSTAMP();                              // 7, 8: conv2 MFMAs issued
    }
    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride + p.y_off, 4));
    const int act = p.act;
    auto epilogue = [&](auto ACT) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + 8 * q + 4 * kg);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[e] = apply_act_fast(fmaf(acc_c[4 * q + e], kSplitInv, acc_m[4 * q + e]) + (bv[e] + rr[q][e]), decltype(ACT)::value);
            if constexpr (YIL) {
                const unsigned vo = (inb && 8 * q + 4 * kg < p.Cout) ? (unsigned)((oy * p.y_ystride + ox) * 4 + 4 * kg * cs32) * 4u : kBufOOB;
                buf_store4(o, rs_y, vo, (unsigned)(8 * q * cs32) * 4u);
            } else {
                const unsigned vo = inb ? (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * kg * cs32) * 4u : kBufOOB;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    buf_store(o[e], rs_y, (8 * q + 4 * kg + e < p.Cout) ? vo : kBufOOB, (unsigned)((8 * q + e) * cs32) * 4u);
            }
        }
    };
    if (act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (act == 2) epilogue(std::integral_constant<int, 2>{});
    else epilogue(std::integral_constant<int, 0>{});
    RT_TSTAMP();                                  // 9: stores issued
#ifdef RT_KERNEL_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    if (dbgp && tid == 0) { dbgp[14] = __builtin_amdgcn_s_memtime(); dbgp[15] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
}

}  // namespace rt
