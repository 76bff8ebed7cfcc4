// Stride-2 transposed 3x3(x3) convolution on fp32 tensors (3-term fp16 split, fp32 accumulation), ALL FOUR output phases per workgroup
// (round 4) -- the fp32 sibling of deconv_f16p.hip.h, for the 3-D decoders of fp32 engines (BASELINE C4; reference
// lib/conv3d_transpose_plugin.cpp:205-243).
//
// The ZSlice form (conv_s3_kernel<2,2,1> with one slice per output phase) launches a workgroup per (tile, depth, PHASE): every phase
// gathers the same input patch again and writes its outputs as single floats 8 bytes apart.  PMC, NVSmall fp32 deconv3D_2: 2.8 GB
// fetched and 1.08 GB written in 0.675 ms for 0.8 GB of operands, 13 % MFMA-busy.  Here a workgroup owns a 4 x 32 tile of the INPUT
// grid = an 8 x 64 tile of the output:
//   * the patch (5 x 33 pixels, 16 gathered channels per chunk) is gathered and split into fp16 high / low parts ONCE and serves the 9
//     taps of the full 3 x 3 window -- tap (ry, rx) contributes to exactly one phase (per dimension r = 1: even outputs from input m,
//     r = 2: odd outputs from m, r = 0: odd outputs from m + 1): 27 MFMAs per wave and chunk into 4 + 4 accumulators (main and
//     cross terms of each phase);
//   * a lane holds BOTH x-phases of its input column, i.e. two neighbouring output pixels: one 8-byte store per channel and output
//     row, 256 contiguous bytes per half-wave instead of 4-byte pieces every 8 bytes;
//   * weights in KERNEL order, split slabs [nblk][chunk][tap 9][hi / lo][k-group][co][8] as conv_s3_kernel reads them (the plan packs
//     them with the same routine); chunks double-buffered in LDS (2 x 31 KB, two workgroups per CU), one barrier per chunk.
// Tensors: input planar (K, Dy, Hy, Wy) through the plan's gather table (depth taps merged into the channel axis, -1 = zeros), output
// planar (D, C, H, W) or with the fused Transform (C, D, H, W), skip tensor planar or (D, C/4, H, W, 4) (ConvArgs::r_il8).  Same
// arithmetic as the ZSlice form, operation for operation (chunks outermost, a phase's taps by ascending input offset, bias and skip
// tensor added after the main + cross sum): bit-identical results (tests/test_deconv3d_half2.py).
#pragma once
#include "common.hip.h"
#include "conv_mfma.hip.h"
#include "conv_split.hip.h"

namespace rt {

struct DeconvS3PCfg {
    static constexpr int TY = 4, TX = 32, CC = 16;
    static constexpr int PR = TY + 1, PC = TX + 1, NPIX = PR * PC;     // patch rows / columns (+1 halo below / right)
    static constexpr int NKP = (NPIX + 63) / 64;                       // patch pixels per lane (each wave gathers one group of 4 channels)
    static constexpr int PXB = 80;                                     // bytes per patch pixel: 32 hi + 32 lo + 16 pad
    static constexpr int IN_BYTES = (NPIX * PXB + 15) / 16 * 16;
    static constexpr int W_SLOTS = 9 * 2 * 2 * 32;
    static constexpr int NK_W = (W_SLOTS + 255) / 256;
    static constexpr int BUF_BYTES = IN_BYTES + W_SLOTS * 16;
    static_assert(2 * BUF_BYTES <= 65536, "static LDS");
};

// p.zs: one ZSlice per output depth of the launch's class (offsets of phase (0, 0), gather-table row); p.Ho / p.Wo: the FULL output
// plane (Hx, Wx) = (2 Hi - 1, 2 Wi - 1); p.y_ystride = 2 * Wx.
// YIL: the output is channel-interleaved in groups of 4 -- (D, C/4, H, W, 4) or, with the fused Transform, (C/4, D, H, W, 4), the tensor
// the last layer's matrix-core kernel reads (deconv3d_small.hip.h: deconv3d_s2_il4_kernel); ZSlice::y_off_il8 carries its depth offset.
template <bool YIL>
__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(2) deconv_s3p_kernel(ConvArgs p) {
    using Cfg = DeconvS3PCfg;
    constexpr int NKP = Cfg::NKP, NK_W = Cfg::NK_W, PXB = Cfg::PXB, PC = Cfg::PC;
    __shared__ __attribute__((aligned(16))) char smem[2 * Cfg::BUF_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    RT_WG_TILE(p, tile, zi, n)
    const int tx0 = (tile % p.tiles_x) * Cfg::TX;
    const int ty0 = (tile / p.tiles_x) * Cfg::TY;
    const int nblk = blockIdx.y;
    const char* __restrict__ xb = elem_ptr(p.x, (int64_t)n * p.x_bstride, 4);
    const int nchunks = p.CinPad / Cfg::CC;
    const ZSlice z = p.zs[zi];

    // ---- staging: wave w gathers channels 4w .. 4w + 3 of each chunk for every patch pixel ------------------------------------------
    const int* __restrict__ tab = p.ch_off + (int64_t)z.ch_row * p.CinPad + 4 * wv;
    unsigned voff[NKP];
    int lidx[NKP];
#pragma unroll
    for (int k = 0; k < NKP; k++) {
        const int pidx = lane + 64 * k;
        const int pr = pidx / PC, pc = pidx - pr * PC;
        const int iy = ty0 + pr, ix = tx0 + pc;
        const bool own = pidx < Cfg::NPIX;
        voff[k] = (own && iy < p.Hi && ix < p.Wi) ? (unsigned)(iy * p.x_pitch + ix) * 4u : kBufOOB;
        lidx[k] = own ? pidx * PXB + wv * 8 : -1;
    }
    const char* __restrict__ wsrc = reinterpret_cast<const char*>(p.w) + ((int64_t)nblk * nchunks) * Cfg::W_SLOTS * 16;
    const buf_rsrc rs_w = make_buf(wsrc);
    f32x4 rin[NKP], rw[NK_W];
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int off = tab[ch * Cfg::CC + j];                     // wave-uniform; -1 = zeros (depth tap out of range, channel padding)
            const buf_rsrc rs = make_buf(xb, off >= 0);
#pragma unroll
            for (int k = 0; k < NKP; k++) rin[k][j] = buf_load(rs, voff[k], (unsigned)off * 4u);
        }
        const unsigned so = (unsigned)ch * (unsigned)(Cfg::W_SLOTS * 16);
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + 256 * k;
            rw[k] = buf_load4(rs_w, idx < Cfg::W_SLOTS ? (unsigned)idx * 16u : kBufOOB, so);
        }
    };
    auto stage = [&](int buf) {
        char* sIn = smem + buf * Cfg::BUF_BYTES;
        f32x4* sW = reinterpret_cast<f32x4*>(sIn + Cfg::IN_BYTES);
#pragma unroll
        for (int k = 0; k < NKP; k++) {
            if (lidx[k] < 0) continue;
            const S3Split s = s3_split(rin[k]);
            *reinterpret_cast<f16x4*>(sIn + lidx[k]) = s.hi;
            *reinterpret_cast<f16x4*>(sIn + lidx[k] + 32) = s.lo;
        }
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + 256 * k;
            if (idx < Cfg::W_SLOTS) sW[idx] = rw[k];
        }
    };

    f32x16 acc_m[4], acc_c[4];                     // phase 2 * py + px: main terms, cross terms (scaled by 2^11)
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc_m[f][r] = 0.f; acc_c[f][r] = 0.f; }
    const int a_base = kg * 32 + l31;
    const int b_base = (wv * PC + l31) * PXB + kg * 16;
    auto compute = [&](int buf) {
        const char* sIn = smem + buf * Cfg::BUF_BYTES;
        const f32x4* sW = reinterpret_cast<const f32x4*>(sIn + Cfg::IN_BYTES);
        f16x8 bh[2][2], bl[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                const char* bp = sIn + b_base + (dy * PC + dx) * PXB;
                bh[dy][dx] = *reinterpret_cast<const f16x8*>(bp);
                bl[dy][dx] = *reinterpret_cast<const f16x8*>(bp + 32);
            }
        // taps in the order 1, 2, 0 per dimension: every phase then sees its taps by ascending input offset (m, then m + 1), which is
        // the order of the phase windows of the ZSlice form -- same accumulation order, same bits
#pragma unroll
        for (int iy = 0; iy < 3; iy++)
#pragma unroll
            for (int ix = 0; ix < 3; ix++) {
                const int ry = iy == 0 ? 1 : (iy == 1 ? 2 : 0), rx = ix == 0 ? 1 : (ix == 1 ? 2 : 0);
                // kernel tap r: 1 -> even output from input m; 2 -> odd output from m; 0 -> odd output from m + 1
                const int py = ry == 1 ? 0 : 1, dy = ry == 0 ? 1 : 0, px = rx == 1 ? 0 : 1, dx = rx == 0 ? 1 : 0;
                const int t = ry * 3 + rx, f = 2 * py + px;
                const f16x8 ah = __builtin_bit_cast(f16x8, sW[a_base + (t * 2 + 0) * 64]);
                const f16x8 al = __builtin_bit_cast(f16x8, sW[a_base + (t * 2 + 1) * 64]);
                acc_m[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[dy][dx], acc_m[f], 0, 0, 0);
                acc_c[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[dy][dx], acc_c[f], 0, 0, 0);
                acc_c[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[dy][dx], acc_c[f], 0, 0, 0);
            }
    };

    prefetch(0);
    stage(0);
    wg_barrier();
    for (int ch = 0; ch < nchunks; ch++) {
        const bool more = ch + 1 < nchunks;
        if (more) prefetch(ch + 1);
        compute(ch & 1);
        if (more) stage((ch + 1) & 1);
        wg_barrier();
    }

    // ---- epilogue: the lane's 16 channels (4 (2 q + kg) .. + 3, q = 0 .. 3) of output pixels (2 my + py, 2 mx) and (2 my + py, 2 mx + 1) ----
    const bool r_il = p.r_il8 != 0, has_r = p.resid != nullptr;
    const int64_t ybase = (int64_t)n * p.y_bstride + (YIL ? z.y_off_il8 : z.y_off);
    const int64_t rbase = (int64_t)n * p.r_bstride + (r_il ? z.r_off_il4 : z.r_off);
    const int cs32 = (int)p.y_cstride, rs32 = (int)p.r_cstride;
    const int my = ty0 + wv, mx = tx0 + l31;
    const int Wx = p.y_ystride >> 1;
    const int act = p.act;
    const int cb = nblk * 32;
    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, ybase, 4));
    const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, rbase, 4), has_r);
    const bool right_edge = tx0 + Cfg::TX >= p.Wi;                    // wave-uniform: the tile holds the last output column (odd width: no pair)
    // (the activation as a compile-time constant of the epilogue: on a run-time `act`, apply_act_fast is a scalar branch tree per VALUE)
    auto epilogue = [&](auto actc) __attribute__((always_inline)) {
    constexpr int ACT = decltype(actc)::value;                         // -1: whatever p.act says, per value
#pragma unroll
    for (int py = 0; py < 2; py++) {
        const int oy = 2 * my + py, ox = 2 * mx;
        const bool row_ok = my < p.Hi && mx < p.Wi && oy < p.Ho;
        const bool ok0 = row_ok && ox < p.Wo, ok1 = row_ok && ox + 1 < p.Wo;
        const unsigned pix = (unsigned)(my * p.y_ystride + py * Wx + ox);
        f32x4 sk[2][4];
        if (has_r && r_il) {
#pragma unroll
            for (int px = 0; px < 2; px++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    sk[px][q] = buf_load4(rs_r, ((px ? ok1 : ok0) && cb + 8 * q + 4 * kg < p.Cout) ? ((pix + px) * 4u + (unsigned)(4 * kg * rs32)) * 4u : kBufOOB,
                                          (unsigned)((cb + 8 * q) * rs32) * 4u);
        } else if (has_r) {
#pragma unroll
            for (int px = 0; px < 2; px++)
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        sk[px][q][e] = buf_load(rs_r, ((px ? ok1 : ok0) && cb + 8 * q + 4 * kg + e < p.Cout) ? (pix + px + (unsigned)(4 * kg * rs32)) * 4u : kBufOOB,
                                                (unsigned)((cb + 8 * q + e) * rs32) * 4u);
        } else {
#pragma unroll
            for (int px = 0; px < 2; px++)
#pragma unroll
                for (int q = 0; q < 4; q++) sk[px][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + cb + 8 * q + 4 * kg);     // padded to 64 channels
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                // (conv_s3_kernel's epilogue, operation for operation: main + cross, then + (bias + skip))
                const float v0 = fmaf(acc_c[2 * py][4 * q + e], kSplitInv, acc_m[2 * py][4 * q + e]);
                const float v1 = fmaf(acc_c[2 * py + 1][4 * q + e], kSplitInv, acc_m[2 * py + 1][4 * q + e]);
                o0[e] = apply_act_fast(v0 + (bv[e] + sk[0][q][e]), ACT < 0 ? act : ACT);
                o1[e] = apply_act_fast(v1 + (bv[e] + sk[1][q][e]), ACT < 0 ? act : ACT);
            }
            if constexpr (YIL) {
                // the lane's 4 channels are one 16-byte slot of each of its two pixels
                const bool gok = cb + 8 * q + 4 * kg < p.Cout;
                const unsigned so = (unsigned)((cb + 8 * q) * cs32) * 4u;
                buf_store4(o0, rs_y, (ok0 && gok) ? (pix * 4u + (unsigned)(4 * kg * cs32)) * 4u : kBufOOB, so);
                buf_store4(o1, rs_y, (ok1 && gok) ? ((pix + 1u) * 4u + (unsigned)(4 * kg * cs32)) * 4u : kBufOOB, so);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const unsigned vo = (pix + (unsigned)(4 * kg * cs32)) * 4u, so = (unsigned)((cb + 8 * q + e) * cs32) * 4u;
                    const bool cok = cb + 8 * q + 4 * kg + e < p.Cout;
                    // the pair (2 mx, 2 mx + 1): 8 bytes at a 4-byte aligned address (rows of odd width)
                    const float a0 = o0[e], a1 = o1[e];      // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0 with this clang)
                    const u32x2_t pair = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1)};
                    __builtin_amdgcn_raw_buffer_store_b64(pair, rs_y, (ok1 && cok) ? vo : kBufOOB, so, 0);
                    if (right_edge) buf_store(a0, rs_y, (ok0 && !ok1 && cok) ? vo : kBufOOB, so);
                }
            }
        }
    }
    };
    if (act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (act == 0) epilogue(std::integral_constant<int, 0>{});
    else epilogue(std::integral_constant<int, -1>{});
}

}  // namespace rt
