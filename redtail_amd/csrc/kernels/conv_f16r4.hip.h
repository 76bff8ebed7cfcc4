// 3x3 stride-1 convolution on fp16 operands, channel-interleaved tensors, FOUR output rows per wave (round 4).
//
// conv_f16mma_kernel (conv_f16.hip.h) gives every wave one row of a 4 x 32 tile: per tap and 16-channel chunk it reads its A operand
// (weights) AND its B operand (patch pixels) from LDS for ONE v_mfma_f32_32x32x16_f16 -- 2 KB of LDS reads per 32-cycle MFMA and wave,
// i.e. 256 B/clk per CU with the four SIMDs busy, which is all the LDS delivers (MI355X_MICROARCH.md: ds_read_b128, 256 B/clk): the
// Conv3D layers of the 3-D models in half2 mode (BASELINE C5: 27 * C * K multiplies per voxel, 1.5 of NVSmall's 2.7 ms) sat at
// 0.30-0.33 of the fp16 matrix peak.  MEASURED (round 4): this kernel is 3-6 % faster than that one on those layers and sits at 50-53 %
// MFMA-busy (profiles/r04_traffic_3d.json) -- the layers were not LDS-bandwidth-bound; what the bigger tile buys is halo and weight traffic.
//
// Here a wave owns 4 output rows x 32 pixels x 32 output channels (4 accumulators = 64 VGPRs) of a 16 x 32 workgroup tile:
//   * for a column shift s the three weight operands A[r][s] (r = 0..2) are read once and stay in 12 VGPRs;
//   * patch row p (0..5) at shift s is read once and feeds every (output row y, tap row r) with y + r = p: up to 3 MFMAs;
//   => 27 ds_read_b128 for 36 MFMAs per chunk (0.75 instead of 2 per MFMA);
//   * the 16-row tile reads 18 x 34 patch pixels for 16 x 32 outputs (1.19x halo instead of 1.59x for the 4-row tile);
//   * chunks are double-buffered in LDS (2 x 28.8 KB, two workgroups per CU): the next chunk's global loads fly under the MFMAs and
//     land in the other buffer -- one barrier per chunk.
// Same contraction description as conv_f16mma_kernel (gather table with depth taps merged into the channel axis, x-shift table of the
// folded cost volume, fp16 weight slabs [nblk][chunk][tap][h][co][8] -- the plan packs nothing new), same numerics (fp16 operands,
// fp32 accumulation, bias in the accumulator init, one rounding of the output).  Input and output must be channel-interleaved; the
// residual may be either (ConvArgs::r_il8).  Uniform slices only (no ZSlice phases).
#pragma once
#include <type_traits>
#include "common.hip.h"
#include "conv_mfma.hip.h"
#include "conv_f16.hip.h"

namespace rt {

struct ConvF16R4Cfg {
    static constexpr int NW = 4, RPW = 4, TY = NW * RPW, TX = 32, CC = 16;
    static constexpr int PR = TY + 2, PC = TX + 2;
    static constexpr int PCA = PC;                              // 16-byte slots per (row, channel group)
    static constexpr int GSLOTS = PR * PC;                      // patch slots of one channel group
    static constexpr int NKG = (GSLOTS + 255) / 256;            // ... per thread
    static constexpr int IN_SLOTS = PR * 2 * PCA;
    static constexpr int W_SLOTS = 9 * 2 * 32;
    static constexpr int NK_W = (W_SLOTS + 255) / 256;
    static constexpr int BUF_SLOTS = IN_SLOTS + W_SLOTS;
    static constexpr int LDS_BYTES = 2 * BUF_SLOTS * 16;
};

#ifndef RT_F16R4_WAVES
#define RT_F16R4_WAVES 2
#endif

__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(RT_F16R4_WAVES) conv_f16r4_kernel(ConvArgs p) {
    using Cfg = ConvF16R4Cfg;
    constexpr int PCA = Cfg::PCA, NKG = Cfg::NKG, NK_W = Cfg::NK_W, RPW = Cfg::RPW;
    constexpr unsigned ES = 2;
    __shared__ __attribute__((aligned(16))) f32x4 smem[2 * Cfg::BUF_SLOTS];

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    RT_WG_TILE_NB(p, tile, zi, n, nblk)
    const int tx0 = (tile % p.tiles_x) * Cfg::TX;
    const int ty0 = (tile / p.tiles_x) * Cfg::TY;
    const char* __restrict__ xb = elem_ptr(p.x, (int64_t)n * p.x_bstride, ES);
    const int nchunks = p.CinPad / Cfg::CC;
    const int Ho = p.Ho, Wo = p.Wo;
    const int64_t y_off = p.y_off + (int64_t)zi * p.y_zstride;

    // ---- staging: thread t owns patch slots t, t + 256, ... of EACH of the chunk's two channel groups and weight slots t, t + 256, ...
    const int* __restrict__ tab = p.ch_off + (int64_t)zi * p.CinPad;
    const int* __restrict__ shtab = p.ch_shift ? p.ch_shift + (int64_t)zi * p.CinPad : nullptr;
    unsigned voff[NKG];
    int xcol[NKG], lidx[NKG];
#pragma unroll
    for (int k = 0; k < NKG; k++) {
        const int s = tid + 256 * k;
        const int pr = s / Cfg::PC, pc = s - pr * Cfg::PC;
        const int iy = ty0 - p.pad_y + pr, ix = tx0 - p.pad_x + pc;
        const bool own = s < Cfg::GSLOTS;
        voff[k] = (own && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) ? (unsigned)(iy * p.x_pitch + ix) * 16u : kBufOOB;
        xcol[k] = ix;
        lidx[k] = own ? pr * 2 * PCA + pc : -1;
    }
    const char* __restrict__ wsrc = reinterpret_cast<const char*>(p.w) + ((int64_t)nblk * nchunks) * Cfg::W_SLOTS * 16;
    const buf_rsrc rs_w = make_buf(wsrc);
    f32x4 rin[2][NKG], rw[NK_W];
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int off = tab[ch * Cfg::CC + 8 * g];            // group offset == planar offset of its first channel; -1 = zeros
            const buf_rsrc rs = make_buf(xb, off >= 0);
            const unsigned so = (unsigned)off * ES;
            const int sh = shtab ? shtab[ch * Cfg::CC + 8 * g] : 0;   // folded cost volume: the group is read at column x - sh, zero for x < sh
#pragma unroll
            for (int k = 0; k < NKG; k++) {
                const unsigned vo = sh == 0 ? voff[k] : ((voff[k] != kBufOOB && xcol[k] >= sh) ? voff[k] - (unsigned)sh * 16u : kBufOOB);
                rin[g][k] = buf_load4(rs, vo, so);
            }
        }
        const unsigned so = (unsigned)ch * (unsigned)(Cfg::W_SLOTS * 16);
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + 256 * k;
            rw[k] = buf_load4(rs_w, idx < Cfg::W_SLOTS ? (unsigned)idx * 16u : kBufOOB, so);
        }
    };
    auto stage = [&](int buf) {
        f32x4* sIn = smem + buf * Cfg::BUF_SLOTS;
        f32x4* sW = sIn + Cfg::IN_SLOTS;
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int k = 0; k < NKG; k++)
                if (lidx[k] >= 0) sIn[lidx[k] + g * PCA] = rin[g][k];
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + 256 * k;
            if (idx < Cfg::W_SLOTS) sW[idx] = rw[k];
        }
    };

    // ---- accumulators: bias ------------------------------------------------------------------------------------------------------
    f32x16 acc[RPW];
    {
        const float* bsrc = p.bias + nblk * 32 + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bsrc + 8 * q);
#pragma unroll
            for (int y = 0; y < RPW; y++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[y][4 * q + e] = bv[e];
        }
    }

    const int a_base = half * 32 + l31;
    const int b_base = (wv * RPW * 2 + half) * PCA + l31;
    auto compute = [&](int buf) {
        const f32x4* sIn = smem + buf * Cfg::BUF_SLOTS;
        const f32x4* sW = sIn + Cfg::IN_SLOTS;
#pragma unroll
        for (int s = 0; s < 3; s++) {
            f16x8_t a[3];
#pragma unroll
            for (int r = 0; r < 3; r++) a[r] = __builtin_bit_cast(f16x8_t, sW[a_base + (r * 3 + s) * 64]);
#pragma unroll
            for (int pp = 0; pp < RPW + 2; pp++) {
                const f16x8_t b = __builtin_bit_cast(f16x8_t, sIn[b_base + pp * 2 * PCA + s]);
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const int y = pp - r;
                    if (y >= 0 && y < RPW) acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[r], b, acc[y], 0, 0, 0);
                }
            }
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

    // ---- epilogue: residual, activation, 8-byte stores of a lane's 4 consecutive channels per pixel -------------------------------------
    const int64_t ybase = (int64_t)n * p.y_bstride + y_off;
    const int64_t rbase = (int64_t)n * p.r_bstride + y_off;
    const int cs32 = (int)p.y_cstride, rs32 = (int)p.r_cstride;
    const bool r_il8 = p.r_il8 != 0;
    const bool has_r = p.resid != nullptr;
    const int ox = tx0 + l31;
    const int act = p.act;
    // (the activation as a compile-time constant of the epilogue: `apply_act_fast(x, act)` on a run-time `act` is a scalar branch tree per VALUE)
    auto epilogue = [&](auto actc) __attribute__((always_inline)) {
    constexpr int ACT = decltype(actc)::value;                         // -1: whatever p.act says, per value
#pragma unroll
    for (int y = 0; y < RPW; y++) {
        const int oy = ty0 + wv * RPW + y;
        const bool inb = oy < Ho && ox < Wo;
        const unsigned il8off = (unsigned)((oy * p.y_ystride + ox) * 8 + 4 * half) * ES;
        const unsigned yvoff = inb ? il8off : kBufOOB;
        if (has_r) {                                              // uniform
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cs = nblk * 32 + 8 * q;
                const buf_rsrc rs = make_buf(elem_ptr(p.resid, rbase, ES), cs < p.Cout);
                if (r_il8) {
                    const u32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(rs, yvoff, (unsigned)(cs * rs32) * ES, 0);
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[y][4 * q + e] += (float)__builtin_bit_cast(_Float16, (unsigned short)(u[e >> 1] >> (16 * (e & 1))));
                } else {
                    const unsigned pv = inb ? (unsigned)(oy * p.y_ystride + ox + 4 * half * rs32) * ES : kBufOOB;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        acc[y][4 * q + e] += Io<_Float16>::load(rs, (cs + 4 * half + e < p.Cout) ? pv : kBufOOB, (unsigned)((cs + e) * rs32) * ES);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int cs = nblk * 32 + 8 * q;
            const buf_rsrc rs = make_buf(elem_ptr(p.y, ybase, ES), cs < p.Cout);
            u32x2_t o;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = apply_act_fast(acc[y][4 * q + e], ACT < 0 ? act : ACT);
#pragma unroll
            for (int e = 0; e < 2; e++)
                o[e] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * e]) |
                       ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * e + 1]) << 16);
            __builtin_amdgcn_raw_buffer_store_b64(o, rs, yvoff, (unsigned)(cs * cs32) * ES, 0);
        }
    }
    };
    if (act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (act == 0) epilogue(std::integral_constant<int, 0>{});
    else epilogue(std::integral_constant<int, -1>{});
}

}  // namespace rt
