// Implicit-GEMM convolution with fp16 operands on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate):
// the half2-mode form of conv_mfma_f32_kernel / conv_wino_f32_kernel for layers whose input and output tensors are
// both stored as fp16 (TensorRT half2 mode, IBuilder::setHalf2Mode, reference sample_app/main.cpp:256-262; weight
// file trt_weights_fp16.bin).  Operands are exactly the stored values -- fp16 activations, fp16 weights -- and the
// products are accumulated in fp32, so nothing is rounded that the fp32-arithmetic path of half2 mode does not round
// too; the matrix cores just run 16x faster (2.5 PFLOP/s dense) and the layer becomes bound by data movement.
//
// Same contraction description as conv_mfma.hip.h (tap window, gather table, ZSlice phases, tap masks, residual and
// bias in the accumulator init, buffer addressing).  Per chunk of 16 input channels:
//   LDS patch  [row][h][col][8 halfs]   channel = 8*h + e   -> one ds_read_b128 = the B operand of a lane (pixel, h)
//   LDS weights [tap][h][co][8 halfs]                        -> one ds_read_b128 = the A operand of a lane (co, h)
// and ONE MFMA (K = 16) per tap.  Staging: wave w gathers channel group h = w&1 for half of the patch, one 4-byte
// load per channel and PIXEL PAIR, transposed in registers to two 16-byte LDS writes.  Tile: 4 rows x 32 pixels x
// 32 channels, 4 waves.
//
// Where the time goes (MI355X, 32->32 @185x629 + residual + ELU, batch 8; HBM floor ~18 us, MFMA ~8 us):
//  * planar tensors, 45 us.  tools/time_phases.py 8 f16 (s_memtime stamps): of a workgroup's 33 k cycles, 17 k pass
//    before its first 31 vector-memory instructions (16 residual, 4 bias, 8 gathers, 3 weights) are ISSUED and 8 k
//    while the 11 of the second chunk are; data is there ~100 cycles after the issue completes.  With 32 waves per CU
//    that is ~16 cycles of the CU's memory front end per wave-instruction: the kernel is bound by memory instructions
//    x cache lines touched, not by bytes -- a planar fp16 row of a 32-pixel tile is 64 B, half a line.
//  * tools/ablate_conv.py runocc16: 80 / 67 / 61 / 58 / 51 us at 4 .. 8 waves per SIMD -> registers are budgeted
//    for 8 (RT_F16_WAVES); run16: 30 us remain with every global access compiled out, 50 without the MFMAs.
//  * pixel-PAIR gathers (the planar input path) halved the gather instructions: 51.5 -> 45.3 us.  Measured and
//    rejected: (a) 8-byte pixel-quad gathers + pixel-pair stores through a DPP lane swap: 112 VGPRs -> 4 waves, -17 %,
//    and the pair stores alone cost 13 % (same half-lines per plane, plus the swap); (b) a persistent grid with
//    LDS-resident weights and next-chunk prefetch: 2x slower (4 waves, tiles of a workgroup serialise).
//  * channel-interleaved tensors (XIL8 / YIL8 below; the executor's layout inside the towers): 22 instead of 58
//    memory instructions per wave, whole lines: 45 -> 37 us (workgroup lifetime 33 k -> 22 k cycles), HBM roofline
//    fraction 0.45 -> 0.55 by bench.py's events.  What is left is the prologue (8 k cycles until 13 instructions are
//    issued) and the exposed latency of the first chunk.
#pragma once
#include <type_traits>
#include "common.hip.h"
#include "conv_mfma.hip.h"

namespace rt {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// NW waves per workgroup = NW output rows per tile.  8 (weights staged once per 8 rows, 10 patch rows for 8 output
// rows) was measured on the interleaved form: no difference (41.7 us per layer at batch 8 both ways); 4 is built.
template <int KH, int KW, int S, int NW = 4>
struct ConvF16Cfg {
    static constexpr int TY = NW, TX = 32, CC = 16, TAPS = KH * KW, NT = 64 * NW, NP = NW / 2;
    static constexpr int PR = (TY - 1) * S + KH, PC = (TX - 1) * S + KW;
    static constexpr int PCA = (PC + 1 + 1) / 2 * 2;              // LDS columns: alignment offset (0 / 1) + patch, whole pairs
    static constexpr int NQ = PCA / 2, TG = PR * NQ;              // pixel pairs per row / per channel group
    static constexpr int NKT = (TG + NP * 64 - 1) / (NP * 64);    // pairs per lane (NP waves per channel group)
    static constexpr int NPIX = PR * PC, NKP = (NPIX + NP * 64 - 1) / (NP * 64);   // interleaved input: patch pixels per lane
    static constexpr int W_SLOTS = TAPS * 2 * 32;                 // 16-byte slots of the weight slab of one chunk
    static constexpr int NK_W = (W_SLOTS + NT - 1) / NT;
};

// throughput follows occupancy here (tools/ablate_conv.py runocc16: 80 / 67 / 61 / 58 / 51 us at 4 .. 8 waves per SIMD)
#ifndef RT_F16_WAVES
#define RT_F16_WAVES 8
#endif
// XIL8 / YIL8: the input / output tensor is channel-interleaved, (C/8, H, pitch, 8): one 16-byte slot per pixel and
// group of 8 channels (the executor's layout for fp16 tensors that only this kernel touches).  A gather is then ONE
// 16-byte load per pixel and group that goes to LDS as it is, a lane stores its 4 consecutive channels of a pixel as
// 8 bytes, and every cache line is used in full: ~230 lines per tile instead of ~540 half-used ones.  The offset of
// channel group c/8 equals the planar offset of channel c, so gather table and channel strides are shared.
// The residual's layout is a run-time flag (ConvArgs::r_il8): a block's skip connection may be either.
template <int KH, int KW, int S, bool XIL8 = false, bool YIL8 = false, int NW = 4>
__global__ void __launch_bounds__(64 * NW) RT_WAVES_PER_EU(RT_F16_WAVES) conv_f16mma_kernel(ConvArgs p) {
    using Cfg = ConvF16Cfg<KH, KW, S, NW>;
    constexpr int NT = Cfg::NT;
    constexpr int TY = Cfg::TY, TX = Cfg::TX, CC = Cfg::CC, TAPS = Cfg::TAPS, PCA = Cfg::PCA, NQ = Cfg::NQ;
    constexpr int NKT = Cfg::NKT, NK_W = Cfg::NK_W;
    constexpr int NV = XIL8 ? Cfg::NKP : NKT;     // gather positions per lane (pixels / pixel pairs)
    constexpr int NS = XIL8 ? Cfg::NKP : 2 * NKT; // 16-byte LDS slots per lane
    constexpr unsigned ES = 2;

    __shared__ __attribute__((aligned(16))) f32x4 sIn[Cfg::PR * 2 * PCA];
    __shared__ __attribute__((aligned(16))) f32x4 sW[Cfg::W_SLOTS];

    const int tid = threadIdx.x;
#ifdef RT_KERNEL_TIMING
    unsigned long long* dbgp = p.dbg ? p.dbg + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 : nullptr;
    int dbi = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    RT_TSTAMP();                                  // 0: start
    const int lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    RT_WG_TILE_NB(p, tile, zi, n, nblk)
    const int tx0 = (tile % p.tiles_x) * TX;
    const int ty0 = (tile / p.tiles_x) * TY;
    const char* __restrict__ xb = elem_ptr(p.x, (int64_t)n * p.x_bstride, ES);
    const int nchunks = p.CinPad / CC;
    int pad_y = p.pad_y, pad_x = p.pad_x, Ho = p.Ho, Wo = p.Wo, ch_row = zi;
    int64_t y_off = p.y_off + (int64_t)zi * p.y_zstride, w_off = 0, r_off = y_off;
    unsigned tap_mask = ~0u;
    if (p.zs) {
        const ZSlice z = p.zs[zi];
        pad_y = z.pad_y; pad_x = z.pad_x; Ho = z.Ho; Wo = z.Wo; ch_row = z.ch_row; w_off = z.w_off;
        y_off = YIL8 ? z.y_off_il8 : z.y_off;          // phases of a transposed convolution: the pixel part of an interleaved offset counts 8 elements
        r_off = p.r_il8 ? z.r_off_il8 : z.r_off;
        tap_mask = z.tap_mask;
    }
    const int act = p.act;

    // ---- staging roles: wave w -> channel group g = w & 1 (channels 8g .. 8g+7 of the chunk), pair range w >> 1 ------
    // A lane gathers PIXEL PAIRS (one 4-byte load per channel: half the loads of a per-pixel gather and no register
    // holds a lone half).  Pairs start at even columns -- the row pitch is even -- so the LDS patch starts at the
    // even column at or below the first patch column.
    const int g = wv & 1, spart = wv >> 1;
    const int* __restrict__ tab = p.ch_off + (int64_t)ch_row * p.CinPad + 8 * g;
    const int ix0 = tx0 * S - pad_x, ax0 = ix0 & ~1, dcol = ix0 - ax0;
    unsigned voff[NV];
    bool odd_ok[NV];         // the pair's second pixel is inside the row (columns >= Wi of a pitched row hold anything)
    int lidx[NV];
    int xcol[NV];            // interleaved input: image column of the slot (folded cost volume: a channel group may be read shifted)
    const int* __restrict__ shtab = p.ch_shift ? p.ch_shift + (int64_t)ch_row * p.CinPad + 8 * g : nullptr;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        if constexpr (XIL8) {
            const int pidx = spart * (Cfg::NKP * 64) + lane + 64 * k;
            const int pr = pidx / Cfg::PC, pc = pidx - pr * Cfg::PC;
            const int iy = ty0 * S - pad_y + pr, ix = ix0 + pc;
            const bool own = pidx < Cfg::NPIX;
            voff[k] = (own && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) ? (unsigned)(iy * p.x_pitch + ix) * 16u : kBufOOB;
            odd_ok[k] = false;
            xcol[k] = ix;
            lidx[k] = own ? (pr * 2 + g) * PCA + dcol + pc : -1;
        } else {
            const int q = spart * (NKT * 64) + lane + 64 * k;
            const int pr = q / NQ, qc = q - pr * NQ;
            const int iy = ty0 * S - pad_y + pr, ix = ax0 + 2 * qc;
            const bool own = q < Cfg::TG;
            voff[k] = (own && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) ? (unsigned)(iy * p.x_pitch + ix) * ES : kBufOOB;
            odd_ok[k] = ix + 1 < p.Wi;
            xcol[k] = ix;
            lidx[k] = own ? (pr * 2 + g) * PCA + 2 * qc : -1;
        }
    }
    // w_off and the slab size count 16-byte slots here (8 halfs)
    const char* __restrict__ wsrc = reinterpret_cast<const char*>(p.w) + (w_off + ((int64_t)nblk * nchunks) * Cfg::W_SLOTS) * 16;
    const buf_rsrc rs_w = make_buf(wsrc);
    unsigned wvoff[NK_W];
#pragma unroll
    for (int k = 0; k < NK_W; k++) {
        const int idx = tid + NT * k;
        wvoff[k] = idx < Cfg::W_SLOTS ? (unsigned)idx * 16u : kBufOOB;
    }

    f32x4 rin[NS];           // 8 halfs (channels) per patch pixel; planar input: [pair][pixel]
    f32x4 rw[NK_W];
    auto prefetch = [&](int ch) {
        if constexpr (XIL8) {
            const int off = tab[ch * CC];                          // group offset == planar offset of its first channel
            const buf_rsrc rs = make_buf(xb, off >= 0);
            const unsigned so = (unsigned)off * ES;
            // folded default cost volume (rtConv3dDesc::cv_fold): the right-image channel groups of depth slice d are read at column
            // x - d and are zero for x < d (lib/kernels.cu:72-97); one shift per group of 8 channels, wave-uniform
            const int sh = shtab ? shtab[ch * CC] : 0;
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const unsigned vo = sh == 0 ? voff[k] : ((voff[k] != kBufOOB && xcol[k] >= sh) ? voff[k] - (unsigned)sh * 16u : kBufOOB);
                rin[k] = kAblGather ? f32x4{(float)(off + (int)voff[k]), 0.f, 0.f, 0.f} : buf_load4(rs, vo, so);
            }
        } else {
            unsigned u[NKT][8];  // channel e of the chunk's group: pixel 0 in the low half, pixel 1 in the high half
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int off = tab[ch * CC + e];                  // wave-uniform scalar load
                const buf_rsrc rs = make_buf(xb, off >= 0);
                const unsigned so = (unsigned)off * ES;
#pragma unroll
                for (int k = 0; k < NKT; k++) u[k][e] = kAblGather ? (unsigned)(off + (int)voff[k]) : __builtin_amdgcn_raw_buffer_load_b32(rs, voff[k], so, 0);
            }
#pragma unroll
            for (int k = 0; k < NKT; k++) {                        // 8 x (2 pixels) -> 2 x (8 channels)
                u32x4_t p0, p1;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned a = u[k][2 * q], b = u[k][2 * q + 1];
                    p0[q] = (a & 0xffffu) | (b << 16);
                    p1[q] = odd_ok[k] ? ((a >> 16) | (b & 0xffff0000u)) : 0u;
                }
                rin[2 * k] = __builtin_bit_cast(f32x4, p0);
                rin[2 * k + 1] = __builtin_bit_cast(f32x4, p1);
            }
        }
        const unsigned so = (unsigned)ch * (unsigned)(Cfg::W_SLOTS * 16);
#pragma unroll
        for (int k = 0; k < NK_W; k++) rw[k] = kAblWLoad ? f32x4{(float)wvoff[k], 1.f, 2.f, (float)ch} : buf_load4(rs_w, wvoff[k], so);
    };
    auto stage_to_lds = [&]() {
#pragma unroll
        for (int k = 0; k < NV; k++)
            if (lidx[k] >= 0 && (!kAblLdsWr || rin[k][0] == 12345.678f)) {
                if constexpr (XIL8) {
                    sIn[lidx[k]] = rin[k];
                } else {
                    sIn[lidx[k]] = rin[2 * k];
                    sIn[lidx[k] + 1] = rin[2 * k + 1];
                }
            }
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            const int idx = tid + NT * k;
            if (idx < Cfg::W_SLOTS && (!kAblLdsWr || rw[k][0] == 12345.678f)) sW[idx] = rw[k];
        }
    };

    // ---- output addressing, accumulator init = bias + residual (as in conv_mfma.hip.h) ---------------------------
    const int64_t ybase = (int64_t)n * p.y_bstride + y_off;
    const int64_t rbase = (int64_t)n * p.r_bstride + r_off;
    const int cs32 = (int)p.y_cstride, rs32 = (int)p.r_cstride;
    const bool tail8 = (p.Cout & 7) != 0;
    const int oy = ty0 + wv, ox = tx0 + l31;
    const bool inb = oy < Ho && ox < Wo;
    const bool r_il8 = p.r_il8 != 0;              // uniform
    // planar: element (c, y, x) at c*cstride + y*ystride + x*xstride; interleaved: 16-byte pixel slots, lane's 4 channels
    const unsigned il8off = (unsigned)((oy * p.y_ystride + ox * p.y_xstride) * 8 + 4 * half) * ES;
    const unsigned yvoff = !inb ? kBufOOB : (YIL8 ? il8off : (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * half * cs32) * ES);
    const unsigned rvoff = !inb ? kBufOOB : (r_il8 ? il8off : (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * half * rs32) * ES);

    f32x16 acc;
    {
        const float* bsrc = p.bias + nblk * 32 + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bsrc + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; e++) acc[4 * q + e] = bv[e];
        }
    }
    float rv[16];
    if (r_il8) {
#pragma unroll
        for (int q = 0; q < 4; q++) {                               // 4 consecutive channels of the pixel: 8 bytes
            const int cs = nblk * 32 + 8 * q;
            const buf_rsrc rs = make_buf(elem_ptr(p.resid, rbase, ES), (p.resid != nullptr) & (cs < p.Cout));
            const u32x2_t u = kAblResid ? u32x2_t{0u, 0u} : __builtin_amdgcn_raw_buffer_load_b64(rs, rvoff, (unsigned)(cs * rs32) * ES, 0);
#pragma unroll
            for (int e = 0; e < 4; e++)
                rv[4 * q + e] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u[e >> 1] >> (16 * (e & 1))));
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int cs = nblk * 32 + (r & 3) + 8 * (r >> 2);
            const buf_rsrc rs = make_buf(elem_ptr(p.resid, rbase, ES), (p.resid != nullptr) & (cs < p.Cout));
            const unsigned vo = (tail8 && cs + 4 * half >= p.Cout) ? kBufOOB : rvoff;
            rv[r] = kAblResid ? (float)cs : Io<_Float16>::load(rs, vo, (unsigned)(cs * rs32) * ES);
        }
    }

    const int a_base = half * 32 + l31;
    const int b_base = (wv * S * 2 + half) * PCA + dcol + l31 * S;
    auto compute = [&]() {
#pragma unroll
        for (int t = 0; t < TAPS; t++) {
            if (!((tap_mask >> t) & 1u)) continue;                  // wave-uniform
            const int r = t / KW, s = t % KW;
            const f32x4 av = kAblLdsRd ? f32x4{(float)(a_base + t), 1.f, 2.f, 3.f} : sW[a_base + t * 64];
            const f32x4 bv = kAblLdsRd ? f32x4{(float)(b_base - t), 3.f, 2.f, 1.f} : sIn[b_base + r * 2 * PCA + s];
            const f16x8_t a = __builtin_bit_cast(f16x8_t, av);
            const f16x8_t b = __builtin_bit_cast(f16x8_t, bv);
            if (kAblMfma) acc[t & 15] += av[0] * bv[1];
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    };

    prefetch(0);
    RT_TSTAMP();                                  // 1: residual + first gathers issued
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] += rv[r];
    RT_TSTAMP();                                  // 2: residual arrived
    for (int ch = 0; ch < nchunks; ch++) {
        if (ch) wg_barrier();
        stage_to_lds();
        wg_barrier();
        RT_TSTAMP();                              // 3, 5: chunk in LDS (its gathers arrived)
        if (ch + 1 < nchunks) prefetch(ch + 1);
        compute();
        RT_TSTAMP();                              // 4, 6: MFMAs issued
    }

    auto epilogue = [&](auto ACT) {
        if constexpr (YIL8) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cs = nblk * 32 + 8 * q;
                const buf_rsrc rs = make_buf(elem_ptr(p.y, ybase, ES), cs < p.Cout);
                u32x2_t o;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = apply_act_fast(acc[4 * q + e], decltype(ACT)::value);
#pragma unroll
                for (int e = 0; e < 2; e++)
                    o[e] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * e]) |
                           ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * e + 1]) << 16);
                if (!kAblStore || v[0] == 12345.678f) __builtin_amdgcn_raw_buffer_store_b64(o, rs, yvoff, (unsigned)(cs * cs32) * ES, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int cs = nblk * 32 + (r & 3) + 8 * (r >> 2);
                const buf_rsrc rs = make_buf(elem_ptr(p.y, ybase, ES), cs < p.Cout);
                const unsigned vo = (tail8 && cs + 4 * half >= p.Cout) ? kBufOOB : yvoff;
                const float v = apply_act_fast(acc[r], decltype(ACT)::value);
                if (!kAblStore || v == 12345.678f) Io<_Float16>::store(v, rs, vo, (unsigned)(cs * cs32) * ES);
            }
        }
    };
    RT_TSTAMP();                                  // 7: epilogue start
    if (act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (act == 2) epilogue(std::integral_constant<int, 2>{});
    else epilogue(std::integral_constant<int, 0>{});
#ifdef RT_KERNEL_TIMING
    __builtin_amdgcn_s_waitcnt(0);
#endif
    RT_TSTAMP();                                  // 8: stores acknowledged
#ifdef RT_KERNEL_TIMING
    if (dbgp && tid == 0) dbgp[15] = __builtin_amdgcn_s_memrealtime() - rt0;
#endif
}

}  // namespace rt
