// Winograd F(2x2, 3x3) convolution on the gfx950 matrix cores, fp32 (v_mfma_f32_16x16x4_f32).
//
// Serves the stride-1 3x3 windows of the hot path -- 40 of the 49 launches of ResNet-18 2D (reference
// resnet18_2D_513x257_net.cpp:48-719) and, through the same gather table as conv_mfma_f32_kernel, the
// (D*C)-merged 3x3x3 Conv3DPlugin layers (reference lib/conv3d_plugin.cpp:187-216) -- with 16 instead of 36
// multiplies per 2x2 output tile and input channel (the minimal-filtering algorithm cuDNN / TensorRT pick for
// fp32 3x3 convolutions on the reference's side):
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        d = 4x4 input tile, g = 3x3 filter, Y = 2x2 outputs
// On gfx950 the fp32 MFMA and the vector ALU share one issue slot (tools/micro/mfma_valu.hip), so the contraction
// is priced in MFMA cycles + VALU cycles; this formulation needs 16 MFMA k-steps of 32 cycles per 4 input
// channels and 16 cout x 16 tiles, against 36 equivalent ones for the direct form.
//
// GEMM view: one GEMM per Winograd position p = 4*pr + pc:   M_p[cout][tile] = sum_cin U_p[cout][cin] V_p[cin][tile]
//   A operand = U_p (G g G^T, transformed on the host, exact: G holds only 0, 1, +-1/2)
//   B operand = V_p (B^T d B), computed by the lane that feeds it: lane (k4 = lane>>4, t = lane&15) owns tile t
//               of its wave and input channel 4*j + k4 -> 16 LDS words in, 32 adds, 16 MFMA operands out.
//   Accumulators: 16 positions x 4 registers (16 cout x 16 tiles per wave) stay live over all input channels,
//               so the output transform A^T M A runs once, in the epilogue (24 adds per output 2x2).
// Bias and the residual (skip connection) are folded into the accumulator init: M_0, M_3, M_12, M_15 are the
// positions that reach exactly one of the four outputs (with signs +, -, -, +).
//
// Workgroup = NW waves = (NW/2 tile rows) x (2 blocks of 16 output channels): tile = NW rows x 32 pixels x 32
// channels.  LDS per chunk of 8 input channels: raw patch [ch][row][36] (channel stride = 32 mod 64 words, so the
// ds_read_b64 of the four k4 groups hit disjoint banks) + transformed weights [k-step][k4][p/4][cout][p%4]
// (one conflict-free ds_read_b128 per 4 positions).  Staging and addressing as in conv_mfma.hip.h.
#pragma once
#include <type_traits>
#include "common.hip.h"
#include "conv_mfma.hip.h"

namespace rt {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef RT_WINO_MINW
#define RT_WINO_MINW 4
#endif
#ifndef RT_WINO_LDS_DMA
// 1: transformed weights go global -> LDS without passing through registers (buffer_load_dwordx4 ... lds, double-
// buffered slabs, 40 KB of LDS per workgroup).  Measured on the 32->32 layer with interleaved tensors: 145 vs 132 us at
// batch 8, 21.1 vs 19.9 us at batch 1 -- slower (the freed registers are re-used by the scheduler up to the 128 limit,
// with 1-8 spills), so the register-staged form is built.
#define RT_WINO_LDS_DMA 0
#endif
// (The seven in-kernel probes of the round-4 hunt for the multi-context deviation of the interleaved instantiations -- idle cycles,
// waits and operand orders around the MFMAs, stores and residual loads -- are described with their results in profiles/r04_race.txt;
// they left the product source in round 5: git show 49a042d:redtail_amd/csrc/kernels/conv_wino.hip.h has them.)
template <int NW>
struct WinoCfg {
    static constexpr int CC = 8;                          // input channels per chunk (2 MFMA k-steps of 4)
    static constexpr int TR = NW / 2;                     // tile rows (of 2 output rows) per workgroup
    static constexpr int TY = 2 * TR, TX = 32;            // output tile
    static constexpr int PR = TY + 2, PC = TX + 2;        // input patch
    static constexpr int PCP = 36;                        // LDS row pitch (words)
    static constexpr int CHS = ((PR * PCP - 32 + 63) / 64) * 64 + 32;   // channel stride: >= PR*PCP, = 32 (mod 64)
    static constexpr int NPIX = PR * PC;
    static constexpr int CPW = CC / NW > 0 ? CC / NW : 1; // channels staged per wave
    static constexpr int NKP = (NPIX + 63) / 64;          // patch pixels per lane and channel
    static constexpr int U_ELEMS = 2 * 4 * 4 * 32 * 4;    // transformed weights per chunk and 32-channel block
    static constexpr int NTHR = 64 * NW;
    static constexpr int NK_W = U_ELEMS / 4 / NTHR;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(CC % NW == 0 || NW == 8, "");
    static_assert(U_ELEMS / 4 % NTHR == 0, "weight slab is copied as whole float4 rounds");
};

// TIN / TOUT: storage type of the input and of the output + residual (float, or _Float16 in half2 mode)
// XIL / YIL (fp32 tensors, NW = 4): the input / output tensor is channel-interleaved, (C/4, H, pitch, 4) -- one 16-byte
// slot per pixel and group of 4 channels, the fp32 form of the layout described in conv_f16.hip.h.  A lane's 4 output
// channels of a pixel are one 16-byte store (4 instead of 8 stores and residual loads), a gather is one 16-byte load
// per pixel and group (2 instead of 8 per lane and chunk).  The residual's layout is the run-time flag ConvArgs::r_il8
// (RRT: compiled in; the all-planar instantiation leaves it out -- it sits at the 128-register limit).
template <int NW, typename TIN = float, typename TOUT = float, bool XIL = false, bool YIL = false, bool RRT = (XIL || YIL)>
__global__ void __launch_bounds__(64 * NW, RT_WINO_MINW) conv_wino_f32_kernel(ConvArgs p) {
    using Cfg = WinoCfg<NW>;
    static_assert(!(XIL || YIL) || (NW == 4 && std::is_same<TIN, float>::value && std::is_same<TOUT, float>::value),
                  "interleaved tensors: fp32, 4-wave tile");
    constexpr unsigned ESX = Io<TIN>::ES, ESY = Io<TOUT>::ES;
    constexpr int CC = Cfg::CC, TY = Cfg::TY, TX = Cfg::TX, PC = Cfg::PC, PCP = Cfg::PCP, CHS = Cfg::CHS;
    constexpr int NPIX = Cfg::NPIX, CPW = Cfg::CPW, NKP = Cfg::NKP, NTHR = Cfg::NTHR, NK_W = Cfg::NK_W;

    __shared__ __attribute__((aligned(16))) float sIn[CC * CHS];
    // Transformed weights: with WDMA the 16 KB slab of a chunk goes from global memory straight into LDS
    // (buffer_load_dwordx4 ... lds: lane l of a wave writes 16 bytes at M0 base + 16*l, i.e. a straight copy of the
    // packed slab), double-buffered so that the slab of chunk ch+1 lands while chunk ch is being multiplied.  No
    // VGPRs hold weights in flight (16 fewer) and no ds_write instructions stage them.
    constexpr bool WDMA = RT_WINO_LDS_DMA != 0 && NW == 4;
    __shared__ __attribute__((aligned(16))) float sU[(WDMA ? 2 : 1) * Cfg::U_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int k4 = lane >> 4, t = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wv & 1, tg = wv >> 1;          // 16-channel block / tile row of this wave

    int tile = blockIdx.x;
    if (p.xcd_order) {                            // contiguous tile range per XCD (see conv_mfma.hip.h)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx0 = (tile % p.tiles_x) * TX;
    const int ty0 = (tile / p.tiles_x) * TY;
    const int nblk = blockIdx.y;
    const int zi = blockIdx.z % p.nz;
    const int n = blockIdx.z / p.nz;
    const char* __restrict__ xb = elem_ptr(p.x, (int64_t)n * p.x_bstride, ESX);
    const int nchunks = p.CinPad / CC;
    const int Ho = p.Ho, Wo = p.Wo;
    const int64_t ybase = (int64_t)n * p.y_bstride + p.y_off + (int64_t)zi * p.y_zstride;
    // the residual has its own per-sample stride: it may be a channel range of a concatenated buffer (engine.cpp: foldConcats) while the
    // output is a dense tensor -- round 3: with the output's stride, sample 1 of a batch read the wrong rows (exact-fp32 engines only)
    const int64_t rbase = (int64_t)n * p.r_bstride + p.y_off + (int64_t)zi * p.y_zstride;
    const int act = p.act;

    // ---- staging roles: wave w gathers channels w*CPW .. of each chunk (NW = 8: one channel per wave) ----------
    // interleaved input: wave w gathers channel group w & 1 (4 channels) for half w >> 1 of the patch pixels
    constexpr int NKX = (NPIX + 127) / 128;
    constexpr int NV = XIL ? NKX : NKP;
    const int xg = wv & 1;
    const int* __restrict__ tab = p.ch_off + (int64_t)zi * p.CinPad + (XIL ? 4 * xg : (wv % CC) * CPW);
    unsigned voff[NV];
    int loff[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int pidx = XIL ? (wv >> 1) * (NKX * 64) + lane + 64 * k : lane + 64 * k;
        const int pr = pidx / PC, pc = pidx - pr * PC;
        const int iy = ty0 - p.pad_y + pr, ix = tx0 - p.pad_x + pc;
        const bool own = pidx < NPIX;
        voff[k] = (own && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) ? (unsigned)(iy * p.x_pitch + ix) * (XIL ? 16u : ESX) : kBufOOB;
        loff[k] = own ? (XIL ? 4 * xg : (wv % CC) * CPW) * CHS + pr * PCP + pc : -1;
    }
    const float* __restrict__ wsrc = p.w + ((int64_t)nblk * nchunks) * Cfg::U_ELEMS;
    const buf_rsrc rs_w = make_buf(wsrc);

    float rin[XIL ? 1 : CPW][XIL ? 1 : NKP];
    f32x4 rin4[XIL ? NKX : 1];
    f32x4 rw[WDMA ? 1 : NK_W];
    auto weights_to_lds = [&](int ch) {                   // WDMA: slab of chunk ch -> buffer ch & 1
        const unsigned so = (unsigned)ch * (unsigned)(Cfg::U_ELEMS * 4);
#pragma unroll
        for (int k = 0; k < NK_W; k++) {
            float* dst = sU + (ch & 1) * Cfg::U_ELEMS + (wv * 64 + NTHR * k) * 4;      // wave-uniform base, lane l lands at + 16*l bytes
            if (!kAblWLoad)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, RT_LDS_PTR(dst), 16, (unsigned)(tid + NTHR * k) * 16u, so, 0, 0);
        }
    };
    auto prefetch = [&](int ch) {
        if constexpr (XIL) {
            const int off = tab[ch * CC];                 // group offset == planar offset of its first channel
            const buf_rsrc rs = make_buf(xb, off >= 0);
#pragma unroll
            for (int k = 0; k < NKX; k++)
                rin4[k] = kAblGather ? f32x4{(float)(off + (int)voff[k]), 0.f, 0.f, 0.f} : buf_load4(rs, voff[k], (unsigned)off * 4u);
        } else {
#pragma unroll
            for (int q = 0; q < CPW; q++) {
                const int off = tab[ch * CC + q];         // wave-uniform scalar load
                const buf_rsrc rs = make_buf(xb, off >= 0);
                const unsigned so = (unsigned)off * ESX;
#pragma unroll
                for (int k = 0; k < NKP; k++) rin[q][k] = kAblGather ? (float)(off + (int)voff[k]) : Io<TIN>::load(rs, voff[k], so);
            }
        }
        if constexpr (WDMA) {
            weights_to_lds(ch);
        } else {
            const unsigned so = (unsigned)ch * (unsigned)(Cfg::U_ELEMS * 4);
#pragma unroll
            for (int k = 0; k < NK_W; k++) rw[k] = kAblWLoad ? f32x4{(float)tid, 1.f, 2.f, (float)ch} : buf_load4(rs_w, (unsigned)(tid + NTHR * k) * 16u, so);
        }
    };
    auto stage_to_lds = [&]() {
        if constexpr (XIL) {
#pragma unroll
            for (int k = 0; k < NKX; k++)
                if (loff[k] >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) sIn[loff[k] + j * CHS] = rin4[k][j];
                }
        } else {
#pragma unroll
            for (int q = 0; q < CPW; q++)
#pragma unroll
                for (int k = 0; k < NKP; k++)
                    if (loff[k] >= 0) sIn[loff[k] + q * CHS] = rin[q][k];
        }
        if constexpr (WDMA) {
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's part of the slab is in LDS
        } else {
#pragma unroll
            for (int k = 0; k < NK_W; k++) reinterpret_cast<f32x4*>(sU)[tid + NTHR * k] = rw[k];
        }
    };

    // ---- output addressing + accumulator init (bias, residual) -------------------------------------------------
    // lane (k4, t) ends up with channels cbase + i (i = 0..3) of tile t: outputs (2*tg + a, 2*t + b)
    const int cs32 = (int)p.y_cstride;
    const int cbase = nblk * 32 + cb * 16 + 4 * k4;       // + i
    const bool tail4 = (p.Cout & 3) != 0;                 // only then does validity depend on i
    // each lane owns output pixel pairs (2t, 2t+1) of rows 2*tg + a: one 8-byte access per pair; the pair that
    // straddles the right image edge (odd widths) falls back to a 4-byte access behind a scalar branch
    const bool edge_tile = tx0 + TX > Wo;                 // wave-uniform
    // (recomputed in the epilogue instead of being kept live across the main loop: 4 registers at the 128-VGPR limit)
    unsigned yv2[2], yv1[2];
    const bool r_il = RRT && p.r_il8 != 0;                // uniform
    // interleaved tensors: the 16-byte slot of pixel (a, b) of this lane's tile; the channel group goes into soffset
    auto il_off = [&](int a, int b) {
        const int oy = ty0 + 2 * tg + a, ox = tx0 + 2 * t + b;
        return (oy < Ho && ox < Wo && cbase < p.Cout) ? (unsigned)((oy * p.y_ystride + ox) * 4 + (cb * 16 + 4 * k4) * cs32) * 4u : kBufOOB;
    };
    auto out_offsets = [&]() {
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int oy = ty0 + 2 * tg + a, ox = tx0 + 2 * t;
            const bool row_ok = oy < Ho && cbase < p.Cout;
            const unsigned off = (unsigned)(oy * p.y_ystride + ox * p.y_xstride + (cb * 16 + 4 * k4) * cs32) * ESY;
            yv2[a] = (row_ok && ox + 1 < Wo) ? off : kBufOOB;
            yv1[a] = (row_ok && ox + 1 == Wo) ? off : kBufOOB;
        }
    };
    out_offsets();
    // residual values are requested first (HBM latency overlaps the first gather) and consumed after prefetch(0)
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + cbase);      // bias is padded to 64 channels
    float rr[4][2][2];
    if (r_il) {
        const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, rbase, ESY), p.resid != nullptr);
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const f32x4 v = kAblResid ? f32x4{0.f, 1.f, 2.f, 3.f} : buf_load4(rs_r, il_off(a, b), (unsigned)(nblk * 32 * cs32) * 4u);
#pragma unroll
                for (int i = 0; i < 4; i++) rr[i][a][b] = v[i];
            }
    } else {
        const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, rbase, ESY), p.resid != nullptr);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned so = (unsigned)((nblk * 32 + i) * cs32) * ESY;
            const bool dead = tail4 && cbase + i >= p.Cout;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const f32x2 v = kAblResid ? f32x2{(float)i, (float)a} : Io<TOUT>::load2(rs_r, dead ? kBufOOB : yv2[a], so);
                rr[i][a][0] = v[0];
                rr[i][a][1] = v[1];
            }
        }
        if (edge_tile) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned so = (unsigned)((nblk * 32 + i) * cs32) * ESY;
                const bool dead = tail4 && cbase + i >= p.Cout;
#pragma unroll
                for (int a = 0; a < 2; a++) rr[i][a][0] += Io<TOUT>::load(rs_r, dead ? kBufOOB : yv1[a], so);
            }
        }
    }
    f32x4 acc[16];

    // ---- one chunk: 2 k-steps x (input transform of this lane's tile + 16 MFMAs) --------------------------------
    const float* dbase = sIn + k4 * CHS + (2 * tg) * PCP + 2 * t;
    const f32x4* ubase = reinterpret_cast<const f32x4*>(sU) + (k4 * 4) * 32 + cb * 16 + t;
    auto compute = [&](int ub) {                          // ub: weight buffer of this chunk, in 16-byte slots
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float* dp = dbase + (4 * j) * CHS;
            f32x2 d[4][2];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                d[r][0] = *reinterpret_cast<const f32x2*>(dp + r * PCP);
                d[r][1] = *reinterpret_cast<const f32x2*>(dp + r * PCP + 2);
            }
            f32x4 a4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) a4[q] = ubase[ub + (j * 16 + q) * 32];
            // rows: w = B^T d  (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]), two columns at a time
            f32x2 w[4][2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                w[0][h] = d[0][h] - d[2][h];
                w[1][h] = d[1][h] + d[2][h];
                w[2][h] = d[2][h] - d[1][h];
                w[3][h] = d[1][h] - d[3][h];
            }
            // columns: V = w B   (written per element: the packed form of this step measured 4 % slower)
            float v[16];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const f32x2 lo = kAblXform ? d[r][0] : w[r][0], hi = kAblXform ? d[r][1] : w[r][1];
                const float x0 = lo[0], x1 = lo[1], x2 = hi[0], x3 = hi[1];
                if (kAblXform) { v[4 * r] = x0; v[4 * r + 1] = x1; v[4 * r + 2] = x2; v[4 * r + 3] = x3; continue; }
                v[4 * r + 0] = x0 - x2;
                v[4 * r + 1] = x1 + x2;
                v[4 * r + 2] = x2 - x1;
                v[4 * r + 3] = x1 - x3;
            }
#pragma unroll
            for (int q = 0; q < 16; q++)
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q >> 2][q & 3], v[q], acc[q], 0, 0, 0);
        }
    };

    prefetch(0);
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        acc[0][i] = rr[i][0][0] + bv[i];
        acc[3][i] = -(rr[i][0][1] + bv[i]);
        acc[12][i] = -(rr[i][1][0] + bv[i]);
        acc[15][i] = rr[i][1][1] + bv[i];
    }
    for (int ch = 0; ch < nchunks; ch++) {
        if (ch) __syncthreads();       // everyone finished reading the previous chunk from LDS
        stage_to_lds();
        __syncthreads();
        if (ch + 1 < nchunks) prefetch(ch + 1);
        compute(WDMA ? (ch & 1) * (Cfg::U_ELEMS / 4) : 0);
    }

    // ---- epilogue: Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]), activation, stores ---------------------------------
    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, ybase, ESY));
    asm volatile("" ::: "memory");     // keep the recomputation below the loop
    out_offsets();
    auto epilogue = [&](auto ACT) {
        float yo[YIL ? 4 : 1][2][2];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float t0[4], t1[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                t0[c] = acc[c][i] + acc[4 + c][i] + acc[8 + c][i];
                t1[c] = acc[4 + c][i] - acc[8 + c][i] - acc[12 + c][i];
            }
            float y[2][2];
            y[0][0] = t0[0] + t0[1] + t0[2];
            y[0][1] = t0[1] - t0[2] - t0[3];
            y[1][0] = t1[0] + t1[1] + t1[2];
            y[1][1] = t1[1] - t1[2] - t1[3];
            if constexpr (YIL) {
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) yo[i][a][b] = apply_act_fast(y[a][b], decltype(ACT)::value);
            } else {
                const unsigned so = (unsigned)((nblk * 32 + i) * cs32) * ESY;
                const bool dead = tail4 && cbase + i >= p.Cout;
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const f32x2 o = {apply_act_fast(y[a][0], decltype(ACT)::value), apply_act_fast(y[a][1], decltype(ACT)::value)};
                    if (!kAblStore || o[0] == 12345.678f) Io<TOUT>::store2(o, rs_y, dead ? kBufOOB : yv2[a], so);
                    if (edge_tile) Io<TOUT>::store(o[0], rs_y, dead ? kBufOOB : yv1[a], so);
                }
            }
        }
        if constexpr (YIL) {                              // the lane's 4 channels of each of its 4 pixels: 16 bytes
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const f32x4 o = {yo[0][a][b], yo[1][a][b], yo[2][a][b], yo[3][a][b]};
                    if (!kAblStore || o[0] == 12345.678f)
                        buf_store4(o, rs_y, il_off(a, b), (unsigned)(nblk * 32 * cs32) * 4u);
                }
        }
    };
    if (act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (act == 2) epilogue(std::integral_constant<int, 2>{});
    else epilogue(std::integral_constant<int, 0>{});
}

}  // namespace rt
