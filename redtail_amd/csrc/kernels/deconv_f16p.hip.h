// Stride-2 transposed 3x3(x3) convolution on fp16 operands, channel-interleaved tensors, ALL FOUR output phases per workgroup (round 4).
//
// The ZSlice form (conv_f16mma_kernel<2,2,1> with one slice per output phase, rt_capi.hip: build_deconv3d_subs) launches a workgroup
// per (tile, depth, PHASE): every phase gathers the same 5 x 33 input patch again and runs 1, 2, 2 or 4 of the 9 kernel taps on it --
// 37.5 KB staged and eight barriers per 36 MFMAs.  Conv3DTranspose layers of the 3-D decoders (reference
// lib/conv3d_transpose_plugin.cpp:205-243; NVSmall deconv3D_2: 55 GFLOP for 572 MB of tensors) are memory-bound: what counts is bytes
// and launches of work per byte.  Here a workgroup owns a 4 x 32 tile of the INPUT grid = an 8 x 64 tile of the output:
//   * the patch (5 x 33 pixels, 16 gathered channels per chunk) is staged ONCE and serves the 9 taps of the full 3 x 3 window;
//   * tap (ry, rx) contributes to exactly one phase: per dimension r = 1 -> even outputs from input m, r = 2 -> odd outputs from input m,
//     r = 0 -> odd outputs from input m + 1 (stride 2, pad 1: o = 2 i + r - 1) -- 9 MFMAs per wave and chunk into 4 accumulators
//     (one per phase), 4 patch operands (the 2 x 2 input offsets) + 9 weight operands from LDS;
//   * the weights are packed in KERNEL order [nblk][chunk][tap 9][h][co][8] -- no per-phase slabs padded with zero taps;
//   * chunks are double-buffered in LDS (2 x 14.7 KB) with one barrier per chunk, as in conv_f16r4.hip.h.
// Depth taps are merged into the gathered channel axis by the plan (one launch per output-depth class); bias, skip tensor
// (interleaved or planar), ELU and the optional fused Transform (ZSlice::y_off_il8) as in the ZSlice form.
#pragma once
#include "common.hip.h"
#include "conv_mfma.hip.h"
#include "conv_f16.hip.h"

namespace rt {

// tools/r05: instrumented builds (-DRT_DP_ABL=<mask>: 1 no global loads in the chunk loop after the first chunk, 2 no skip-tensor loads and
// no stores, 4 no MFMAs) -- where the time of the transposed layers goes
#ifndef RT_DP_ABL
#define RT_DP_ABL 0
#endif
constexpr int kDpAbl = RT_DP_ABL;

struct DeconvF16PCfg {
    static constexpr int TY = 4, TX = 32, CC = 16;
    static constexpr int PR = TY + 1, PC = TX + 1, PCA = PC + 1;    // patch rows / columns (+1 halo), LDS slots per (row, group)
    static constexpr int GSLOTS = PR * PC;
    static constexpr int NKG = (GSLOTS + 255) / 256;
    static constexpr int IN_SLOTS = PR * 2 * PCA;
    static constexpr int W_SLOTS = 9 * 2 * 32;
    static constexpr int NK_W = (W_SLOTS + 255) / 256;
    static constexpr int BUF_SLOTS = IN_SLOTS + W_SLOTS;
};

// p.zs: one ZSlice per output depth of the launch's class (phase (0, 0) of that depth: offsets, gather-table row); p.Ho / p.Wo: the
// FULL output plane (Hx, Wx); p.Hi / p.Wi: the input plane; p.y_ystride = 2 * Wx, p.y_xstride = 2.
__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(3) deconv_f16p_kernel(ConvArgs p) {
    using Cfg = DeconvF16PCfg;
    constexpr int PCA = Cfg::PCA, NKG = Cfg::NKG, NK_W = Cfg::NK_W;
    constexpr unsigned ES = 2;
    __shared__ __attribute__((aligned(16))) f32x4 smem[2 * Cfg::BUF_SLOTS];

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    RT_WG_TILE(p, tile, zi, n)
    const int tx0 = (tile % p.tiles_x) * Cfg::TX;
    const int ty0 = (tile / p.tiles_x) * Cfg::TY;
    const int nblk = blockIdx.y;
    const char* __restrict__ xb = elem_ptr(p.x, (int64_t)n * p.x_bstride, ES);
    const int nchunks = p.CinPad / Cfg::CC;
    const ZSlice z = p.zs[zi];

    const int* __restrict__ tab = p.ch_off + (int64_t)z.ch_row * p.CinPad;
    unsigned voff[NKG];
    int lidx[NKG];
#pragma unroll
    for (int k = 0; k < NKG; k++) {
        const int s = tid + 256 * k;
        const int pr = s / Cfg::PC, pc = s - pr * Cfg::PC;
        const int iy = ty0 + pr, ix = tx0 + pc;
        const bool own = s < Cfg::GSLOTS;
        voff[k] = (own && iy < p.Hi && ix < p.Wi) ? (unsigned)(iy * p.x_pitch + ix) * 16u : kBufOOB;
        lidx[k] = own ? pr * 2 * PCA + pc : -1;
    }
    const char* __restrict__ wsrc = reinterpret_cast<const char*>(p.w) + ((int64_t)nblk * nchunks) * Cfg::W_SLOTS * 16;
    const buf_rsrc rs_w = make_buf(wsrc);
    f32x4 rin[2][NKG], rw[NK_W];
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int off = tab[ch * Cfg::CC + 8 * g];            // element offset of the group's slot plane; -1 = zeros (depth tap out of range)
            const buf_rsrc rs = make_buf(xb, off >= 0);
#pragma unroll
            for (int k = 0; k < NKG; k++) rin[g][k] = buf_load4(rs, voff[k], (unsigned)off * ES);
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

    f32x16 acc[4];                                 // phase 2 * py + px
    {
        const float* bsrc = p.bias + nblk * 32 + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bsrc + 8 * q);
#pragma unroll
            for (int f = 0; f < 4; f++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[f][4 * q + e] = bv[e];
        }
    }
    const int a_base = half * 32 + l31;
    const int b_base = (wv * 2 + half) * PCA + l31;
    auto compute = [&](int buf) {
        const f32x4* sIn = smem + buf * Cfg::BUF_SLOTS;
        const f32x4* sW = sIn + Cfg::IN_SLOTS;
        f16x8_t b[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dx = 0; dx < 2; dx++) b[dy][dx] = __builtin_bit_cast(f16x8_t, sIn[b_base + dy * 2 * PCA + dx]);
#pragma unroll
        for (int ry = 0; ry < 3; ry++)
#pragma unroll
            for (int rx = 0; rx < 3; rx++) {
                // kernel tap r: 1 -> even output from input m; 2 -> odd output from m; 0 -> odd output from m + 1
                const int py = ry == 1 ? 0 : 1, dy = ry == 0 ? 1 : 0, px = rx == 1 ? 0 : 1, dx = rx == 0 ? 1 : 0;
                const f16x8_t a = __builtin_bit_cast(f16x8_t, sW[a_base + (ry * 3 + rx) * 64]);
                acc[2 * py + px] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[dy][dx], acc[2 * py + px], 0, 0, 0);
            }
    };

    // ---- output addressing.  The two x-phases of a lane are neighbouring pixels, and the two half-waves of a pixel hold its channels 0-3
    // and 4-7 of a group: one v_permlane32_swap per dword gives the lower half-wave the whole 16-byte slot of pixel 2 mx and the upper one
    // that of pixel 2 mx + 1, so a wave's store (and its skip-tensor load) covers 64 consecutive slots = 1 KB of whole cache lines instead
    // of 8-byte pieces every 32 bytes.
    const bool r_il8 = p.r_il8 != 0, has_r = p.resid != nullptr;
    const int64_t ybase = (int64_t)n * p.y_bstride + z.y_off_il8;
    const int64_t rbase = (int64_t)n * p.r_bstride + (r_il8 ? z.r_off_il8 : z.r_off);
    const int cs32 = (int)p.y_cstride, rs32 = (int)p.r_cstride;
    const int my = ty0 + wv, mx = tx0 + l31;
    const int Wx = p.y_ystride >> 1;
    const int act = p.act;
    unsigned slot_off[2];                          // byte offset of this lane's 16-byte slot (pixel 2 mx + half) in output row 2 my + py
#pragma unroll
    for (int py = 0; py < 2; py++) {
        const int oy = 2 * my + py, ox = 2 * mx + half;
        const bool inb = my < p.Hi && mx < p.Wi && oy < p.Ho && ox < p.Wo;
        slot_off[py] = inb ? (unsigned)(my * p.y_ystride + py * Wx + ox) * 16u : kBufOOB;
    }
    u32x4_t skip[2][4] = {};                       // interleaved skip tensor: [py][q] = the slot's 8 channels
    auto load_skip = [&]() {
#pragma unroll
        for (int py = 0; py < 2; py++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cs = nblk * 32 + 8 * q;
                const buf_rsrc rs = make_buf(elem_ptr(p.resid, rbase, ES), has_r & r_il8 & (cs < p.Cout));
                skip[py][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, slot_off[py], (unsigned)(cs * rs32) * ES, 0);
            }
    };

    prefetch(0);
    stage(0);
    wg_barrier();
    for (int ch = 0; ch < nchunks; ch++) {
        const bool more = ch + 1 < nchunks;
        if (more) { if (!(kDpAbl & 1)) prefetch(ch + 1); }
        else if (!(kDpAbl & 2)) load_skip();       // the skip tensor arrives under the last chunk's MFMAs
        if (!(kDpAbl & 4)) compute(ch & 1);
        if (more) stage((ch + 1) & 1);
        wg_barrier();
    }

    // ---- epilogue --------------------------------------------------------------------------------------------------------------------
    auto h2f = [](unsigned u, int hi) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> (16 * hi))); };
    // (the activation as a compile-time constant of the epilogue: on a run-time `act`, apply_act_fast is a scalar branch tree per VALUE)
    auto epilogue = [&](auto actc) __attribute__((always_inline)) {
    constexpr int ACT = decltype(actc)::value;                         // -1: whatever p.act says, per value
#pragma unroll
    for (int py = 0; py < 2; py++) {
        if (has_r && !r_il8) {                     // planar skip tensor (not what the executor uses): per-element loads at the lane's own pixels
#pragma unroll
            for (int px = 0; px < 2; px++) {
                const int oy = 2 * my + py, ox = 2 * mx + px;
                const bool inb = my < p.Hi && mx < p.Wi && oy < p.Ho && ox < p.Wo;
                const unsigned pv = inb ? ((unsigned)(my * p.y_ystride + py * Wx + ox) + (unsigned)(4 * half * rs32)) * ES : kBufOOB;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int cs = nblk * 32 + 8 * q;
                    const buf_rsrc rs = make_buf(elem_ptr(p.resid, rbase, ES), cs < p.Cout);
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        acc[2 * py + px][4 * q + e] += Io<_Float16>::load(rs, (cs + 4 * half + e < p.Cout) ? pv : kBufOOB, (unsigned)((cs + e) * rs32) * ES);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int cs = nblk * 32 + 8 * q;
            // the lane's 4 channels (4 half .. 4 half + 3 of the group) of pixel 2 mx (px = 0) and of pixel 2 mx + 1 (px = 1)
            float v[2][4];
#pragma unroll
            for (int px = 0; px < 2; px++)
#pragma unroll
                for (int e = 0; e < 4; e++) v[px][e] = acc[2 * py + px][4 * q + e];
            if (has_r && r_il8) {
                // the loaded slot is pixel 2 mx + half, channels 0-7: bring the pieces to the lanes that hold the matching accumulators
                unsigned s0 = skip[py][q][0], s1 = skip[py][q][1], s2 = skip[py][q][2], s3 = skip[py][q][3];
                // lower half-wave has pixel 2mx (ch 0-3 in s0,s1; ch 4-7 in s2,s3), upper pixel 2mx+1: swap(lower's ch4-7, upper's ch0-3)
                const auto w0 = __builtin_amdgcn_permlane32_swap(s0, s2, false, false);   // -> {a: low lanes s0 (px0 ch0-1), high lanes <- low s2? see below
                const auto w1 = __builtin_amdgcn_permlane32_swap(s1, s3, false, false);
                // after swap(a = s0, b = s2): a' = [s0 low | s2 low], b' = [s0 high | s2 high]:
                //   lanes 0-31 (half 0): a' = px0 ch0-1, b' = px1 ch0-1 (came from the upper half-wave's s0)
                //   lanes 32-63 (half 1): a' = px0 ch4-5 (the lower half-wave's s2), b' = px1 ch4-5
                const unsigned px0_lo = w0[0], px1_lo = w0[1], px0_hi = w1[0], px1_hi = w1[1];
                v[0][0] += h2f(px0_lo, 0); v[0][1] += h2f(px0_lo, 1); v[0][2] += h2f(px0_hi, 0); v[0][3] += h2f(px0_hi, 1);
                v[1][0] += h2f(px1_lo, 0); v[1][1] += h2f(px1_lo, 1); v[1][2] += h2f(px1_hi, 0); v[1][3] += h2f(px1_hi, 1);
            }
            unsigned o[2][2];
#pragma unroll
            for (int px = 0; px < 2; px++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float x0 = apply_act_fast(v[px][2 * e], ACT < 0 ? act : ACT), x1 = apply_act_fast(v[px][2 * e + 1], ACT < 0 ? act : ACT);
                    o[px][e] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x0) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)x1) << 16);
                }
            // inverse exchange: a = px0 pieces, b = px1 pieces -> lower half-wave: (a', b') = px0's (own ch 0-3, upper's ch 4-7) ...
            const auto t0 = __builtin_amdgcn_permlane32_swap(o[0][0], o[1][0], false, false);
            const auto t1 = __builtin_amdgcn_permlane32_swap(o[0][1], o[1][1], false, false);
            // t0 = {[px0 e0 of half0 | px1 e0 of half0], [px0 e0 of half1 | px1 e0 of half1]}: lanes 0-31 hold pixel 2mx: ch0-1 (t0[0]), ch4-5 (t0[1]);
            // lanes 32-63 hold pixel 2mx+1: ch0-1 (t0[0]), ch4-5 (t0[1]).  Same for t1 with channels 2-3 / 6-7.
            const u32x4_t slot = {t0[0], t1[0], t0[1], t1[1]};
            const buf_rsrc rs = make_buf(elem_ptr(p.y, ybase, ES), cs < p.Cout);
            if (!(kDpAbl & 2) || slot[0] == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b128(slot, rs, slot_off[py] + (unsigned)(cs * cs32) * ES, 0u, 0);
        }
    }
    };
    if (act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (act == 0) epilogue(std::integral_constant<int, 0>{});
    else epilogue(std::integral_constant<int, -1>{});
}


// ---- the same layer WALKING DOWN THE OUTPUT DEPTHS of its class (end of round 5) --------------------------------------------------------
// Instrumented builds of the kernel above (RT_DP_ABL, NVSmall deconv3D_2 at batch 8, 1.43 ms): without the chunk loop's loads 1.12, without
// skip-tensor loads and stores 0.97, without MFMAs 0.86, without all three 0.49 -- the parts ADD UP.  A workgroup lives for 4 or 8 chunks
// of 9 MFMAs: its first loads, the one-chunk look-ahead through staging registers, an LDS read right in front of every MFMA
// (`s_waitcnt lgkmcnt` x 9 per chunk in the ISA) and the epilogue are latency chains that three workgroups per CU do not cover.  Here:
//   * a workgroup keeps its tile and walks a SEGMENT of the class's output depths: one continuous chunk pipeline, the next slice's first
//     chunks are in flight during the epilogue, prologue and address set-up once per walk;
//   * chunks arrive by LDS-DMA (`buffer_load ... lds`) in a ring of three 16 KB buffers, two chunks ahead, no staging registers: every
//     wave issues exactly four 1 KB pieces per chunk (3 + 3 patch pieces, 9 weight pieces, one idle piece that lands in padding), so
//     `s_waitcnt vmcnt(4)` = "my pieces of this chunk have landed" (loads return in order); the epilogue waits for everything before its
//     first store (its skip tensor was requested after the pieces), so the step after it needs no wait;
//   * the three buffers are three OBJECTS and the step is instantiated per ring position: with one array and a run-time ring index the
//     compiler cannot tell the DMA's target from the buffer being read and puts `s_waitcnt vmcnt(0)` in front of the first LDS read after
//     every request (ISA of the first version) -- the look-ahead was gone;
//   * what the walk needs per slice -- gather-table entries, output / skip-tensor offsets -- comes by explicit scalar loads (sload2_*):
//     left to the compiler they are VECTOR loads behind a vmcnt(0) inside a loop that stores (and a buffer resource built from them gets a
//     waterfall loop); kept in LDS, every read of them gets a compiler-made vmcnt(0) too, because an LDS-DMA is in flight somewhere.  The
//     bias stays in registers for the same reason.
//   * BOTH classes in one launch: the even output depth 2 m reads input slice m, the odd one 2 m + 1 input slices m and m + 1; as two launches
//     the input travels from HBM three times (440 MB fetched per pair for NVSmall's deconv3D_2 against 318 MB of input + skip tensor); a walk
//     that takes slice m of the even class and then slice m of the odd class finds the second and third reading in the L2.
// Same chunk and tap order per accumulator: bit-identical to the kernel above.  p.nz = depth segments, p.dw_seg = slice indices per
// segment, p.dw_nseg = slice indices in all (the longer class), p.dw_cpc = slices of class A (p.zs), b = class B.  Skip tensor: interleaved
// or none (a planar one: the kernel above).
struct DeconvF16PWCfg {
    static constexpr int TY = 4, TX = 32, CC = 16;
    static constexpr int PR = TY + 1, PC = TX + 1, GSLOTS = PR * PC;   // 5 x 33 patch pixels per channel group
    static constexpr int GREG = 192;                                   // LDS slots per group (three 64-slot pieces; 165 used)
    static constexpr int W_SLOTS = 9 * 64;                             // nine pieces
    static constexpr int BUF = 2 * GREG + W_SLOTS + 64;                // + the idle piece: 1024 slots = 16 KB
};

// the launch's second class of output depths (the odd ones beside the even ones): its own weights, slices and gather table; zs == nullptr: none
struct DeconvWalkB {
    const float* w;
    const ZSlice* zs;
    const int* ch_off;
    int CinPad, nz;
};

__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(3) deconv_f16pw_kernel(ConvArgs p, DeconvWalkB b) {
    using Cfg = DeconvF16PWCfg;
    constexpr unsigned ES = 2;
    constexpr int PC = Cfg::PC, GREG = Cfg::GREG, BUF = Cfg::BUF;
    __shared__ __attribute__((aligned(16))) f32x4 ring0[BUF];
    __shared__ __attribute__((aligned(16))) f32x4 ring1[BUF];
    __shared__ __attribute__((aligned(16))) f32x4 ring2[BUF];

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    RT_WG_TILE(p, tile, seg, n)
    const int tx0 = (tile % p.tiles_x) * Cfg::TX;
    const int ty0 = (tile / p.tiles_x) * Cfg::TY;
    const int nblk = blockIdx.y;
    const char* __restrict__ xb = elem_ptr(p.x, (int64_t)n * p.x_bstride, ES);
    // class A = p (p.dw_cpc slices), class B = b (b.nz slices, or none): the walk takes slice m of A, then slice m of B, for m in its segment
    const int nchA = p.CinPad / Cfg::CC, nchB = b.zs ? b.CinPad / Cfg::CC : 0;
    const int nzA = p.dw_cpc, nzB = b.zs ? b.nz : 0;
    const int z0 = seg * p.dw_seg, z1 = z0 + p.dw_seg < p.dw_nseg ? z0 + p.dw_seg : p.dw_nseg;
    auto clampn = [&](int nz) { const int v = nz - z0; return v < 0 ? 0 : (v > z1 - z0 ? z1 - z0 : v); };
    const int nq = clampn(nzA) * nchA + clampn(nzB) * nchB;
    const bool has_r = p.resid != nullptr;

    // ---- LDS-DMA duties.  Patch piece k of a group covers patch slots 64 k .. 64 k + 63 (slot = row * 33 + column; beyond 165 and outside
    // the image: out of range, zeros)
    unsigned pvoff[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int s = 64 * k + lane;
        const int pr = s / PC, pc = s - pr * PC;
        const int iy = ty0 + pr, ix = tx0 + pc;
        pvoff[k] = (s < Cfg::GSLOTS && iy < p.Hi && ix < p.Wi) ? (unsigned)(iy * p.x_pitch + ix) * 16u : kBufOOB;
    }
    const buf_rsrc rs_wA = make_buf(reinterpret_cast<const char*>(p.w) + ((int64_t)nblk * nchA) * Cfg::W_SLOTS * 16);
    const buf_rsrc rs_wB = make_buf(reinterpret_cast<const char*>(b.w) + ((int64_t)nblk * nchB) * Cfg::W_SLOTS * 16, b.zs != nullptr);
    const unsigned wlane = (unsigned)lane * 16u;
    int iq_m = z0, iq_cls = z0 < nzA ? 0 : 1, iq_c = 0;                // the next chunk to request: slice index, class, chunk of its slice
    const int* tabq;                                                   // ... and its slice's row of the gather table
    auto load_row = [&]() __attribute__((always_inline)) {
        int row, unused;
        const bool use_b = iq_cls != 0 && nzB > 0;                     // (past the walk's end the state names a slice that does not exist: any valid row)
        const int nzc = use_b ? nzB : nzA;
        const int zm = iq_m < nzc ? iq_m : nzc - 1;
        const int zo = zm * (int)sizeof(ZSlice) + (int)offsetof(ZSlice, ch_row);
        sload2_i32(use_b ? b.zs : p.zs, zo, zo, row, unused);
        tabq = (use_b ? b.ch_off : p.ch_off) + (int64_t)row * (use_b ? b.CinPad : p.CinPad);
    };
    load_row();
    auto dma = [&](buf_rsrc rs, f32x4* dst, unsigned voff, unsigned soff) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, RT_LDS_PTR(dst), 16, voff, soff, 0, 0);
    };
    auto issue = [&](f32x4* const g0) __attribute__((always_inline)) {
        // element offset of the group's slot plane; -1 = zeros (depth tap out of range)
        int o0, o1;
        sload2_i32(tabq, iq_c * (Cfg::CC * 4), iq_c * (Cfg::CC * 4) + 32, o0, o1);
        const buf_rsrc r0 = make_buf(xb, o0 >= 0), r1 = make_buf(xb, o1 >= 0);
        const unsigned s0 = (unsigned)o0 * ES, s1 = (unsigned)o1 * ES;
        f32x4* const g1 = g0 + GREG;
        f32x4* const sw = g0 + 2 * GREG;
        const buf_rsrc rs_w = iq_cls ? rs_wB : rs_wA;
        const unsigned wso = (unsigned)iq_c * (unsigned)(Cfg::W_SLOTS * 16);
        if (wv == 0) {
            dma(r0, g0, pvoff[0], s0); dma(r0, g0 + 64, pvoff[1], s0); dma(r0, g0 + 128, pvoff[2], s0);
            dma(r1, g1, pvoff[0], s1);
        } else if (wv == 1) {
            dma(r1, g1 + 64, pvoff[1], s1); dma(r1, g1 + 128, pvoff[2], s1);
            dma(rs_w, sw, wlane, wso); dma(rs_w, sw + 64, wlane + 1024u, wso);
        } else if (wv == 2) {
#pragma unroll
            for (int j = 2; j < 6; j++) dma(rs_w, sw + 64 * j, wlane + 1024u * j, wso);
        } else {
#pragma unroll
            for (int j = 6; j < 9; j++) dma(rs_w, sw + 64 * j, wlane + 1024u * j, wso);
            dma(rs_w, sw + 64 * 9, kBufOOB, 0u);                      // the idle piece (zeros into the buffer's padding): four operations per wave and chunk
        }
        // next chunk; at a slice's end slice m of class B follows slice m of class A, then slice m + 1.  Branch-free scalar arithmetic: as
        // nested control flow the compiler carried the class through a VECTOR register -- one that a DMA in flight still named as its
        // address -- and put a vmcnt(0) in front of the write (ISA of the first two-class version)
        // (0 / 1 from sign bits, not from comparisons: a boolean widened to an integer goes through v_cndmask + v_readfirstlane)
        iq_c++;
        const int nch = nchA + iq_cls * (nchB - nchA);
        const int endc = (int)((unsigned)(nch - 1 - iq_c) >> 31);      // iq_c >= nch
        const int to_b = endc & (iq_cls ^ 1) & (int)((unsigned)(iq_m - nzB) >> 31);      // class A just ended and slice m of class B exists
        const int adv = endc & (to_b ^ 1);
        iq_c -= endc * iq_c;
        iq_m += adv;
        const int ge_a = (int)((unsigned)(nzA - 1 - iq_m) >> 31);      // iq_m >= nzA: only class B has this slice
        iq_cls = to_b + (to_b ^ 1) * (adv * ge_a + (adv ^ 1) * iq_cls);
        if (endc) load_row();                                          // (past the segment's end: a row that is never used)
    };

    // ---- output addressing: as in deconv_f16p_kernel (one v_permlane32_swap per dword turns the two half-waves' 8-byte pieces into whole
    // 16-byte slots of neighbouring pixels)
    const int cs32 = (int)p.y_cstride, rs32 = (int)p.r_cstride;
    const int my = ty0 + wv, mx = tx0 + l31;
    const int Wx = p.y_ystride >> 1;
    const int act = p.act;
    unsigned slot_off[2];
#pragma unroll
    for (int py = 0; py < 2; py++) {
        const int oy = 2 * my + py, ox = 2 * mx + half;
        const bool inb = my < p.Hi && mx < p.Wi && oy < p.Ho && ox < p.Wo;
        slot_off[py] = inb ? (unsigned)(my * p.y_ystride + py * Wx + ox) * 16u : kBufOOB;
    }
    auto h2f = [](unsigned u, int hi) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> (16 * hi))); };

    f32x16 acc[4];                                                     // phase 2 * py + px
    f32x4 bias4[4];
    {
        const float* bsrc = p.bias + nblk * 32 + 4 * half;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) bias4[qq] = *reinterpret_cast<const f32x4*>(bsrc + 8 * qq);
    }
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qq = 0; qq < 4; qq++)
#pragma unroll
            for (int f = 0; f < 4; f++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[f][4 * qq + e] = bias4[qq][e];
    };
    init_acc();
    issue(ring0);
    if (nq > 1) issue(ring1);

    int q = 0, ch = 0, cm = z0, ccls = z0 < nzA ? 0 : 1;               // chunk of the walk, chunk of its slice, the slice being computed: index, class
    bool fresh = false;                                                // everything requested before this point has landed (set by an epilogue)
    // one chunk of the pipeline at ring position R: wait for my pieces, barrier, request the chunk two ahead, MFMAs; the last chunk of a slice
    // requests the skip tensor before its MFMAs and ends in the slice's epilogue
    auto step = [&](auto rc) __attribute__((always_inline)) {
        constexpr int R = decltype(rc)::value;
        f32x4* const rd = R == 0 ? ring0 : (R == 1 ? ring1 : ring2);
        f32x4* const wr = R == 0 ? ring2 : (R == 1 ? ring0 : ring1);   // ring position (R + 2) % 3: the buffer chunk q - 1 was read from
        if (!fresh) {
            if (q + 1 < nq) wait_vmem_but<4>(); else wait_vmem();
        }
        fresh = false;
        lds_barrier();
        if (q + 2 < nq) issue(wr);
        const bool last = ch == (ccls ? nchB : nchA) - 1;              // uniform
        u32x4_t skip[2][4];                                            // interleaved skip tensor: [py][q] = the slot's 8 channels; lands under the MFMAs
        int64_t ybase = 0;
        if (last) {
            long long yo, ro;
            const int zb = cm * (int)sizeof(ZSlice);
            sload2_i64(ccls ? b.zs : p.zs, zb + (int)offsetof(ZSlice, y_off_il8), zb + (int)offsetof(ZSlice, r_off_il8), yo, ro);
            ybase = (int64_t)n * p.y_bstride + yo;
            const int64_t rbase = (int64_t)n * p.r_bstride + ro;
#pragma unroll
            for (int py = 0; py < 2; py++)
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    const int cs = nblk * 32 + 8 * qq;
                    const buf_rsrc rs = make_buf(elem_ptr(p.resid, rbase, ES), has_r & (cs < p.Cout));
                    skip[py][qq] = __builtin_amdgcn_raw_buffer_load_b128(rs, slot_off[py], (unsigned)(cs * rs32) * ES, 0);
                }
        }
        {
            const f32x4* sIn = rd;
            const f32x4* sW = rd + 2 * GREG;
            // operands one tap ahead of their MFMAs (left to itself the scheduler puts every LDS read right in front of its MFMA -- an
            // `s_waitcnt lgkmcnt(0)` per matrix instruction -- to save registers): the 4 patch operands and the first weight operand, then
            // one read per MFMA
            f16x8_t b[2][2], a[9];
#pragma unroll
            for (int dy = 0; dy < 2; dy++)
#pragma unroll
                for (int dx = 0; dx < 2; dx++) b[dy][dx] = __builtin_bit_cast(f16x8_t, sIn[half * GREG + (wv + dy) * PC + l31 + dx]);
#pragma unroll
            for (int t = 0; t < 9; t++) a[t] = __builtin_bit_cast(f16x8_t, sW[t * 64 + lane]);
#pragma unroll
            for (int ry = 0; ry < 3; ry++)
#pragma unroll
                for (int rx = 0; rx < 3; rx++) {
                    // kernel tap r: 1 -> even output from input m; 2 -> odd output from m; 0 -> odd output from m + 1
                    const int py = ry == 1 ? 0 : 1, dy = ry == 0 ? 1 : 0, px = rx == 1 ? 0 : 1, dx = rx == 0 ? 1 : 0;
                    acc[2 * py + px] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ry * 3 + rx], b[dy][dx], acc[2 * py + px], 0, 0, 0);
                }
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);          // DS reads: 4 patch operands + 2 weight operands
#pragma unroll
            for (int t = 0; t < 7; t++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // the weight operand two taps ahead
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        q++;
        if (!last) { ch++; return; }
        ch = 0;
        if (ccls == 0 && cm < nzB) ccls = 1;
        else { cm++; ccls = cm < nzA ? 0 : 1; }
        // ---- epilogue of the slice.  Everything requested so far (the skip tensor, the pieces of the next two chunks) is waited for before
        // the first store: stores count on vmcnt too and may complete out of order with loads -- no counted wait can tell them apart.
        wait_vmem();
        fresh = true;
        // (the activation as a compile-time constant of the epilogue: `apply_act_fast(x, act)` on a run-time `act` is a scalar branch tree
        //  PER VALUE -- 64 of them per lane and slice, 400 instructions per 8 outputs in the ISA of the first version)
        auto epilogue = [&](auto actc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;                     // -1: whatever p.act says, per value
#pragma unroll
        for (int py = 0; py < 2; py++) {
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const int cs = nblk * 32 + 8 * qq;
                // the lane's 4 channels (4 half .. 4 half + 3 of the group) of pixel 2 mx (px = 0) and of pixel 2 mx + 1 (px = 1)
                float v[2][4];
#pragma unroll
                for (int px = 0; px < 2; px++)
#pragma unroll
                    for (int e = 0; e < 4; e++) v[px][e] = acc[2 * py + px][4 * qq + e];
                if (has_r) {
                    // the loaded slot is pixel 2 mx + half, channels 0-7: bring the pieces to the lanes that hold the matching accumulators
                    unsigned s0 = skip[py][qq][0], s1 = skip[py][qq][1], s2 = skip[py][qq][2], s3 = skip[py][qq][3];
                    const auto w0 = __builtin_amdgcn_permlane32_swap(s0, s2, false, false);
                    const auto w1 = __builtin_amdgcn_permlane32_swap(s1, s3, false, false);
                    const unsigned px0_lo = w0[0], px1_lo = w0[1], px0_hi = w1[0], px1_hi = w1[1];
                    v[0][0] += h2f(px0_lo, 0); v[0][1] += h2f(px0_lo, 1); v[0][2] += h2f(px0_hi, 0); v[0][3] += h2f(px0_hi, 1);
                    v[1][0] += h2f(px1_lo, 0); v[1][1] += h2f(px1_lo, 1); v[1][2] += h2f(px1_hi, 0); v[1][3] += h2f(px1_hi, 1);
                }
                unsigned o[2][2];
#pragma unroll
                for (int px = 0; px < 2; px++)
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const float x0 = apply_act_fast(v[px][2 * e], ACT < 0 ? act : ACT), x1 = apply_act_fast(v[px][2 * e + 1], ACT < 0 ? act : ACT);
                        o[px][e] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x0) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)x1) << 16);
                    }
                // inverse exchange: lanes 0-31 hold pixel 2 mx (ch 0-1, 4-5 in t0; 2-3, 6-7 in t1), lanes 32-63 pixel 2 mx + 1
                const auto t0 = __builtin_amdgcn_permlane32_swap(o[0][0], o[1][0], false, false);
                const auto t1 = __builtin_amdgcn_permlane32_swap(o[0][1], o[1][1], false, false);
                const u32x4_t slot = {t0[0], t1[0], t0[1], t1[1]};
                const buf_rsrc rs = make_buf(elem_ptr(p.y, ybase, ES), cs < p.Cout);
                __builtin_amdgcn_raw_buffer_store_b128(slot, rs, slot_off[py] + (unsigned)(cs * cs32) * ES, 0u, 0);
            }
        }
        };
        if (act == 1) epilogue(std::integral_constant<int, 1>{});      // ELU: every layer of the reference's networks that reaches this kernel
        else if (act == 0) epilogue(std::integral_constant<int, 0>{});
        else epilogue(std::integral_constant<int, -1>{});
        init_acc();
    };
    for (;;) {
        step(std::integral_constant<int, 0>{});
        if (q == nq) break;
        step(std::integral_constant<int, 1>{});
        if (q == nq) break;
        step(std::integral_constant<int, 2>{});
        if (q == nq) break;
    }
}

}  // namespace rt
