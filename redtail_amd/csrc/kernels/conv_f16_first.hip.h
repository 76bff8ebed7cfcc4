// First layer of the 2-D network in half2 mode: 5x5 stride-2 convolution of the fp32 image binding (3 channels) to an
// fp16 tensor (reference resnet18_2D_513x257_net.cpp:48-63: conv1 5x5 s2 3 -> 32 + ELU; TensorRT half2 mode converts
// the input to fp16 and runs the layer on fp16 operands as well).  With 3 input channels a channel-chunked contraction
// is 75 % padding (conv_mfma_f32_kernel pads 3 channels to 4 and issues 50 fp32 MFMAs per tile); here a ROW of the
// window is the contraction index:
//     k = 3*s + c   (s = 0..4 window column, c = 0..2 channel; 15 values, the 16th has zero weights)
// which is contiguous in an LDS patch stored as [row][col][channel] halfs, so the B operand of a lane (output pixel x,
// k-half h) is the 16 bytes at half offset 6*x + 8*h of patch row 2*y + r, and ONE v_mfma_f32_32x32x16_f16 per
// window row does the work: 5 per tile.  The 5 A operands (weights [r][h][co][8]) sit in registers for the whole tile,
// there is no weight staging and a single barrier.  Operands: image values and weights rounded to fp16, fp32 accumulate.
// Output: planar fp16 or channel-interleaved (YIL8, see conv_f16.hip.h).  Bound by data movement: 11 x 67 x 3 fp32 in,
// 4 x 32 x 32 fp16 out per tile.
#pragma once
#include <type_traits>
#include "common.hip.h"
#include "conv_mfma.hip.h"
#include "conv_f16.hip.h"

namespace rt {

struct ConvF16FirstCfg {
    static constexpr int KH = 5, KW = 5, S = 2, TY = 4, TX = 32, CMAX = 3;
    static constexpr int PR = (TY - 1) * S + KH, PC = (TX - 1) * S + KW;      // 11 x 67 input pixels per channel
    static constexpr int NPAIR = (PC + 1) / 2;                                // column pairs per row: 34 (68 columns)
    static constexpr int RS = 208;                                            // LDS row stride in halfs >= 68*3 = 204, 16-byte multiple
    static constexpr int NTASK = CMAX * PR * NPAIR, NK = (NTASK + 255) / 256; // (channel, row, pair) gathers per lane
    static constexpr int W_SLOTS = KH * 2 * 32;                               // 16-byte weight slots per 32-channel block
};

template <bool YIL8>
__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(8) conv_f16_first_kernel(ConvArgs p, int cin) {
    using Cfg = ConvF16FirstCfg;
    constexpr int KH = Cfg::KH, S = Cfg::S, TY = Cfg::TY, TX = Cfg::TX, PR = Cfg::PR, NPAIR = Cfg::NPAIR, RS = Cfg::RS;
    constexpr unsigned ESY = 2;

    __shared__ __attribute__((aligned(16))) _Float16 sIn[PR * RS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
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

    // ---- weights: the 5 A operands of this lane (co = l31, k-half = half), straight from global memory ---------------
    f32x4 wa[KH];
    {
        const buf_rsrc rs_w = make_buf(reinterpret_cast<const char*>(p.w) + (int64_t)nblk * Cfg::W_SLOTS * 16);
#pragma unroll
        for (int r = 0; r < KH; r++) wa[r] = buf_load4(rs_w, (unsigned)((r * 2 + half) * 32 + l31) * 16u, 0);
    }

    // ---- gather: fp32 image -> fp16 patch [row][col][channel]; a lane takes a column pair of one channel ------------
    const buf_rsrc rs_x = make_buf((p.x2 && n >= p.x2_from) ? p.x2 + (int64_t)(n - p.x2_from) * p.x_bstride : p.x + (int64_t)n * p.x_bstride);
    const int ix0 = tx0 * S - p.pad_x, iy0 = ty0 * S - p.pad_y;
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
            sIn[pr * RS + pc * Cfg::CMAX + c] = (_Float16)v0;
            sIn[pr * RS + (pc + 1) * Cfg::CMAX + c] = (_Float16)v1;
        }
    }
    __syncthreads();

    // ---- 5 MFMAs: window row r, B operand = 8 consecutive (column, channel) halfs of patch row 2*wv + r ------------------
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
    const unsigned* __restrict__ sIn32 = reinterpret_cast<const unsigned*>(sIn);      // 6*x + 8*h halfs is an even offset
#pragma unroll
    for (int r = 0; r < KH; r++) {
        const int base = ((wv * S + r) * RS + 6 * l31 + 8 * half) >> 1;
        u32x4_t b4;
#pragma unroll
        for (int e = 0; e < 4; e++) b4[e] = sIn32[base + e];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wa[r]), __builtin_bit_cast(f16x8_t, b4), acc, 0, 0, 0);
    }

    // ---- epilogue: activation, fp16 stores (planar: 16 x 2 bytes; interleaved: the lane's 4 channels as 8 bytes) -----
    const int oy = ty0 + wv, ox = tx0 + l31;
    const bool inb = oy < p.Ho && ox < p.Wo;
    const int64_t ybase = (int64_t)n * p.y_bstride + p.y_off;
    const int cs32 = (int)p.y_cstride;
    const bool tail8 = (p.Cout & 7) != 0;
    const unsigned yvoff = !inb ? kBufOOB : (YIL8 ? (unsigned)((oy * p.y_ystride + ox) * 8 + 4 * half) * ESY
                                                  : (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * half * cs32) * ESY);
    const int act = p.act;
    auto epilogue = [&](auto ACT) {
        if constexpr (YIL8) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cs = nblk * 32 + 8 * q;
                const buf_rsrc rs = make_buf(elem_ptr(p.y, ybase, ESY), cs < p.Cout);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = apply_act_fast(acc[4 * q + e], decltype(ACT)::value);
                u32x2_t o;
#pragma unroll
                for (int e = 0; e < 2; e++)
                    o[e] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * e]) |
                           ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * e + 1]) << 16);
                __builtin_amdgcn_raw_buffer_store_b64(o, rs, yvoff, (unsigned)(cs * cs32) * ESY, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int cs = nblk * 32 + (r & 3) + 8 * (r >> 2);
                const buf_rsrc rs = make_buf(elem_ptr(p.y, ybase, ESY), cs < p.Cout);
                const unsigned vo = (tail8 && cs + 4 * half >= p.Cout) ? kBufOOB : yvoff;
                Io<_Float16>::store(apply_act_fast(acc[r], decltype(ACT)::value), rs, vo, (unsigned)(cs * cs32) * ESY);
            }
        }
    };
    if (act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (act == 2) epilogue(std::integral_constant<int, 2>{});
    else epilogue(std::integral_constant<int, 0>{});
}

}  // namespace rt
