// conv_f16rbd_kernel: the tower block in half2 mode -- fp16 tensors, fp16 weights, fp32 accumulation -- in ONE launch (round 6).
//
//       y = fp16( ELU( conv3x3( fp16( ELU( conv3x3(x) + b1 ) ) ) + b2 + x ) )          32 -> 32 -> 32 channels, stride 1
//
// (reference: resblockN_conv1 -> ELU -> resblockN_conv2 -> add -> ELU, resnet18_2D_513x257_net.cpp:66-575, under IBuilder::setHalf2Mode,
// sample_app/main.cpp:228-266.)  Layer by layer (conv_f16mma_kernel twice) the block moves x, t, t, x, y through HBM and each layer is bound
// by exactly that (conv_f16.hip.h); fused, the intermediate never leaves the CU: two tensor passes instead of five.  Structure of
// conv_s3rbd_kernel (conv_rbd.hip.h) with everything the fp32 form needs the split for taken out:
//   * the channel-interleaved fp16 tensors (C/8, H, pitch, 8) ARE the B operands: the x ring is filled by `buffer_load_dwordx4 ... lds`,
//     9 one-KB pieces per 4-row step (36 columns per ring row so that a batch is a whole number of pieces; 34 are fetched);
//     16-byte records of consecutive pixels are bank-conflict free as they lie.  A step is ~1.5 k cycles here -- about one HBM latency --
//     so batches are requested TWO steps ahead, by the conv1 waves only: they store nothing, so `s_waitcnt vmcnt(pieces just issued)`
//     is exactly "the older batch has landed" (loads return in order among themselves), and the conv2 waves cross the barriers with
//     their stores in flight;
//   * ONE MFMA per tap and 16-channel chunk; the fp16 weights of a wave's convolution (18 KB) stay in 72 VGPRs, nothing else is staged;
//   * t is rounded to fp16 into its LDS ring exactly as the layer-by-layer path rounds it into HBM, the skip connection enters the
//     accumulator before the first MFMA (bias + x, as conv_f16mma_kernel initialises it), and the MFMAs run chunk-major, taps inside,
//     like that kernel's: results are BIT-IDENTICAL to the two launches (tests/test_f16_storage.py).
// Strip of 30 columns x segment of rows, 4 rows per step, waves 0-3 conv1, waves 4-7 conv2, one barrier per step (conv_rbd.hip.h).
#pragma once
#include "conv_rbd.hip.h"

namespace rt {

struct F16RBDCfg {
    static constexpr int NW = 8, NT = 512;
    static constexpr int SW = 30, XCOL = 36, TCOL = 32, STEP = 4;
    static constexpr int RX = 20, RT = 10;                      // x ring: window 6 + skip rows 4 + TWO landing batches
    static constexpr int GXB = XCOL * 16, XROWB = 4 * GXB;      // 576, 2304
    static constexpr int GTB = TCOL * 16, TROWB = 4 * GTB;      // 512, 2048
    static constexpr int BSLOTS = STEP * XROWB / 16;            // 576 = 9 pieces of 64
    static constexpr int NPIECE = BSLOTS / 64;
    static constexpr int PPW = (NPIECE + 3) / 4;                // the four conv1 waves move the pieces: w, w + 4, and wave 0 the ninth
    static constexpr int W_SLOTS = 18 * 64;
    static_assert(BSLOTS % 64 == 0, "a batch is a whole number of DMA pieces");
};

// 128 registers and 71 KB of LDS: TWO workgroups per CU, four waves per SIMD -- a wave's 18 MFMAs of a row are one dependent chain (the
// order of summation of the layer-by-layer kernel), and what hides that chain's latency is other waves.
// element J of a packed fp16 pair + an fp32 value
template <int J>
__device__ static __forceinline__ float rbh_add16(unsigned h2, float b) {
#ifdef HIPEMU
    return b + (float)__builtin_bit_cast(_Float16, (unsigned short)(h2 >> (16 * J)));
#else
    float r;
    if (J == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(b));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(b));
    return r;
#endif
}

__global__ void __launch_bounds__(512) RT_WAVES_PER_EU(4) conv_f16rbd_kernel(RBArgs a) {
    using Cfg = F16RBDCfg;
    const ConvArgs& p = a.c;
    constexpr int RX = Cfg::RX, RT = Cfg::RT, GXB = Cfg::GXB, XROWB = Cfg::XROWB, GTB = Cfg::GTB, TROWB = Cfg::TROWB, PPW = Cfg::PPW;
    constexpr unsigned ES = 2;

    __shared__ __attribute__((aligned(1024))) char sX[RX * XROWB];
    __shared__ __attribute__((aligned(1024))) char sT[(RT + 2) * TROWB];       // rows 10, 11 mirror rows 0, 1: conv2's 3-row window never wraps
    // prologue only: the two weight slabs on their way to registers pass through ring rows nothing else uses yet -- conv1's through the
    // t ring, conv2's through the last 8 rows of the x ring (batches -1, 0, 1 land in its rows 0 .. 11)
    static_assert(Cfg::W_SLOTS * 16 <= (RT + 2) * TROWB && Cfg::W_SLOTS * 16 <= 8 * XROWB, "weight staging fits the idle ring rows");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is1 = wv < 4;
    const int wr = wv & 3;

    int tile = blockIdx.x;
    if (p.xcd_order) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int H = p.Hi, W = p.Wi;
    const int c0 = (tile % p.tiles_x) * Cfg::SW;
    const int y0 = (tile / p.tiles_x) * a.seg;
    const int y1 = y0 + a.seg < H ? y0 + a.seg : H;
    const int t0 = y0 - 1;
    const int n = blockIdx.z;
    const int nstep = (y1 - t0 + 4) / 4 + 1;
    const int last1 = (y1 - t0) / 4;
    const int row_hi = (y1 + 1 < H - 1 ? y1 + 1 : H - 1);

    // ---- DMA duties (conv_rbd.hip.h): batch b = x rows t0 + 1 + 4 b .. + 3 -> ring slots (4 b + 4) % 20 .. + 3, ring order [row][group][pixel]
    const buf_rsrc rs_x = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, ES));
    const unsigned gsb_x = (unsigned)p.x_cstride * 16u, rowb_x = (unsigned)p.x_pitch * 16u;
    unsigned xvo[PPW];
    int xnb[PPW];
#pragma unroll
    for (int j = 0; j < PPW; j++) {
        const int k = (wr + 4 * j) * 64 + lane;
        const int r = k / (XROWB / 16), rem = k - r * (XROWB / 16);
        const int g = rem / Cfg::XCOL, px = rem - g * Cfg::XCOL;
        const int ix = c0 - 2 + px;
        const bool valid = is1 && wr + 4 * j < Cfg::NPIECE && px < 34 && ix >= 0 && ix < W;
        xvo[j] = valid ? (unsigned)g * gsb_x + (unsigned)(t0 + 1 + r) * rowb_x + (unsigned)ix * 16u : kBufOOB;
        const int d = row_hi - (t0 + 1 + r);
        xnb[j] = valid && d >= 0 ? (d >> 2) + 1 : 0;
    }
    auto issue_batch = [&](int b) __attribute__((always_inline)) {
        const unsigned so = (unsigned)(4 * b) * rowb_x;
        char* dst = sX + ((4 * b + 4) % RX) * XROWB;
#pragma unroll
        for (int j = 0; j < PPW; j++)
            if (wr + 4 * j < Cfg::NPIECE) rbd_dma16(rs_x, dst + (wr + 4 * j) * 1024, b < xnb[j] ? xvo[j] : kBufOOB, so);
    };

    // ---- prologue: batches -1 (rows t0 - 1, t0 if they exist), 0 and 1, this wave's convolution's weights, its bias
    if (is1) {
#pragma unroll
        for (int j = 0; j < PPW; j++) {
            if (wr + 4 * j >= Cfg::NPIECE) continue;
            const int r = ((wr + 4 * j) * 64 + lane) / (XROWB / 16);
            const bool ok = xvo[j] != kBufOOB && r >= 2 && t0 - 3 + r >= 0;
            rbd_dma16(rs_x, sX + (wr + 4 * j) * 1024, ok ? xvo[j] - 4u * rowb_x : kBufOOB, 0u);
        }
        issue_batch(0);
        issue_batch(1);
    }
    {
        // slab of a convolution (rt_capi.hip): [chunk * 9 + tap][lane] 16-byte slots, output channels in conv_s3rbd_kernel's row order
        const buf_rsrc rs_w = make_buf(is1 ? a.w1 : p.w);
        for (int t = wr; t < 18; t += 4)
            rbd_dma16(rs_w, (is1 ? sT : sX + 12 * XROWB) + t * 1024, (unsigned)lane * 16u, (unsigned)t * 1024u);
    }
    f32x16 biasv;
    {
        const float* bsrc = is1 ? a.bias1 : p.bias;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bsrc + 16 * (q >> 1) + 8 * kg + 4 * (q & 1));
#pragma unroll
            for (int e = 0; e < 4; e++) biasv[4 * q + e] = bv[e];
        }
    }
    wait_vmem();
#ifndef HIPEMU
    asm volatile("" : "+v"(biasv));                     // see conv_rbd.hip.h: the compiler's wait for the bias loads belongs here
#endif
    __syncthreads();
    f16x8 wh[18];
    {
        const f32x4* whs = reinterpret_cast<const f32x4*>(is1 ? sT : sX + 12 * XROWB) + lane;
#pragma unroll
        for (int t = 0; t < 18; t++) wh[t] = __builtin_bit_cast(f16x8, whs[t * 64]);
    }

    // B operand of (window row r, column shift s, chunk c): ring row + c * 2 groups + this lane's (k-group, pixel l31 + s)
    const int bo = kg * (is1 ? GXB : GTB) + l31 * 16;
    // taps run chunk-major like conv_f16mma_kernel's (t = chunk * 9 + tap): same order of summation.  Reads run PF taps ahead.
    constexpr int PF = 1;
    auto contract = [&](auto ring1, int first, f32x16 init) __attribute__((always_inline)) -> f32x16 {
        constexpr bool R1 = decltype(ring1)::value;
        constexpr int NR = R1 ? RX : RT, ROWB = R1 ? XROWB : TROWB, GB = R1 ? GXB : GTB;
        const char* ring = R1 ? sX : sT;
        int so[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int slot = first + r;
            so[r] = R1 ? (slot >= NR ? slot - NR : slot) * ROWB : first * ROWB + r * ROWB;       // (the t ring's window never wraps: mirror rows)
        }
        auto b_at = [&](int t) { return *reinterpret_cast<const f16x8*>(ring + so[(t % 9) / 3] + (t / 9) * (2 * GB) + ((t % 9) % 3) * 16 + bo); };
        f16x8 b[PF + 1];
#pragma unroll
        for (int i = 0; i < PF; i++) b[i] = b_at(i);
        f32x16 acc = init;
#pragma unroll
        for (int t = 0; t < 18; t++) {
            if (t + PF < 18) b[PF] = b_at(t + PF);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], b[0], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PF; i++) b[i] = b[i + 1];
        }
        return acc;
    };
    // 8 accumulator values (one 8-channel group of this lane's pixel) -> ELU -> fp16, 16 bytes
    auto elu_pack = [&](const f32x16& acc, int g) __attribute__((always_inline)) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = pack_f16(apply_act_fast(acc[8 * g + 2 * e], 1), apply_act_fast(acc[8 * g + 2 * e + 1], 1));
        return o;
    };
    __syncthreads();                                    // every wave has its weights: the ring rows they passed through are free

    if (is1) {
        // ================= conv1 waves: x ring -> t ring =================
        const int gx = c0 - 1 + l31;
        const bool col_in = gx >= 0 && gx < W;
        const bool strip_edge = c0 == 0 || c0 + 31 > W;
        const int tl = kg * GTB + l31 * 16;
        for (int s = 0; s < nstep; s++) {
            const bool more = s + 2 <= last1;           // conv1 of step s + 2 needs batch s + 2
            if (more) issue_batch(s + 2);
            if (s <= last1) {
                const int row = t0 + 4 * s + wr;
                const f32x16 acc = contract(std::true_type(), (4 * s + wr + 2) % RX, biasv);
                const bool row_in = row >= 0 && row < H;
                u32x4_t o[2];
#pragma unroll
                for (int g = 0; g < 2; g++) o[g] = elu_pack(acc, g);
                if (!row_in || strip_edge) {            // rows and columns outside the image are conv2's zero padding
                    const unsigned m = (row_in && col_in) ? 0xffffffffu : 0u;
#pragma unroll
                    for (int g = 0; g < 2; g++)
#pragma unroll
                        for (int e = 0; e < 4; e++) o[g][e] &= m;
                }
                const int slot = (4 * s + wr) % RT;
                char* trow = sT + slot * TROWB + tl;
#pragma unroll
                for (int g = 0; g < 2; g++) *reinterpret_cast<u32x4_t*>(trow + g * (2 * GTB)) = o[g];
                if (slot < 2) {
#pragma unroll
                    for (int g = 0; g < 2; g++) *reinterpret_cast<u32x4_t*>(trow + RT * TROWB + g * (2 * GTB)) = o[g];
                }
            }
            // batch s + 1 has landed when at most the pieces of batch s + 2 are outstanding
            if (!more) wait_vmem();
            else if (wr == 0) wait_vmem_but<3>();
            else wait_vmem_but<2>();
            lds_barrier();
        }
    } else {
        // ================= conv2 waves: t ring -> y =================
        const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride + p.y_off, ES));
        const unsigned cs_y = (unsigned)p.y_cstride;
        f32x16 acc;
        int prow = -1;
        const int ox = c0 + l31;
        const bool col_ok = l31 < Cfg::SW && ox < W;
        const int xl = kg * GXB + (l31 + 2) * 16;
        auto epilogue2 = [&]() __attribute__((always_inline)) {
            const unsigned vo = (col_ok && prow >= 0) ? (unsigned)(prow * p.y_ystride + ox) * 16u + (unsigned)kg * cs_y * 16u : kBufOOB;
#pragma unroll
            for (int g = 0; g < 2; g++) buf_store4(__builtin_bit_cast(f32x4, elu_pack(acc, g)), rs_y, vo, (unsigned)(2 * g) * cs_y * 16u);
        };
        for (int s = 0; s < nstep; s++) {
            if (s >= 2) epilogue2();                    // rows of step s - 1
            if (s >= 1) {
                const int row = t0 + 4 * s - 5 + wr;
                // accumulator = bias + skip connection (fp32 additions, as conv_f16mma_kernel starts), x out of the ring: slot (row - t0 + 3) % 20
                const char* xs = sX + ((4 * s - 2 + wr) % RX) * XROWB + xl;
                f32x16 init;
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const u32x4_t sk = *reinterpret_cast<const u32x4_t*>(xs + g * (2 * GXB));
#pragma unroll
                    for (int e = 0; e < 4; e++) {       // bias + x: one v_fma_mix_f32 per value (x * 1 + bias, rounded once like the addition)
                        init[8 * g + 2 * e] = rbh_add16<0>(sk[e], biasv[8 * g + 2 * e]);
                        init[8 * g + 2 * e + 1] = rbh_add16<1>(sk[e], biasv[8 * g + 2 * e + 1]);
                    }
                }
                acc = contract(std::false_type(), (4 * s + wr + 4) % RT, init);
                prow = (row >= y0 && row < y1) ? row : -1;
            }
            lds_barrier();
        }
        epilogue2();
    }
}

}  // namespace rt
