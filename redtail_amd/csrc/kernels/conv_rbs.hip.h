// conv_s3rbs_kernel: a residual block of the feature towers in one launch, STREAMING down the image.
//
//       y = ELU( conv3x3( ELU( conv3x3(x) + b1 ) ) + b2 + x )          32 -> 32 -> 32 channels, stride 1, interleaved tensors
//
// (reference resnet18_2D_513x257_net.cpp:66-575: resblockN_conv1 -> ELU -> resblockN_conv2 -> add -> ELU, 8 blocks per
// side = 32 of the network's 48 launches.)  Layer by layer the block moves x, t, t, x, y through HBM -- five tensor
// passes, and with four contexts in flight the towers run at ~3.8 TB/s of such traffic: memory-bound.
// Fused, the intermediate t never leaves the CU: two passes (+ halo).  conv_s3rb_kernel (conv_split.hip.h)
// does that per 4 x 32 tile and pays for it with a recomputed halo in BOTH directions, weights staged per tile and phases
// that nothing overlaps; this kernel removes those three costs:
//
//   * a workgroup owns a STRIP of 30 output columns and a SEGMENT of 16 output rows and walks down it 4 rows per step.
//     x rows and t rows live in two LDS ring buffers (10 rows each): every x row is gathered once, every t row computed
//     once per segment (vertical halo 2 rows per 16, horizontal 2 columns per 30: 1.2x conv1's multiplies instead of 1.75x).
//   * 8 waves, SPECIALISED: waves 0-3 are conv1 (t row t0 + 4s + w of step s), waves 4-7 are conv2 (y rows two behind).
//     Each SIMD holds one wave of each kind.  The HIGH parts of a wave's split weights -- 9 taps x 2 chunks x 16 B per lane
//     = 72 VGPRs -- stay in REGISTERS for the whole segment, the low parts (used by one MFMA in three) in LDS, staged once
//     per workgroup: a tap costs 3 ds_read_b128 for 3 MFMAs and nothing is re-staged per tile.
//   * one barrier per step; the x rows of step s + 1 are in flight (global -> registers) during the MFMAs of step s; conv2's
//     epilogue (bias, skip connection, ELU, stores) is deferred to the start of the next step, so that on every SIMD one
//     wave's epilogue runs under the other wave's MFMAs.
//
// LDS images: ring[slot][chunk of 16 channels][34 pixels][16 x fp16 hi | 16 x scaled fp16 lo | 16 B pad = 80 B] -- the
// pixel stride of conv_s3_kernel, conflict-free for the B fetches (lanes = consecutive pixels of ONE row here, always).
// The skip connection is re-read from global memory as fp32 (exact; it was gathered by this workgroup a few steps before).
#pragma once
#include "conv_split.hip.h"

namespace rt {

struct S3RBSCfg {
    static constexpr int NW = 8, NT = 512;
    static constexpr int SW = 30;                       // output columns per strip
    static constexpr int TCOL = 32, XCOL = 34;          // intermediate / input columns per strip
    static constexpr int STEP = 4, SEG = 16;            // rows per step, output rows per workgroup
    // ring depth: a step reads 6 consecutive rows while the 4 rows of the next step are written -- 10 consecutive rows are live
    static constexpr int RING = 10, PXB = 80;
    static constexpr int ROWB = 2 * XCOL * PXB;         // bytes of one ring row: [chunk][34 pixels][80 B]
    static constexpr int NSLOT = STEP * XCOL * 8;       // 16-byte gathers of one 4-row batch: (channel group, row, pixel)
    static constexpr int NKX = (NSLOT + NT - 1) / NT;   // ... per thread
    static constexpr int WL_SLOTS = 18 * 64;            // 16-byte slots of one convolution's low weight parts: [chunk * 9 + tap][lane]
    static constexpr int NK_WL = (WL_SLOTS + 255) / 256;
};

// Y_SPLIT: the output is written as the PRE-SPLIT tensor (C/8, H, pitch, [8 x fp16 hi | 8 x scaled fp16 lo]) the DMA-fed block reads
// (conv_rbd.hip.h): the first block of a tower, whose input is the fp32 tensor of the 5x5 layer.
template <bool Y_SPLIT>
__global__ void __launch_bounds__(512) conv_s3rbs_kernel(RBArgs a) {
    using Cfg = S3RBSCfg;
    const ConvArgs& p = a.c;
    constexpr int XCOL = Cfg::XCOL, PXB = Cfg::PXB, ROWB = Cfg::ROWB, RING = Cfg::RING, NKX = Cfg::NKX, NT = Cfg::NT;

    __shared__ __attribute__((aligned(16))) char sX[RING * ROWB];
    __shared__ __attribute__((aligned(16))) char sT[RING * ROWB];
    __shared__ __attribute__((aligned(16))) f32x4 sWl[2 * Cfg::WL_SLOTS];     // conv1's | conv2's low weight parts
    __shared__ __attribute__((aligned(16))) float sBias[64];                  // conv1's | conv2's

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is1 = wv < 4;                            // conv1 wave / conv2 wave
    const int wr = wv & 3;                              // row of the step this wave computes
#ifdef RT_KERNEL_TIMING
    // phase stamps of wave 0 (conv1) and wave 4 (conv2): [workgroup][role][16] (tools/time_phases_split.py, RT_TIME_BLOCK=1)
    unsigned long long* dbgp = (p.dbg && (tid & 255) == 0) ? p.dbg + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2 + (tid >> 8)) * 16 : nullptr;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#define RBS_STAMP(i) do { if (dbgp) dbgp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RBS_STAMP(i) do { } while (0)
#endif
    RBS_STAMP(0);

    int tile = blockIdx.x;
    if (p.xcd_order) {                                  // contiguous tile range per XCD (see conv_mfma.hip.h)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int H = p.Hi, W = p.Wi;                       // stride 1, same padding: the three grids coincide
    const int c0 = (tile % p.tiles_x) * Cfg::SW;        // first output column of the strip
    const int y0 = (tile / p.tiles_x) * a.seg;          // first output row of the segment (a.seg rows, a multiple of 4)
    const int y1 = y0 + a.seg < H ? y0 + a.seg : H;
    const int t0 = y0 - 1;                              // first intermediate row
    const int n = blockIdx.z;
    // step s: conv1 rows t0 + 4s .. + 3 (needed up to row y1), conv2 rows t0 + 4s - 5 .. - 2 (valid in [y0, y1))
    const int nstep = (y1 - t0 + 4) / 4 + 1;
    const int last1 = (y1 - t0) / 4;                    // last step with a needed conv1 row

    // ---- gathers: x rows [t0 + 4b + 1, + 4) x columns [c0 - 2, + 34) x 8 channel groups, 16 bytes each ------------------------
    const buf_rsrc rs_x = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, 4));
    const unsigned cs_x = (unsigned)p.x_cstride, rowb_x = (unsigned)p.x_pitch * 16u;
    // per gather slot: byte offset of (group, first row of batch 0, column) -- a batch adds a scalar row offset --, and packed:
    // LDS offset inside a ring row << 8 | (last batch in which the slot's row is needed + 1) << 2 | row of the batch
    unsigned xbase[NKX];
    int xmeta[NKX];
    const int row_hi = (y1 + 1 < H - 1 ? y1 + 1 : H - 1);          // last input row the segment needs
#pragma unroll
    for (int k = 0; k < NKX; k++) {
        const int idx = tid + NT * k;
        const int g = idx / (Cfg::STEP * XCOL), rem = idx - g * (Cfg::STEP * XCOL);
        const int row = rem / XCOL, px = rem - row * XCOL;
        const int ix = c0 - 2 + px;
        const bool valid = idx < Cfg::NSLOT;
        // batch b holds rows t0 + 4b + 1 + row: needed while that is <= row_hi
        const int d = row_hi - (t0 + 1 + row);
        const int bmax = d < 0 ? -1 : (d >> 2);                    // <= 62: segments of up to 248 rows
        xbase[k] = (valid && ix >= 0 && ix < W) ? (unsigned)g * cs_x * 16u + (unsigned)ix * 16u + (unsigned)(t0 + 1 + row) * rowb_x : kBufOOB;
        xmeta[k] = valid ? ((((g >> 2) * XCOL + px) * PXB + (g & 3) * 8) << 8) | ((bmax + 1) << 2) | row : -1;
    }
    f32x4 rin[NKX];
    // batches b >= 0 (rows >= y0 >= 0): one compare + select per slot, the batch's row offset is a scalar
    auto load_batch = [&](int b, f32x4 (&r)[NKX]) {
        const unsigned so = (unsigned)(4 * b) * rowb_x;
#pragma unroll
        for (int k = 0; k < NKX; k++) {
            const bool ok = b < ((xmeta[k] >> 2) & 63);             // b <= bmax (xmeta = -1: xbase is kBufOOB anyway)
            r[k] = buf_load4(rs_x, ok ? xbase[k] : kBufOOB, so);
        }
    };
    auto store_batch = [&](int b, const f32x4 (&r)[NKX]) {
        const int sb = (4 * b + 9) % RING;                          // ring slot of the batch's first row: (iy - t0 + 8) mod RING
#pragma unroll
        for (int k = 0; k < NKX; k++) {
            if (xmeta[k] < 0) continue;
            int slot = sb + (xmeta[k] & 3);
            slot = slot >= RING ? slot - RING : slot;
            const S3Split sp = s3_split(r[k]);
            char* dst = sX + slot * ROWB + (xmeta[k] >> 8);
            *reinterpret_cast<f16x4*>(dst) = sp.hi;
            *reinterpret_cast<f16x4*>(dst + 32) = sp.lo;
        }
    };

    // ---- prologue: the first two batches, this wave's high weight parts (registers), both low parts and biases (LDS) ------------------
    f32x4 rin0[NKX];
#pragma unroll
    for (int k = 0; k < NKX; k++) {                     // batch -1: rows t0 - 3 .. t0, of which t0 - 1 and t0 are needed (if in the image)
        const int iy = t0 - 3 + (xmeta[k] & 3);
        const bool ok = xbase[k] != kBufOOB && iy >= 0 && iy >= t0 - 1 && iy <= row_hi;
        rin0[k] = buf_load4(rs_x, ok ? xbase[k] - 4u * rowb_x : kBufOOB, 0u);
    }
    load_batch(0, rin);
    // weights: each convolution's split weights are fetched ONCE per workgroup (its four waves would otherwise fetch the same
    // 18 KB of high parts each): low parts into sWl for good, high parts through the (still unused) t ring into registers
    {
        // slab order of the split weights (rt_capi.hip: pack_into): [chunk][tap][hi/lo][k-group][co][8 halfs], 16-byte slots
        const buf_rsrc rs_w = make_buf(is1 ? a.w1 : p.w);
        f32x4 rw[2 * Cfg::NK_WL];
#pragma unroll
        for (int k = 0; k < 2 * Cfg::NK_WL; k++) {
            const int idx = (tid & 255) + 256 * k;      // [tap-chunk][hi/lo][lane]: the slab itself
            rw[k] = buf_load4(rs_w, idx < 2 * Cfg::WL_SLOTS ? (unsigned)idx * 16u : kBufOOB, 0u);
        }
        f32x4* whs = reinterpret_cast<f32x4*>(sT) + (is1 ? 0 : Cfg::WL_SLOTS);
#pragma unroll
        for (int k = 0; k < 2 * Cfg::NK_WL; k++) {
            const int idx = (tid & 255) + 256 * k;
            if (idx >= 2 * Cfg::WL_SLOTS) continue;
            const int tt = idx >> 7, hl = (idx >> 6) & 1, ln = idx & 63;
            if (hl) sWl[(is1 ? 0 : Cfg::WL_SLOTS) + tt * 64 + ln] = rw[k];
            else whs[tt * 64 + ln] = rw[k];
        }
    }
    RBS_STAMP(1);                                       // prologue loads issued
    if (tid < 64) sBias[tid] = tid < 32 ? a.bias1[tid] : p.bias[tid - 32];
    store_batch(-1, rin0);
    store_batch(0, rin);
    __syncthreads();
    f16x8 wh[18];                                       // [chunk * 9 + tap]: high parts of this wave's A operands
    {
        const f32x4* whs = reinterpret_cast<const f32x4*>(sT) + (is1 ? 0 : Cfg::WL_SLOTS) + lane;
#pragma unroll
        for (int t = 0; t < 18; t++) wh[t] = __builtin_bit_cast(f16x8, whs[t * 64]);
    }
    __syncthreads();                                    // the t ring is free for conv1's first rows
    RBS_STAMP(2);                                       // first rows and low weight parts in LDS, high parts in registers

    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride + p.y_off, 4));
    const int cs_y = (int)p.y_cstride;
    const char* ring = is1 ? sX : sT;
    const f32x4* wlp = sWl + (is1 ? 0 : Cfg::WL_SLOTS) + lane;
    const int b_lane = l31 * PXB + kg * 16;
    const float* bias_l = sBias + (is1 ? 0 : 32) + 4 * kg;

    // 9 taps x 2 chunks of the 3-row window whose first row sits in ring slot `first`; the operands of tap t + 1 are fetched
    // before the MFMAs of tap t
    auto contract = [&](int first, f32x16& acc_m, f32x16& acc_c) {
        int so[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int slot = first + r;
            so[r] = (slot >= RING ? slot - RING : slot) * ROWB;
        }
        auto bptr = [&](int t) { return ring + so[(t % 9) / 3] + (t / 9) * (XCOL * PXB) + ((t % 9) % 3) * PXB + b_lane; };
#pragma unroll
        for (int r = 0; r < 16; r++) { acc_m[r] = 0.f; acc_c[r] = 0.f; }
        f16x8 bh = *reinterpret_cast<const f16x8*>(bptr(0)), bl = *reinterpret_cast<const f16x8*>(bptr(0) + 32);
        f16x8 al = __builtin_bit_cast(f16x8, wlp[0]);
#pragma unroll
        for (int t = 0; t < 18; t++) {
            // the three LDS reads of tap t + 1 are ISSUED before the three MFMAs of tap t: scheduling barriers pin that order (left
            // alone the scheduler sinks each read to its use, one operand set, and every tap waits for LDS: 55 instead of 32
            // cycles per MFMA, measured)
            f16x8 nh = bh, nl = bl, na = al;
            if (t + 1 < 18) {
                nh = *reinterpret_cast<const f16x8*>(bptr(t + 1));
                nl = *reinterpret_cast<const f16x8*>(bptr(t + 1) + 32);
                na = __builtin_bit_cast(f16x8, wlp[(t + 1) * 64]);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], bh, acc_m, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc_c, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], bl, acc_c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bh = nh; bl = nl; al = na;
        }
    };

    if (is1) {
        // ================= conv1 waves: x ring -> t ring =================
        for (int s = 0; s < nstep; s++) {
            const bool more = s + 1 <= last1;           // conv1 of step s + 1 needs batch s + 1
            if (more) load_batch(s + 1, rin);
            if (s <= last1) {
                // all four rows of the step are computed, needed or not (rows past y1 read zero-filled x rows and land in ring
                // slots nobody reads); rows outside the image become conv2's zero padding
                const int row = t0 + 4 * s + wr;
                f32x16 acc_m, acc_c;
                contract((4 * s + wr + 7) % RING, acc_m, acc_c);          // x rows row - 1 .. row + 1: slot (row - 1 - t0 + 8) mod RING
                if (s == 1 || s == 2) RBS_STAMP(4 * s);                    // 4, 8: MFMAs issued
                const int gx = c0 - 1 + l31;
                const bool inside = row >= 0 && row < H && gx >= 0 && gx < W;
                char* trow = sT + ((4 * s + wr + 8) % RING) * ROWB + l31 * PXB + kg * 8;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + 8 * q);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float v = apply_act_fast(fmaf(acc_c[4 * q + e], kSplitInv, acc_m[4 * q + e]) + bv[e], 1);
                        o[e] = inside ? v : 0.f;
                    }
                    const S3Split sp = s3_split(o);
                    char* dst = trow + (q >> 1) * (XCOL * PXB) + (q & 1) * 16;
                    *reinterpret_cast<f16x4*>(dst) = sp.hi;
                    *reinterpret_cast<f16x4*>(dst + 32) = sp.lo;
                }
            }
            if (s == 1 || s == 2) RBS_STAMP(4 * s + 1);    // 5, 9: epilogue issued
            if (more) store_batch(s + 1, rin);
            if (s == 1 || s == 2) RBS_STAMP(4 * s + 2);    // 6, 10: next rows written to LDS
            __syncthreads();
            if (s == 0) RBS_STAMP(3);
            else if (s == 1 || s == 2) RBS_STAMP(4 * s + 3);   // 7, 11: barrier passed
            else if (s == 3 || s == 4) RBS_STAMP(9 + s);       // 12, 13
        }
    } else {
        // ================= conv2 waves: t ring -> y =================
        // The epilogue of a row runs at the START of the following step, while the conv1 wave of the same SIMD issues its MFMAs
        // (and conv1's epilogue runs under conv2's MFMAs): bias, skip connection, activation, 16-byte stores.
        f32x16 acc_m, acc_c;
        f32x4 rr[4];
        int prow = -1;                                  // output row whose accumulators are pending
        const int ox = c0 + l31;
        const bool col_ok = l31 < Cfg::SW && ox < W;
        auto epilogue2 = [&]() {
            const unsigned vo = (col_ok && prow >= 0) ? (unsigned)((prow * p.y_ystride + ox) * 4 + 4 * kg * cs_y) * 4u : kBufOOB;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + 8 * q);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    o[e] = apply_act_fast(fmaf(acc_c[4 * q + e], kSplitInv, acc_m[4 * q + e]) + (bv[e] + rr[q][e]), 1);
                if (Y_SPLIT) {
                    // lane (pixel, kg) holds channels 8 q + 4 kg .. + 3: the kg-th half of group q's 16-byte hi slot and of its lo slot
                    const S3Split sp = s3_split(o);
                    const unsigned vs = (col_ok && prow >= 0) ? (unsigned)(prow * p.y_ystride + ox) * 32u + 8u * kg : kBufOOB;
                    buf_store2(__builtin_bit_cast(f32x2_t, sp.hi), rs_y, vs, (unsigned)(8 * q * cs_y) * 4u);
                    buf_store2(__builtin_bit_cast(f32x2_t, sp.lo), rs_y, vs, (unsigned)(8 * q * cs_y) * 4u + 16u);
                } else {
                    buf_store4(o, rs_y, vo, (unsigned)(8 * q * cs_y) * 4u);          // see common.hip.h: no register soffset on 16-byte stores
                }
            }
        };
        for (int s = 0; s < nstep; s++) {
            const bool more = s + 1 <= last1;
            if (more) load_batch(s + 1, rin);
            if (s >= 2) epilogue2();                    // rows of step s - 1 (prow < 0: nothing is stored)
            if (s == 1 || s == 2) RBS_STAMP(4 * s);     // 4, 8: previous rows stored
            if (s >= 1) {
                // rows above / below the segment are computed too (garbage in, nothing stored): no divergent accumulator paths
                const int row = t0 + 4 * s - 5 + wr;
                contract((4 * s + wr + 2) % RING, acc_m, acc_c);          // t rows row - 1 .. row + 1
                prow = (row >= y0 && row < y1) ? row : -1;
                // skip connection: fp32 x at (row, c0 + l31), requested after the MFMAs (their operand registers are free again)
                // and consumed after the barrier
                const unsigned vo = (col_ok && prow >= 0) ? (unsigned)(row * p.x_pitch + ox) * 16u + (unsigned)(kg * cs_x) * 16u : kBufOOB;
#pragma unroll
                for (int q = 0; q < 4; q++) rr[q] = buf_load4(rs_x, vo, (unsigned)(2 * q) * cs_x * 16u);
            }
            if (s == 1 || s == 2) RBS_STAMP(4 * s + 1);    // 5, 9: MFMAs issued
            if (more) store_batch(s + 1, rin);
            if (s == 1 || s == 2) RBS_STAMP(4 * s + 2);    // 6, 10: next rows written to LDS
            __syncthreads();
            if (s == 0) RBS_STAMP(3);
            else if (s == 1 || s == 2) RBS_STAMP(4 * s + 3);   // 7, 11: barrier passed
            else if (s == 3 || s == 4) RBS_STAMP(9 + s);       // 12, 13
        }
        epilogue2();                                    // nstep >= 2 always
    }
#ifdef RT_KERNEL_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    if (dbgp) { dbgp[14] = __builtin_amdgcn_s_memtime(); dbgp[15] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
#undef RBS_STAMP
}

}  // namespace rt
