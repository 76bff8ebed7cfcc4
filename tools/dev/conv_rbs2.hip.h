// conv_s3rbs2_kernel: the streaming residual block of conv_rbs.hip.h with every wave's vector work INSIDE its own matrix stream.
//
//       y = ELU( conv3x3( ELU( conv3x3(x) + b1 ) ) + b2 + x )          32 -> 32 -> 32 channels, stride 1, interleaved tensors
//
// (reference resnet18_2D_513x257_net.cpp:66-575: resblockN_conv1 -> ELU -> resblockN_conv2 -> add -> ELU, 8 blocks per side.)
//
// Why a second version.  Round 2's kernel gave each SIMD one conv1 wave and one conv2 wave and relied on one wave's epilogue
// running "under" the other wave's MFMAs.  tools/micro/mfma_interleave.hip (profiles/r03_mfma_interleave.txt) shows what a gfx950
// SIMD really does: a v_mfma_f32_32x32x16_f16 holds the vector issue port for ~14.6 of its 32 cycles, the remaining ~17 cycles take
// 4-cycle VALU instructions (v_exp_f32, v_cvt_pk_f16_f32: 8 cycles; packed fp32: not at all) FROM THE SAME STREAM; a wave that issues
// MFMAs back to back starves its partner's VALU (round 2's phase stamps: the partner's 130-instruction epilogue took 3.5 k cycles).
// So here every MFMA is followed by a SLICE of <= 16 cycles of that wave's own deferred vector work, pinned with scheduling barriers:
//
//   conv1 wave, step s:   54 MFMAs of t row (s)      | slices: ELU + fp16 split + LDS stores of t row (s - 1)  (accumulators double-buffered)
//   conv2 wave, step s:   54 MFMAs of y row (s)      | slices: requests for the x rows of step s + 1 and for the skip row of step s + 1,
//                                                    |         ELU + 16-byte stores of y row (s - 1), fp16 split + LDS stores of the x rows
//
// Bias and skip connection cost no vector instruction: conv2's main accumulator chain STARTS from the skip row (srcC of the first MFMA),
// both cross-term chains start from bias * 2^11 (read from LDS straight into the accumulator registers).  Deferring conv1's epilogue puts
// conv2 one more step behind (y rows t0 + 4s - 9 ..); a segment of 32 rows takes 11 steps + a tail instead of 10, each ~0.6x as long.
// The out-of-image zeroing of t (conv2's zero padding) is a masked store of zeros AFTER the step's regular stores, only in workgroups
// / rows that touch the border.  No packed-fp32 instructions: the translation unit is built with -fno-slp-vectorize (build.py).
//
// LDS images, ring geometry and weight staging are conv_rbs.hip.h's: ring[slot][chunk of 16 channels][34 pixels][16 x fp16 hi |
// 16 x scaled fp16 lo | 16 B pad = 80 B], 10 rows each for x and t, high weight parts in 72 VGPRs per wave, low parts in LDS.
#pragma once
#include <utility>
#include "kernels/conv_split.hip.h"

namespace rt {

template <typename F, int... I>
__device__ static __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ static __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#ifndef RB2_PRIO2
#define RB2_PRIO2 0    // s_setprio of the conv2 waves (the younger half of the workgroup loses every arbitration at equal priority)
#endif
#ifndef RB2_PRIO1
#define RB2_PRIO1 0
#endif
#ifndef RB2_ABL
#define RB2_ABL 0      // tools/dev/rbs_dev.hip: 1 no LDS stores, 2 no global loads / stores, 4 no slices at all, 8 no LDS operand reads (steady state; results wrong)
#endif
struct S3RB2Cfg {
    static constexpr int NW = 8, NT = 512;
    static constexpr int SW = 30;                       // output columns per strip
    static constexpr int TCOL = 32, XCOL = 34;          // intermediate / input columns per strip
    static constexpr int STEP = 4, SEG = 16;            // rows per step, default output rows per workgroup
    static constexpr int RING = 10, PXB = 80;
    static constexpr int ROWB = 2 * XCOL * PXB;         // bytes of one ring row: [chunk][34 pixels][80 B]
    static constexpr int NSLOT = STEP * XCOL * 8;       // 16-byte gathers of one 4-row batch: (channel group, row, pixel) = 1088 = 17 x 64
    static constexpr int NKG = 5;                       // ... per thread of a 256-thread half (the 5th pass re-gathers unit 16 in all four waves)
    static constexpr int WL_SLOTS = 18 * 64;            // 16-byte slots of one convolution's low weight parts: [chunk * 9 + tap][lane]
    static constexpr int NK_WL = (WL_SLOTS + 255) / 256;
};

// value barrier: everything that computes v is issued before this point, everything that uses it after (no instruction)
#ifndef HIPEMU
#define RB2_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define RB2_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
typedef unsigned long long rb2_mask;
// x > 0 as a lane mask in an SGPR pair / select by such a mask.  Written out because the compiler funnels compare + select pairs
// through VCC with two wait states each (gfx950: VALU write of an SGPR -> VALU read); four compares in one slice and the four
// selects in the next have the distance built in.
__device__ static __forceinline__ rb2_mask rb2_gt0(float x) {
    rb2_mask m;
    asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m) : "v"(x));
    return m;
}
__device__ static __forceinline__ float rb2_sel(rb2_mask m, float if_set, float if_clear) {
    float r;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
// fp16 split x = h + l * 2^-11 in 2.5 instructions per value (s3_split's arithmetic bit for bit: h = fp16(x) to nearest even,
// x - h exact, l = fp16((x - h) * 2^11)): v_cvt_pk_f16_f32, v_fma_mix_f32 (fp16 source widened in the instruction),
// v_fma_mixlo/hi_f16 (product rounded once, to fp16)
__device__ static __forceinline__ unsigned rb2_hi2(float a, float b) {
    unsigned r;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ static __forceinline__ float rb2_rem_lo(unsigned h, float x) {      // x - fp16 in the low half of h
    float d;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(x));
    return d;
}
__device__ static __forceinline__ float rb2_rem_hi(unsigned h, float x) {      // x - fp16 in the high half of h
    float d;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(x));
    return d;
}
__device__ static __forceinline__ unsigned rb2_lo2(float d0, float d1) {       // fp16(d0 * 2^11) | fp16(d1 * 2^11) << 16
    unsigned r;
    asm volatile("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %2, %3, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]"
                 : "=&v"(r) : "v"(d0), "v"(d1), "s"(kSplitScale));
    return r;
}
#else
#define RB2_PIN4(a, b, c, d) ((void)0)
#define RB2_PIN2(a, b) ((void)0)
typedef bool rb2_mask;
static inline rb2_mask rb2_gt0(float x) { return x > 0.f; }
static inline float rb2_sel(rb2_mask m, float if_set, float if_clear) { return m ? if_set : if_clear; }
static inline unsigned rb2_pack_halfs(_Float16 a, _Float16 b) {
    unsigned short ua, ub;
    std::memcpy(&ua, &a, 2); std::memcpy(&ub, &b, 2);
    return (unsigned)ua | ((unsigned)ub << 16);
}
static inline _Float16 rb2_half_of(unsigned h, int hi) {
    const unsigned short u = (unsigned short)(hi ? h >> 16 : h & 0xffffu);
    _Float16 r;
    std::memcpy(&r, &u, 2);
    return r;
}
static inline unsigned rb2_hi2(float a, float b) { return rb2_pack_halfs((_Float16)a, (_Float16)b); }
static inline float rb2_rem_lo(unsigned h, float x) { return x - (float)rb2_half_of(h, 0); }
static inline float rb2_rem_hi(unsigned h, float x) { return x - (float)rb2_half_of(h, 1); }
static inline unsigned rb2_lo2(float d0, float d1) { return rb2_pack_halfs((_Float16)(d0 * kSplitScale), (_Float16)(d1 * kSplitScale)); }
#endif

__global__ void __launch_bounds__(512) conv_s3rbs2_kernel(RBArgs a) {
    using Cfg = S3RB2Cfg;
    const ConvArgs& p = a.c;
    constexpr int XCOL = Cfg::XCOL, PXB = Cfg::PXB, ROWB = Cfg::ROWB, RING = Cfg::RING, NKG = Cfg::NKG;
    constexpr float kLog2e = 1.44269504088896341f;

    __shared__ __attribute__((aligned(16))) char sX[RING * ROWB];
    __shared__ __attribute__((aligned(16))) char sT[RING * ROWB];
    __shared__ __attribute__((aligned(16))) f32x4 sWl[2 * Cfg::WL_SLOTS];     // conv1's | conv2's low weight parts
    __shared__ __attribute__((aligned(16))) float sBias[64];                  // (conv1's | conv2's) * 2^11: start of the cross-term chains

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
#define RB2_STAMP(i) do { if (dbgp) dbgp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RB2_STAMP(i) do { } while (0)
#endif
    RB2_STAMP(0);

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
    // step s: conv1 multiplies t rows t0 + 4s .. + 3 (needed up to row y1) and finishes the rows of step s - 1;
    //         conv2 multiplies y rows t0 + 4s - 9 .. - 6 (valid in [y0, y1)) and finishes the rows of step s - 1
    const int last1 = (y1 - t0) / 4;                    // last step with a needed conv1 row
    const int nstep = (y1 - y0 + 9) / 4 + 1;            // the step of y row y1 - 1 is the last one

    // ---- gathers: x rows [t0 + 4b + 1, + 4) x columns [c0 - 2, + 34) x 8 channel groups, 16 bytes each -----------------------------
    // One mapping for both 256-thread halves: slot = t2 + 256 k, k < 5 (the 5th pass wraps to unit 16 in every wave: the same 64
    // slots are written four times with the same bytes).  Steady state: the conv2 half gathers; prologue: conv1 half batch -1, conv2
    // half batch 0.
    const buf_rsrc rs_x = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, 4));
    const unsigned cs_x = (unsigned)p.x_cstride, rowb_x = (unsigned)p.x_pitch * 16u;
    unsigned xbase[NKG];                                // byte offset of (group, first row of batch 0, column); a batch adds a scalar row offset
    int xmeta[NKG];                                     // LDS offset inside a ring row << 8 | (last batch in which the slot's row is needed + 1) << 2 | row
    const int row_hi = (y1 + 1 < H - 1 ? y1 + 1 : H - 1);          // last input row the segment needs
    const int t2 = tid & 255;
#pragma unroll
    for (int k = 0; k < NKG; k++) {
        const int idx = k < 4 ? t2 + 256 * k : 1024 + (t2 & 63);
        const int g = idx / (Cfg::STEP * XCOL), rem = idx - g * (Cfg::STEP * XCOL);
        const int row = rem / XCOL, px = rem - row * XCOL;
        const int ix = c0 - 2 + px;
        const int d = row_hi - (t0 + 1 + row);          // batch b holds rows t0 + 4b + 1 + row: needed while that is <= row_hi
        const int bmax = d < 0 ? -1 : (d >> 2);         // <= 62: segments of up to 248 rows
        xbase[k] = (ix >= 0 && ix < W) ? (unsigned)g * cs_x * 16u + (unsigned)ix * 16u + (unsigned)(t0 + 1 + row) * rowb_x : kBufOOB;
        xmeta[k] = ((((g >> 2) * XCOL + px) * PXB + (g & 3) * 8) << 8) | ((bmax + 1) << 2) | row;
    }
    f32x4 rin[NKG];
    auto load_slot = [&](int b, int k) {                // batches b >= 0 (rows >= y0 >= 0): one compare + select, the row offset is a scalar
        if (RB2_ABL & 2) return;
        const bool ok = b < ((xmeta[k] >> 2) & 63);
        rin[k] = buf_load4(rs_x, ok ? xbase[k] : kBufOOB, (unsigned)(4 * b) * rowb_x);
    };
    auto slot_dst = [&](int b, int k) -> char* {        // where slot k of batch b lives in the x ring
        int slot = (4 * b + 9) % RING + (xmeta[k] & 3); // ring slot of the batch's first row: (iy - t0 + 8) mod RING
        slot = slot >= RING ? slot - RING : slot;
        return sX + slot * ROWB + (xmeta[k] >> 8);
    };

    // ---- prologue: the first two batches, this wave's high weight parts (registers), both low parts and biases (LDS) ------------------
    if (is1) {
#pragma unroll
        for (int k = 0; k < NKG; k++) {                 // batch -1: rows t0 - 3 .. t0, of which t0 - 1 and t0 are needed (if in the image)
            const int iy = t0 - 3 + (xmeta[k] & 3);
            const bool ok = xbase[k] != kBufOOB && iy >= 0 && iy >= t0 - 1 && iy <= row_hi;
            rin[k] = buf_load4(rs_x, ok ? xbase[k] - 4u * rowb_x : kBufOOB, 0u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NKG; k++) load_slot(0, k);
    }
    // weights: each convolution's split weights are fetched ONCE per workgroup: low parts into sWl for good, high parts through the
    // (still unused) t ring into registers
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
    RB2_STAMP(1);                                       // prologue loads issued
    if (tid < 64) sBias[tid] = (tid < 32 ? a.bias1[tid] : p.bias[tid - 32]) * kSplitScale;
#pragma unroll
    for (int k = 0; k < NKG; k++) {
        const S3Split sp = s3_split(rin[k]);
        char* dst = slot_dst(is1 ? -1 : 0, k);
        *reinterpret_cast<f16x4*>(dst) = sp.hi;
        *reinterpret_cast<f16x4*>(dst + 32) = sp.lo;
    }
    __syncthreads();
    f16x8 wh[18];                                       // [chunk * 9 + tap]: high parts of this wave's A operands
    {
        const f32x4* whs = reinterpret_cast<const f32x4*>(sT) + (is1 ? 0 : Cfg::WL_SLOTS) + lane;
#pragma unroll
        for (int t = 0; t < 18; t++) wh[t] = __builtin_bit_cast(f16x8, whs[t * 64]);
    }
    __syncthreads();                                    // the t ring is free for conv1's first rows
    RB2_STAMP(2);                                       // first rows and low weight parts in LDS, high parts in registers

    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride + p.y_off, 4));
    const int cs_y = (int)p.y_cstride;
    const char* ring = is1 ? sX : sT;
    const f32x4* wlp = sWl + (is1 ? 0 : Cfg::WL_SLOTS) + lane;
    const int b_lane = l31 * PXB + kg * 16;
    const float* bias_l = sBias + (is1 ? 0 : 32) + 4 * kg;

    // The 54 MFMAs of one row -- 9 taps x 2 chunks of the 3-row window whose first row sits in ring slot `first` -- with slice(i)
    // after MFMA i.  The operands of tap t + 1 are requested before the MFMAs of tap t.  The main chain starts from `c_init` (zero /
    // skip row), the cross-term chain from bias * 2^11.  Returns the finished row  m + c * 2^-11  (16 values per lane).
    auto contract = [&](int first, const f32x16& c_init, auto&& slice) -> f32x16 {
        int so[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int slot = first + r;
            so[r] = (slot >= RING ? slot - RING : slot) * ROWB;
        }
        auto bptr = [&](int t) { return ring + so[(t % 9) / 3] + (t / 9) * (XCOL * PXB) + ((t % 9) % 3) * PXB + b_lane; };
        f32x16 acc_m, acc_c;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; e++) acc_c[4 * q + e] = bv[e];
        }
        f16x8 bh = *reinterpret_cast<const f16x8*>(bptr(0)), bl = *reinterpret_cast<const f16x8*>(bptr(0) + 32);
        f16x8 al = __builtin_bit_cast(f16x8, wlp[0]);
        static_for<18>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            f16x8 nh = bh, nl = bl, na = al;
            if constexpr (t + 1 < 18 && !(RB2_ABL & 8)) {
                nh = *reinterpret_cast<const f16x8*>(bptr(t + 1));
                nl = *reinterpret_cast<const f16x8*>(bptr(t + 1) + 32);
                na = __builtin_bit_cast(f16x8, wlp[(t + 1) * 64]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (t == 0) acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], bh, c_init, 0, 0, 0);
            else acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], bh, acc_m, 0, 0, 0);
            slice(std::integral_constant<int, 3 * t>{});
            __builtin_amdgcn_sched_barrier(0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc_c, 0, 0, 0);
            slice(std::integral_constant<int, 3 * t + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], bl, acc_c, 0, 0, 0);
            slice(std::integral_constant<int, 3 * t + 2>{});
            __builtin_amdgcn_sched_barrier(0);
            bh = nh; bl = nl; al = na;
        });
        f32x16 out;
#pragma unroll
        for (int r = 0; r < 16; r++) out[r] = fmaf(acc_c[r], kSplitInv, acc_m[r]);
        return out;
    };

    // ELU of four values of the pending row in six phases of <= 16 cycles of vector issue:
    //   0: e = v * log2(e)     1, 2: e = 2^e (two per phase)     3: e -= 1     4: masks v > 0     5: v = mask ? v : e
    float v[4], e[4];
    rb2_mask mk[4];
    auto elu_phase = [&](auto phc, int q, const f32x16& pend) {
        constexpr int ph = decltype(phc)::value;
        if constexpr (ph == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) e[j] = pend[4 * q + j] * kLog2e;
            RB2_PIN4(e[0], e[1], e[2], e[3]);
        } else if constexpr (ph == 1) {
            e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
            RB2_PIN2(e[0], e[1]);
        } else if constexpr (ph == 2) {
            e[2] = __builtin_amdgcn_exp2f(e[2]); e[3] = __builtin_amdgcn_exp2f(e[3]);
            RB2_PIN2(e[2], e[3]);
        } else if constexpr (ph == 3) {
#pragma unroll
            for (int j = 0; j < 4; j++) e[j] -= 1.f;
            RB2_PIN4(e[0], e[1], e[2], e[3]);
        } else if constexpr (ph == 4) {
#pragma unroll
            for (int j = 0; j < 4; j++) mk[j] = rb2_gt0(pend[4 * q + j]);
        } else if constexpr (ph == 5) {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = rb2_sel(mk[j], pend[4 * q + j], e[j]);
        }
    };
    // fp16 split of four values in three phases: high parts, remainders, scaled low parts; then one 2 x 8-byte LDS store
    unsigned h01, h23, l01, l23;
    float dd[4];
    auto split_phase = [&](auto phc, float x0, float x1, float x2, float x3) {
        constexpr int ph = decltype(phc)::value;
        if constexpr (ph == 0) {
            h01 = rb2_hi2(x0, x1); h23 = rb2_hi2(x2, x3);
        } else if constexpr (ph == 1) {
            dd[0] = rb2_rem_lo(h01, x0); dd[1] = rb2_rem_hi(h01, x1); dd[2] = rb2_rem_lo(h23, x2); dd[3] = rb2_rem_hi(h23, x3);
        } else if constexpr (ph == 2) {
            l01 = rb2_lo2(dd[0], dd[1]); l23 = rb2_lo2(dd[2], dd[3]);
        }
    };
    auto split_store = [&](char* dst) {
        if (RB2_ABL & 1) return;
        *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{h01, h23};
        *reinterpret_cast<u32x2_t*>(dst + 32) = u32x2_t{l01, l23};
    };
    // gather slot k of batch b: 3 split phases + the store
    auto gather_phase = [&](auto phc, int b, int k) {
        constexpr int ph = decltype(phc)::value;
        if constexpr (ph < 3) split_phase(phc, rin[k][0], rin[k][1], rin[k][2], rin[k][3]);
        else split_store(slot_dst(b, k));
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 pend = zero16;                               // the row multiplied in the previous step, finished in this one
    constexpr int KG1 = 3;                              // gather slots 0 .. 2: conv1 waves, 3 .. 4: conv2 waves

    if (is1) {
        // ================= conv1 waves: x ring -> t ring, gather slots 0 .. 2 =================
        if (RB2_PRIO1) __builtin_amdgcn_s_setprio(RB2_PRIO1);
        const int gx = c0 - 1 + l31;
        const bool col_in = gx >= 0 && gx < W;
        const bool cols_all_in = c0 >= 1 && c0 + 30 < W;        // every lane's column is inside the image (wave-uniform)
        for (int s = 0; s < nstep; s++) {
            if (s <= last1 + 1) {
                // MFMAs of t row t0 + 4s + wr; slices: requests for batch s + 1, finish of t row t0 + 4(s - 1) + wr, split + stores of the batch
                const int b = s + 1;
                const int orow = t0 + 4 * (s - 1) + wr;
                char* trow = sT + ((4 * (s - 1) + wr + 8 + RING) % RING) * ROWB + l31 * PXB + kg * 8;
                pend = contract((4 * s + wr + 7) % RING, zero16, [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr ((RB2_ABL & 4) != 0) return;
                    if constexpr (i == 0) { load_slot(b, 0); load_slot(b, 1); }
                    else if constexpr (i == 1) load_slot(b, 2);
                    else if constexpr (i < 38) {            // four values per 9 slices
                        constexpr int q = (i - 2) / 9, ph = (i - 2) % 9;
                        if constexpr (ph < 6) elu_phase(std::integral_constant<int, ph>{}, q, pend);
                        else {
                            split_phase(std::integral_constant<int, ph - 6>{}, v[0], v[1], v[2], v[3]);
                            if constexpr (ph == 8) split_store(trow + (q >> 1) * (XCOL * PXB) + (q & 1) * 16);
                        }
                    } else if constexpr (i < 38 + 4 * KG1) {
                        gather_phase(std::integral_constant<int, (i - 38) % 4>{}, b, (i - 38) / 4);
                    }
                });
                // rows / columns outside the image are conv2's zero padding: overwrite them (border workgroups only)
                const bool row_in = orow >= 0 && orow < H;
                if (!(row_in && cols_all_in)) {
                    if (!(row_in && col_in)) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            char* dst = trow + (q >> 1) * (XCOL * PXB) + (q & 1) * 16;
                            *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{0u, 0u};
                            *reinterpret_cast<u32x2_t*>(dst + 32) = u32x2_t{0u, 0u};
                        }
                    }
                }
            }
            if (s == 5) RB2_STAMP(13);                  // end of this wave's step-5 stream: the rest up to stamp 8 is barrier wait
            __syncthreads();
            if (s < 10) RB2_STAMP(3 + s);
        }
    } else {
        // ================= conv2 waves: t ring -> y, gather slots 3 .. 4 =================
        if (RB2_PRIO2) __builtin_amdgcn_s_setprio(RB2_PRIO2);
        const int ox = c0 + l31;
        const bool col_ok = l31 < Cfg::SW && ox < W;
        const unsigned skip_col = (unsigned)ox * 16u + (unsigned)(kg * cs_x) * 16u;
        f32x16 rr = zero16;                             // skip row of the NEXT step's y row: start of its main accumulator chain
        auto load_skip = [&](int row) {                 // fp32 x at (row, c0 + l31): the exact skip connection
            if (RB2_ABL & 2) return;
            const bool ok = col_ok && row >= y0 && row < y1;
            const unsigned vo = ok ? (unsigned)row * rowb_x + skip_col : kBufOOB;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 t = buf_load4(rs_x, vo, (unsigned)(2 * q) * cs_x * 16u);
#pragma unroll
                for (int j = 0; j < 4; j++) rr[4 * q + j] = t[j];
            }
        };
        // finishing slices of the pending y row `prow` (-1: not a row of this segment): four values per 6 slices
        auto fin = [&](auto ic, int prow) {
            constexpr int i = decltype(ic)::value;
            constexpr int q = i / 6, ph = i % 6;
            elu_phase(std::integral_constant<int, ph>{}, q, pend);
            if constexpr (ph == 5) {
                const unsigned vo = (col_ok && prow >= 0) ? (unsigned)((prow * p.y_ystride + ox) * 4 + 4 * kg * cs_y) * 4u : kBufOOB;
                if (!(RB2_ABL & 2)) buf_store4(f32x4{v[0], v[1], v[2], v[3]}, rs_y, vo, (unsigned)(8 * q * cs_y) * 4u);
            }
        };
        auto row_of = [&](int s) { return t0 + 4 * s - 9 + wr; };
        auto seg_row = [&](int r) { return (r >= y0 && r < y1) ? r : -1; };
        for (int s = 0; s < nstep; s++) {
            const int b = s + 1;
            if (s < 2) {                                // nothing to multiply yet: gather only
#pragma unroll
                for (int k = KG1; k < NKG; k++) load_slot(b, k);
                if (s == 1) load_skip(row_of(2));
#pragma unroll
                for (int k = KG1; k < NKG; k++) {
                    const S3Split sp = s3_split(rin[k]);
                    char* dst = slot_dst(b, k);
                    *reinterpret_cast<f16x4*>(dst) = sp.hi;
                    *reinterpret_cast<f16x4*>(dst + 32) = sp.lo;
                }
            } else {
                // MFMAs of y row t0 + 4s - 9 + wr starting from its skip row; slices: requests (batch s + 1, next skip row), finish of
                // y row (s - 1), split + stores of the batch
                const int prow = seg_row(row_of(s - 1));
                pend = contract((4 * s + wr + 8) % RING, rr, [&](auto ic) {        // t rows row - 1 .. row + 1: slot (row - 1 - t0 + 8) mod RING
                    constexpr int i = decltype(ic)::value;
                    if constexpr ((RB2_ABL & 4) != 0) return;
                    if constexpr (i == 0) { load_slot(b, 3); load_slot(b, 4); }
                    else if constexpr (i == 1) load_skip(row_of(s + 1));
                    else if constexpr (i < 26) fin(std::integral_constant<int, i - 2>{}, prow);
                    else if constexpr (i < 26 + 4 * (NKG - KG1)) gather_phase(std::integral_constant<int, (i - 26) % 4>{}, b, KG1 + (i - 26) / 4);
                });
            }
            if (s == 5) RB2_STAMP(13);
            __syncthreads();
            if (s < 10) RB2_STAMP(3 + s);
        }
        // tail: the y row of the last step
        {
            const int prow = seg_row(row_of(nstep - 1));
            static_for<24>([&](auto ic) { fin(ic, prow); });
        }
    }
#ifdef RT_KERNEL_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    if (dbgp) { dbgp[14] = __builtin_amdgcn_s_memtime(); dbgp[15] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
#undef RB2_STAMP
}

}  // namespace rt
