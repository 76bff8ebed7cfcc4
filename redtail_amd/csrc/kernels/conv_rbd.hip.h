// conv_s3rbd_kernel: the tower block of conv_rbs.hip.h, fed by LDS-DMA from PRE-SPLIT tensors (round 6).
//
//       y = ELU( conv3x3( ELU( conv3x3(x) + b1 ) ) + b2 + x )          32 -> 32 -> 32 channels, stride 1
//
// (reference resnet18_2D_513x257_net.cpp:66-575: resblockN_conv1 -> ELU -> resblockN_conv2 -> add -> ELU, 8 blocks per side.)
// conv_s3rbs_kernel is bound by its vector instructions, not by its MFMAs (profiles/r05_pmc_layer_resblock.txt: 4.9 VALU + 1.2 LDS
// instructions per MFMA): every 16-byte slot of x costs a buffer load, a select, the fp16 split (5 VALU per pair of values) and two
// ds_write_b64, on all 8 waves, before any of it reaches the matrix cores.  Here the tensors BETWEEN the tower blocks are stored as the
// operands the MFMAs want -- per pixel and group of 8 channels 32 bytes: [8 x fp16 hi | 8 x scaled fp16 lo], x = hi + lo * 2^-11, the
// same 4 bytes per element as fp32 -- written by the producing block's epilogue, which has the value in registers anyway:
//   * the x ring is filled by `buffer_load_dwordx4 ... lds`: 17 one-KB pieces per 4-row step and workgroup, no VALU, no ds_write, no
//     staging registers; image borders are the zeros of out-of-range lanes;
//   * ring images have NO pad: pixel p keeps its hi half in 16-byte slot (p >> 3) & 1 of its 32-byte record (lo in the other), which makes
//     every ds_read_b128 lane group of a B fetch touch all 64 banks once (MI355X_MICROARCH.md, LDS: the b128 groups pair pixels 8 and 24
//     apart), for all three column shifts of a 3x3 window; the DMA lane -> global address map does the swizzle for free;
//   * output channels are permuted on the HOST (rows of the A operands): lane (pixel, kg) of an accumulator owns channels 8 kg .. 8 kg + 7
//     and 16 + 8 kg .. + 7 -- two whole 8-groups, so t goes to its ring and y to memory as 16-byte hi / lo slots;
//   * the bias is the C operand of the first MFMA; the skip connection is read from the x ring (it is still there: 16 rows), as
//     hi + lo * 2^-11 = x to 22 bits (v_fma_mix_f32 on the fp16 halves).
// Structure as before: strip of 30 columns x segment of rows, 4 rows per step, waves 0-3 conv1 (x ring -> t ring), waves 4-7 conv2
// (t ring -> y, two steps behind), high weight parts in 72 VGPRs, low parts in LDS, one barrier per step.
// The first block of a tower (fp32 input from the 5x5 layer) runs conv_s3rbs_kernel with a split-writing epilogue; the last block writes
// fp32 (Y_SPLIT = false) for the layers behind the towers.
#pragma once
#include "conv_split.hip.h"

namespace rt {

struct S3RBDCfg {
    static constexpr int NW = 8, NT = 512;
    static constexpr int SW = 30, XCOL = 34, TCOL = 32, STEP = 4;
    static constexpr int RX = 16, RT = 10;                      // ring rows: x (conv1's window, the skip rows of conv2, the landing batch), t
    static constexpr int GXB = XCOL * 32, XROWB = 4 * GXB;      // bytes of one 8-channel group / of one row in the x ring
    static constexpr int GTB = TCOL * 32, TROWB = 4 * GTB;
    static constexpr int BSLOTS = STEP * XROWB / 16;            // 16-byte slots of one 4-row batch: 1088 = 17 pieces of 64
    static constexpr int NPIECE = BSLOTS / 64;
    static constexpr int PPW = (NPIECE + NW - 1) / NW;          // pieces per wave (3: piece w, w + 8, and wave 0 the 17th)
    static constexpr int WL_SLOTS = 18 * 64;                    // 16-byte slots of one convolution's low (or high) weight parts
    static_assert(BSLOTS % 64 == 0, "a batch is a whole number of DMA pieces");
};

// hi half of pixel p's 32-byte ring record: slot (p >> 3) & 1
__device__ static __forceinline__ int rbd_swz(int p) { return ((p >> 3) & 1) * 16; }

// direct global -> LDS load of 16 bytes per lane, issued behind the compiler's back: the waitcnt pass orders every later ds_read behind a
// builtin LDS-DMA it knows of (vmcnt(0) right after the issue); here the kernel waits itself, once per step, before the barrier.
__device__ static __forceinline__ void rbd_dma16(buf_rsrc rs, void* lds_wave_base, unsigned voff, unsigned soff) {
#ifdef HIPEMU
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds_wave_base, 16, voff, soff, 0, 0);
#else
    const unsigned m = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
#endif
}

// hi + lo * 2^-11 of element j (0 / 1) of packed fp16 pairs
template <int J>
__device__ static __forceinline__ float rbd_join(unsigned hi2, unsigned lo2, float inv) {
#ifdef HIPEMU
    const _Float16 h = __builtin_bit_cast(_Float16, (unsigned short)(hi2 >> (16 * J))), l = __builtin_bit_cast(_Float16, (unsigned short)(lo2 >> (16 * J)));
    return fmaf((float)l, inv, (float)h);
#else
    float r;
    if (J == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(lo2), "v"(inv), "v"(hi2));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(lo2), "v"(inv), "v"(hi2));
    return r;
#endif
}

__device__ static __forceinline__ float rbd_elu(float v) { return elu_fast(v); }        // common.hip.h

// development switches (tools/r06/rbd_dev.sh builds variants; the product has the defaults)
#ifndef RT_RBD_PRIO
#define RT_RBD_PRIO 0            // 2: every wave at priority 1 while it issues MFMAs
#endif
#ifndef RT_RBD_ABL
#define RT_RBD_ABL 0             // 1: no epilogue arithmetic, 2: no MFMAs, 4: no B operand reads, 8: no A (low part) reads -- results wrong by construction
#endif
#ifndef RT_RBD_PF
#define RT_RBD_PF 1              // taps the operand reads run ahead of the MFMAs
#endif

// Y_SPLIT: the output is a pre-split tensor (C/8, H, pitch, [8 hi | 8 lo]) -- another tower block reads it; else fp32 (C/4, H, pitch, 4)
template <bool Y_SPLIT>
__global__ void __launch_bounds__(512) RT_WAVES_PER_EU(2) conv_s3rbd_kernel(RBArgs a) {
    using Cfg = S3RBDCfg;
    const ConvArgs& p = a.c;
    constexpr int RX = Cfg::RX, RT = Cfg::RT, GXB = Cfg::GXB, XROWB = Cfg::XROWB, GTB = Cfg::GTB, TROWB = Cfg::TROWB, PPW = Cfg::PPW;

    __shared__ __attribute__((aligned(1024))) char sX[RX * XROWB];
    __shared__ __attribute__((aligned(1024))) char sT[(RT + 2) * TROWB];       // rows 10, 11 mirror rows 0, 1: conv2's 3-row window never wraps
    __shared__ __attribute__((aligned(1024))) f32x4 sWl[2 * Cfg::WL_SLOTS];    // conv1's | conv2's low weight parts

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kg = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is1 = wv < 4;                            // conv1 wave / conv2 wave
    const int wr = wv & 3;                              // row of the step this wave computes
#ifdef RT_KERNEL_TIMING
    unsigned long long* dbgp = (p.dbg && (tid & 255) == 0) ? p.dbg + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2 + (tid >> 8)) * 16 : nullptr;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#define RBD_STAMP(i) do { if (dbgp) dbgp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RBD_STAMP(i) do { } while (0)
#endif
    RBD_STAMP(0);

    int tile = blockIdx.x;
    if (p.xcd_order) {                                  // contiguous tile range per XCD (see conv_mfma.hip.h)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int H = p.Hi, W = p.Wi;
    const int c0 = (tile % p.tiles_x) * Cfg::SW;        // first output column of the strip
    const int y0 = (tile / p.tiles_x) * a.seg;          // first output row of the segment (a.seg rows, a multiple of 4)
    const int y1 = y0 + a.seg < H ? y0 + a.seg : H;
    const int t0 = y0 - 1;                              // first intermediate row
    const int n = blockIdx.z;
    // step s: conv1 rows t0 + 4s .. + 3 (needed up to row y1), conv2 rows t0 + 4s - 5 .. - 2 (valid in [y0, y1))
    const int nstep = (y1 - t0 + 4) / 4 + 1;
    const int last1 = (y1 - t0) / 4;                    // last step with a needed conv1 row
    const int row_hi = (y1 + 1 < H - 1 ? y1 + 1 : H - 1);          // last input row the segment needs

    // ---- DMA duties: batch b = x rows t0 + 1 + 4 b .. + 3, ring slots (4 b + 4) & 15 .. + 3, as 17 pieces of 64 x 16 bytes in ring order
    // [row][group][pixel][half]; this wave moves pieces wv, wv + 8 (and wave 0 the 17th).  Per lane and piece: the global byte offset of its
    // slot in batch 0 (kBufOOB: column outside the image) and the number of batches b >= 0 in which its row is still needed.
    const buf_rsrc rs_x = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, 4));
    const unsigned gsb_x = (unsigned)p.x_cstride * 32u, rowb_x = (unsigned)p.x_pitch * 32u;
    unsigned xvo[PPW];
    int xnb[PPW];
#pragma unroll
    for (int j = 0; j < PPW; j++) {
        const int k = (wv + 8 * j) * 64 + lane;
        const int r = k / (XROWB / 16), rem = k - r * (XROWB / 16);
        const int g = rem / (GXB / 16), rem2 = rem - g * (GXB / 16);
        const int px = rem2 >> 1, lo = (rem2 & 1) != ((px >> 3) & 1);
        const int ix = c0 - 2 + px;
        const bool valid = wv + 8 * j < Cfg::NPIECE && ix >= 0 && ix < W;
        xvo[j] = valid ? (unsigned)g * gsb_x + (unsigned)(t0 + 1 + r) * rowb_x + (unsigned)ix * 32u + (lo ? 16u : 0u) : kBufOOB;
        const int d = row_hi - (t0 + 1 + r);
        xnb[j] = valid && d >= 0 ? (d >> 2) + 1 : 0;
    }
    auto issue_batch = [&](int b) __attribute__((always_inline)) {
        const unsigned so = (unsigned)(4 * b) * rowb_x;
        char* dst = sX + ((4 * b + 4) & (RX - 1)) * XROWB;
#pragma unroll
        for (int j = 0; j < PPW; j++)
            if (wv + 8 * j < Cfg::NPIECE) rbd_dma16(rs_x, dst + (wv + 8 * j) * 1024, b < xnb[j] ? xvo[j] : kBufOOB, so);
    };

    // ---- prologue: batches -1 (rows t0 - 3 .. t0, of which t0 - 1 and t0 are needed if they exist) and 0; each convolution's split
    // weights once per workgroup: low parts into sWl for good, high parts through the (still unused) t ring into registers; biases
#pragma unroll
    for (int j = 0; j < PPW; j++) {
        if (wv + 8 * j >= Cfg::NPIECE) continue;
        const int r = ((wv + 8 * j) * 64 + lane) / (XROWB / 16);
        const bool ok = xvo[j] != kBufOOB && r >= 2 && t0 - 3 + r >= 0;
        rbd_dma16(rs_x, sX + (wv + 8 * j) * 1024, ok ? xvo[j] - 4u * rowb_x : kBufOOB, 0u);
    }
    issue_batch(0);
    {
        // slab of a convolution (rt_capi.hip: pack_rbd): [chunk * 9 + tap][hi / lo][lane] 16-byte slots = 36 pieces; 9 per wave
        const buf_rsrc rs_w = make_buf(is1 ? a.w1 : p.w);
#pragma unroll
        for (int jj = 0; jj < 9; jj++) {
            const int pw = wr + 4 * jj, t = pw >> 1;
            char* dst = (pw & 1) ? reinterpret_cast<char*>(sWl + (is1 ? 0 : Cfg::WL_SLOTS)) + t * 1024 : sT + (is1 ? 0 : Cfg::WL_SLOTS * 16) + t * 1024;
            rbd_dma16(rs_w, dst, (unsigned)lane * 16u, (unsigned)pw * 1024u);
        }
    }
    // bias of this wave's convolution as the accumulator's start value: lane (pixel, kg) owns channels 8 kg + i (i < 8), 16 + 8 kg + i - 8
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
    RBD_STAMP(1);
    wait_vmem();
#ifndef HIPEMU
    // the bias loads are the only vector-memory operations the compiler knows of: using them HERE makes it wait for them here, not at
    // their first use inside the step loop -- where its vmcnt(0) would also wait for the DMA pieces the step has just issued
    asm volatile("" : "+v"(biasv));
#endif
    __syncthreads();
    f16x8 wh[18];                                       // [chunk * 9 + tap]: high parts of this wave's A operands
    {
        const f32x4* whs = reinterpret_cast<const f32x4*>(sT) + (is1 ? 0 : Cfg::WL_SLOTS) + lane;
#pragma unroll
        for (int t = 0; t < 18; t++) wh[t] = __builtin_bit_cast(f16x8, whs[t * 64]);
    }
    __syncthreads();                                    // the t ring is free for conv1's first rows
    RBD_STAMP(2);

    const f32x4* wlp = sWl + (is1 ? 0 : Cfg::WL_SLOTS) + lane;
    // B operand of (row r of the window, column shift s, chunk c): ring row + c * 2 groups + this lane's (k-group, pixel l31 + s, hi | lo)
    int bo_h[3], bo_l[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        bo_h[s] = kg * (is1 ? GXB : GTB) + (l31 + s) * 32 + rbd_swz(l31 + s);
        bo_l[s] = bo_h[s] ^ 16;
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // 9 taps x 2 chunks of the 3-row window whose first row sits in ring slot `first`; the operands of tap t + 1 are fetched before the
    // MFMAs of tap t (scheduling barriers pin that order: left alone the scheduler sinks each read to its use, conv_rbs.hip.h)
    auto contract = [&](auto ring1, int first, f32x16& acc_m, f32x16& acc_c) __attribute__((always_inline)) {
        constexpr bool R1 = decltype(ring1)::value;
        constexpr int NR = R1 ? RX : RT, ROWB = R1 ? XROWB : TROWB, GB = R1 ? GXB : GTB;
        const char* ring = R1 ? sX : sT;
        // the t ring keeps copies of its rows 0, 1 behind its last row (conv1's epilogue writes them twice): conv2's window is three
        // CONSECUTIVE rows, one address per (shift, half) and step, the rest immediates.  The x ring wraps (no LDS left for a mirror).
        int so[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int slot = first + r;
            so[r] = R1 ? (slot >= NR ? slot - NR : slot) * ROWB : first * ROWB + r * ROWB;
        }
        auto bh_at = [&](int t) { return *reinterpret_cast<const f16x8*>(ring + so[(t % 9) / 3] + (t / 9) * (2 * GB) + bo_h[(t % 9) % 3]); };
        auto bl_at = [&](int t) { return *reinterpret_cast<const f16x8*>(ring + so[(t % 9) / 3] + (t / 9) * (2 * GB) + bo_l[(t % 9) % 3]); };
        constexpr int PF = RT_RBD_PF;
        f16x8 bh[PF + 1], bl[PF + 1], al[PF + 1];
#pragma unroll
        for (int i = 0; i < PF; i++) { bh[i] = bh_at(i); bl[i] = bl_at(i); al[i] = __builtin_bit_cast(f16x8, wlp[i * 64]); }
#if RT_RBD_PRIO == 2
        __builtin_amdgcn_s_setprio(1);
#endif
        // (Measured and dropped, profiles/r06_rbs_dev.txt: pinning the order of a tap's three MFMAs, a third accumulator so that no
        //  dependent pair is closer than three MFMAs, reads two taps ahead, a static priority for the conv2 waves -- the matrix pipe is
        //  shared by the SIMD's two waves and fully paced; none of these moved the step.)
#pragma unroll
        for (int t = 0; t < 18; t++) {
            if (t + PF < 18) {
                if (!(RT_RBD_ABL & 4)) { bh[PF] = bh_at(t + PF); bl[PF] = bl_at(t + PF); }
                else { bh[PF] = bl[PF - 1]; bl[PF] = bh[PF - 1]; }
                if (!(RT_RBD_ABL & 8)) al[PF] = __builtin_bit_cast(f16x8, wlp[(t + PF) * 64]);
                else al[PF] = __builtin_bit_cast(f16x8, __builtin_bit_cast(f32x4, al[PF - 1]) + f32x4{1.f, 0.f, 0.f, 0.f});
            }
            __builtin_amdgcn_sched_barrier(0);
            if (RT_RBD_ABL & 2) {
                if (t == 0) { acc_m = biasv; acc_c = zero16; }
                acc_m[t & 15] += (float)bh[0][0] * (float)wh[t][0];
                acc_c[t & 15] += (float)bl[0][0] * (float)al[0][0];
            } else {
                acc_m = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], bh[0], t == 0 ? biasv : acc_m, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0], bh[0], t == 0 ? zero16 : acc_c, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], bl[0], acc_c, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < PF; i++) { bh[i] = bh[i + 1]; bl[i] = bl[i + 1]; al[i] = al[i + 1]; }
        }
#if RT_RBD_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    if (is1) {
        // ================= conv1 waves: x ring -> t ring =================
        const int gx = c0 - 1 + l31;
        const bool col_in = gx >= 0 && gx < W;
        const bool strip_edge = c0 == 0 || c0 + 31 > W;             // some t column of the strip lies outside the image
        const int tl_h = kg * GTB + l31 * 32 + rbd_swz(l31), tl_l = tl_h ^ 16;     // hi / lo slot of (group kg, pixel l31) in ring row 0
        for (int s = 0; s < nstep; s++) {
            const bool more = s + 1 <= last1;           // conv1 of step s + 1 needs batch s + 1
            if (more) issue_batch(s + 1);
            if (s <= last1) {
                // all four rows of the step are computed, needed or not (rows past y1 read zero-filled x rows and land in ring slots nobody
                // reads); rows and columns outside the image become conv2's zero padding
                const int row = t0 + 4 * s + wr;
                f32x16 acc_m, acc_c;
                contract(std::true_type(), (4 * s + wr + 2) & (RX - 1), acc_m, acc_c);    // x rows row - 1 .. row + 1: slot (iy - t0 + 3) & 15
                if (s == 1 || s == 2) RBD_STAMP(4 * s);
                const bool row_in = row >= 0 && row < H;
                char* trow = sT + ((4 * s + wr) % RT) * TROWB;
                u32x4_t hh[2], ll[2];
#pragma unroll
                for (int g = 0; g < 2; g++) {
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = (RT_RBD_ABL & 1) ? ((e & 1) ? acc_c[8 * g + 4 * q + e] : acc_m[8 * g + 4 * q + e]) : rbd_elu(fmaf(acc_c[8 * g + 4 * q + e], kSplitInv, acc_m[8 * g + 4 * q + e]));
                        u32x2_t h2, l2;
                        if (RT_RBD_ABL & 1) { h2 = u32x2_t{__builtin_bit_cast(unsigned, o[0]), __builtin_bit_cast(unsigned, o[1])}; l2 = u32x2_t{__builtin_bit_cast(unsigned, o[2]), __builtin_bit_cast(unsigned, o[3])}; }
                        else { const S3Split sp = s3_split(o); h2 = __builtin_bit_cast(u32x2_t, sp.hi); l2 = __builtin_bit_cast(u32x2_t, sp.lo); }
                        hh[g][2 * q] = h2[0]; hh[g][2 * q + 1] = h2[1];
                        ll[g][2 * q] = l2[0]; ll[g][2 * q + 1] = l2[1];
                    }
                }
                if (!row_in || strip_edge) {            // wave-uniform: interior strips and rows skip the masks
                    const unsigned m = (row_in && col_in) ? 0xffffffffu : 0u;
#pragma unroll
                    for (int g = 0; g < 2; g++)
#pragma unroll
                        for (int e = 0; e < 4; e++) { hh[g][e] &= m; ll[g][e] &= m; }
                }
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    *reinterpret_cast<u32x4_t*>(trow + g * (2 * GTB) + tl_h) = hh[g];
                    *reinterpret_cast<u32x4_t*>(trow + g * (2 * GTB) + tl_l) = ll[g];
                }
                if ((4 * s + wr) % RT < 2) {            // wave-uniform: the mirror of ring rows 0, 1
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        *reinterpret_cast<u32x4_t*>(trow + RT * TROWB + g * (2 * GTB) + tl_h) = hh[g];
                        *reinterpret_cast<u32x4_t*>(trow + RT * TROWB + g * (2 * GTB) + tl_l) = ll[g];
                    }
                }
            }
            if (s == 1 || s == 2) RBD_STAMP(4 * s + 1);
            wait_vmem();
            if (s == 1 || s == 2) RBD_STAMP(4 * s + 2);
            lds_barrier();
            if (s == 0) RBD_STAMP(3);
            else if (s == 1 || s == 2) RBD_STAMP(4 * s + 3);
        }
    } else {
        // ================= conv2 waves: t ring -> y =================
        // The epilogue of a row runs at the START of the following step, while the conv1 wave of the same SIMD issues its MFMAs (and
        // conv1's epilogue runs under conv2's MFMAs): skip connection, activation, split, 16-byte stores.
        const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride + p.y_off, 4));
        const unsigned cs_y = (unsigned)p.y_cstride;
        f32x16 acc_m, acc_c;
        u32x4_t sk[4];                                  // skip connection: [group kg: hi, lo | group 2 + kg: hi, lo]
        int prow = -1;                                  // output row whose accumulators are pending
        const int ox = c0 + l31;
        const bool col_ok = l31 < Cfg::SW && ox < W;
        const int xl_h = kg * GXB + (l31 + 2) * 32 + rbd_swz(l31 + 2), xl_l = xl_h ^ 16;      // hi / lo slot of (group kg, column ox) in ring row 0
        auto epilogue2 = [&]() __attribute__((always_inline)) {
            const bool live = col_ok && prow >= 0;
            f32x4 o[4];
            if (RT_RBD_ABL & 1) {
#pragma unroll
                for (int q = 0; q < 4; q++) { o[q] = f32x4{acc_m[4 * q], acc_m[4 * q + 1], acc_c[4 * q + 2], acc_c[4 * q + 3]}; o[q][0] += __builtin_bit_cast(float, sk[q][0]); }
            } else
#pragma unroll
            for (int g = 0; g < 2; g++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const unsigned h2 = sk[2 * g][e], l2 = sk[2 * g + 1][e];
                    const int i = 8 * g + 2 * e;
                    o[2 * g + (e >> 1)][2 * (e & 1)] = rbd_elu(fmaf(acc_c[i], kSplitInv, acc_m[i]) + rbd_join<0>(h2, l2, kSplitInv));
                    o[2 * g + (e >> 1)][2 * (e & 1) + 1] = rbd_elu(fmaf(acc_c[i + 1], kSplitInv, acc_m[i + 1]) + rbd_join<1>(h2, l2, kSplitInv));
                }
            if (Y_SPLIT && (RT_RBD_ABL & 1)) {
                const unsigned vo = live ? (unsigned)(prow * p.y_ystride + ox) * 32u + (unsigned)kg * cs_y * 32u : kBufOOB;
#pragma unroll
                for (int q = 0; q < 4; q++) buf_store4(o[q], rs_y, vo, (unsigned)(2 * (q >> 1)) * cs_y * 32u + 16u * (q & 1));
            } else if (Y_SPLIT) {
                const unsigned vo = live ? (unsigned)(prow * p.y_ystride + ox) * 32u + (unsigned)kg * cs_y * 32u : kBufOOB;
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const S3Split s0 = s3_split(o[2 * g]), s1 = s3_split(o[2 * g + 1]);
                    const u32x2_t h0 = __builtin_bit_cast(u32x2_t, s0.hi), h1 = __builtin_bit_cast(u32x2_t, s1.hi);
                    const u32x2_t l0 = __builtin_bit_cast(u32x2_t, s0.lo), l1 = __builtin_bit_cast(u32x2_t, s1.lo);
                    buf_store4(__builtin_bit_cast(f32x4, u32x4_t{h0[0], h0[1], h1[0], h1[1]}), rs_y, vo, (unsigned)(2 * g) * cs_y * 32u);
                    buf_store4(__builtin_bit_cast(f32x4, u32x4_t{l0[0], l0[1], l1[0], l1[1]}), rs_y, vo, (unsigned)(2 * g) * cs_y * 32u + 16u);
                }
            } else {
                // fp32 groups of 4 channels: 8 kg + 4 q' (q' = 0, 1) and 16 + 8 kg + 4 q'
                const unsigned vo = live ? (unsigned)(prow * p.y_ystride + ox) * 16u + (unsigned)(2 * kg) * cs_y * 16u : kBufOOB;
#pragma unroll
                for (int q = 0; q < 4; q++) buf_store4(o[q], rs_y, vo, (unsigned)(4 * (q >> 1) + (q & 1)) * cs_y * 16u);
            }
        };
        for (int s = 0; s < nstep; s++) {
            const bool more = s + 1 <= last1;
            if (more) issue_batch(s + 1);
            if (s >= 2) epilogue2();                    // rows of step s - 1 (prow < 0: nothing is stored)
            if (s == 1 || s == 2) RBD_STAMP(4 * s);
            if (s >= 1) {
                // rows above / below the segment are computed too (garbage in, nothing stored): no divergent accumulator paths
                const int row = t0 + 4 * s - 5 + wr;
                contract(std::false_type(), (4 * s + wr + 4) % RT, acc_m, acc_c);         // t rows row - 1 .. row + 1: slot (ty - t0) % 10
                prow = (row >= y0 && row < y1) ? row : -1;
                // skip connection out of the x ring, row `row` = slot (row - t0 + 3) & 15; consumed after the barrier (the slot is
                // overwritten two steps from now)
                const char* xs = sX + ((4 * s - 2 + wr) & (RX - 1)) * XROWB;
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    sk[2 * g] = *reinterpret_cast<const u32x4_t*>(xs + g * (2 * GXB) + xl_h);
                    sk[2 * g + 1] = *reinterpret_cast<const u32x4_t*>(xs + g * (2 * GXB) + xl_l);
                }
            }
            if (s == 1 || s == 2) RBD_STAMP(4 * s + 1);
            wait_vmem();
            if (s == 1 || s == 2) RBD_STAMP(4 * s + 2);
            lds_barrier();
            if (s == 0) RBD_STAMP(3);
            else if (s == 1 || s == 2) RBD_STAMP(4 * s + 3);
        }
        epilogue2();                                    // nstep >= 2 always
    }
#ifdef RT_KERNEL_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    if (dbgp) { dbgp[14] = __builtin_amdgcn_s_memtime(); dbgp[12] = rt0; dbgp[13] = __builtin_amdgcn_s_memrealtime(); dbgp[15] = dbgp[13] - rt0; }
#endif
#undef RBD_STAMP
}

}  // namespace rt
