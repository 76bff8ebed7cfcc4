// 3x3x3 stride-2 transposed 3-D convolution with one or two output channels: the last layer of the 3-D Stereo
// DNN models (reference lib/conv3d_transpose_plugin.cpp:205-243 via cudnnConvolutionBackwardData;
// nvsmall_1025x321_net.cpp:399-408: (32,48,161,513) -> (97,1,321,1025)).  With a single output channel a
// 32-wide MFMA column block would be 97 % padding, so this is a vector-ALU kernel shaped for HBM/L1 traffic:
//
//   every output voxel o = 2m + phi (phi in {0,1}^3) of a 2x2x2 block reads inputs from the SAME 2x2x2 input
//   neighbourhood of m, so one thread owns a block: 8 loads per input channel feed all 8 outputs (the per-phase
//   form loads 27 values for them), the 64 phase x neighbour weights of a channel are wave-uniform and come
//   through the scalar cache (zero where a phase has no tap), and the two x-phases of a row are stored as one
//   8-byte access.  Addressing as in conv_mfma.hip.h: per-lane byte offsets computed once (out-of-range = 2^31,
//   the buffer range check returns zeros), channel stride in an SGPR.
#pragma once
#include "common.hip.h"
#include "conv_split.hip.h"

namespace rt {

struct Deconv3dSmallArgs {
    const float* x;        // (N, K, Dy, Hy, Wy)
    float* y;              // (N, Dx, C, Hx, Wx)
    const float* w;        // packed [K][COUT][phase 8][neighbour 8], phase = 4*pz + 2*py + px, neighbour = 4*jz + 2*jy + jx
    const float* bias;     // [C], padded
    const float* resid;    // like y, or nullptr
    int K, Dy, Hy, Wy;
    int Dx, Hx, Wx, C;
    int bz, by, bx;        // neighbourhood origin relative to m (0 or -1 per dimension)
    int Mz;                // 2x2x2 blocks along z
    int xp, yp;            // row pitch (elements) of the input / output planes (2-D plans may be re-pitched)
    int act;
    int64_t x_bstride, y_bstride;
    int sparse;            // 0: weights packed [K][COUT][phase][neighbour]; 1 + PAT: [K][COUT][27] (2-D: 9), only the pairs that carry a tap (SmallTaps<Z, PAT>)
};

// Which neighbour feeds which output phase.  A 3-tap stride-2 transposed convolution gives, per dimension, one phase a single tap and
// the other phase two.  With pad 1 (origin b = 0; H and W of every network of the reference, TensorFlow SAME) an even output 2m reads
// input m only and an odd one m and m+1: phase 0 -> neighbour 0, phase 1 -> both ("A").  With pad 0 (b = -1; the depth axis of the
// 3-D models' last layer, whose surplus slice the Slice plugin drops) an even output reads m-1 and m, an odd one m only: phase 0 ->
// both, phase 1 -> neighbour 1 ("B", bit d of PAT).  Either way 27 of the 64 (phase, neighbour) products of a 2x2x2 block carry a
// weight (9 of 16 in 2-D); the plan recognises the pattern in the packed weights and drops the structural zeros: 2.4x fewer FMAs and
// scalar weight loads.
template <bool Z, int PAT = 0>
struct SmallTaps {
    static constexpr int NJ = Z ? 8 : 4, ND = Z ? 3 : 2;
    static constexpr bool valid(int f, int j) {
        for (int d = 0; d < ND; d++) {
            const int fd = (f >> d) & 1, jd = (j >> d) & 1;
            if ((PAT >> d) & 1) { if (fd == 1 && jd == 0) return false; }
            else if (fd == 0 && jd == 1) return false;
        }
        return true;
    }
    static constexpr int index(int f, int j) {      // position of (f, j) among the valid pairs, phase-major
        int n = 0;
        for (int ff = 0; ff < NJ; ff++)
            for (int jj = 0; jj < NJ; jj++) {
                if (ff == f && jj == j) return n;
                n += valid(ff, jj) ? 1 : 0;
            }
        return n;
    }
    static constexpr int NV = Z ? 27 : 9;
};

// Z = false: the 2-D form (TensorRT addDeconvolution 3x3 stride 2, resnet18_2D_513x257_net.cpp:758-763): Dy = Dx = 1,
// 2x2 output blocks, weights packed [K][COUT][phase 4][neighbour 4].
template <int COUT, bool Z = true, typename TIN = float, typename TOUT = float>
__global__ void __launch_bounds__(256) deconv3d_s2_small_kernel(Deconv3dSmallArgs p) {
    constexpr int NJ = Z ? 8 : 4;          // neighbours = phases per block
    constexpr unsigned ESX = Io<TIN>::ES, ESY = Io<TOUT>::ES;
    // (depth blocks fastest per XCD, as deconv3d_s2_il_kernel orders them, was measured on this kernel in round 4: 0.402 ms either way
    //  for NVSmall's fp32 last layer -- it is bound by its per-channel 4-byte loads, not by where they are served from)
    const int mx = blockIdx.x * 256 + threadIdx.x;
    const int my = blockIdx.y;
    const int mz = blockIdx.z % p.Mz, n = blockIdx.z / p.Mz;

    // the 8 neighbours of this block in one input channel
    unsigned voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int iz = Z ? mz + p.bz + (j >> 2) : 0, iy = my + p.by + ((j >> 1) & 1), ix = mx + p.bx + (j & 1);
        const bool ok = iz >= 0 && iz < p.Dy && iy >= 0 && iy < p.Hy && ix >= 0 && ix < p.Wy;
        voff[j] = ok ? (unsigned)((iz * p.Hy + iy) * p.xp + ix) * ESX : kBufOOB;
    }
    const buf_rsrc rs_x = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, ESX));
    const unsigned cstride = (unsigned)(p.Dy * p.Hy * p.xp) * ESX;

    float acc[COUT][NJ];
#pragma unroll
    for (int co = 0; co < COUT; co++)
#pragma unroll
        for (int f = 0; f < NJ; f++) acc[co][f] = p.bias[co];

    const float* __restrict__ wk = p.w;
    auto contract_sparse = [&](auto taps) {   // the structural zeros are neither stored nor multiplied
        using Taps = decltype(taps);
        for (int k = 0; k < p.K; k++, wk += COUT * Taps::NV) {
            float xv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) xv[j] = Io<TIN>::load(rs_x, voff[j], (unsigned)k * cstride);
#pragma unroll
            for (int co = 0; co < COUT; co++)
#pragma unroll
                for (int f = 0; f < NJ; f++)
#pragma unroll
                    for (int j = 0; j < NJ; j++)
                        if (Taps::valid(f, j)) acc[co][f] = fmaf(xv[j], wk[co * Taps::NV + Taps::index(f, j)], acc[co][f]);
        }
    };
    if (p.sparse == 1) contract_sparse(SmallTaps<Z, 0>{});                       // wave-uniform
    else if (Z && p.sparse == 1 + 4) contract_sparse(SmallTaps<Z, Z ? 4 : 0>{});    // depth axis with pad 0
    else {
        for (int k = 0; k < p.K; k++, wk += COUT * NJ * NJ) {
            float xv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) xv[j] = Io<TIN>::load(rs_x, voff[j], (unsigned)k * cstride);
#pragma unroll
            for (int co = 0; co < COUT; co++)
#pragma unroll
                for (int f = 0; f < NJ; f++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) acc[co][f] = fmaf(xv[j], wk[(co * NJ + f) * NJ + j], acc[co][f]);
        }
    }

    // outputs (2mz + pz, co, 2my + py, 2mx + {0,1})
    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride, ESY));
    const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, (int64_t)n * p.y_bstride, ESY), p.resid != nullptr);
    const int ox = 2 * mx;
#pragma unroll
    for (int co = 0; co < COUT; co++) {
        if (co >= p.C) break;
#pragma unroll
        for (int pz = 0; pz < (Z ? 2 : 1); pz++)
#pragma unroll
            for (int py = 0; py < 2; py++) {
                const int oz = Z ? 2 * mz + pz : 0, oy = 2 * my + py;
                const bool row_ok = oz < p.Dx && oy < p.Hx;
                const unsigned off = (unsigned)(((oz * p.C + co) * p.Hx + oy) * p.yp + ox) * ESY;
                const unsigned v2 = (row_ok && ox + 1 < p.Wx) ? off : kBufOOB;        // both x-phases inside
                const unsigned v1 = (row_ok && ox + 1 == p.Wx) ? off : kBufOOB;       // only the even one (odd Wx)
                f32x2_t o = {acc[co][4 * pz + 2 * py], acc[co][4 * pz + 2 * py + 1]};
                const f32x2_t r2 = Io<TOUT>::load2(rs_r, v2, 0);
                const float r1 = Io<TOUT>::load(rs_r, v1, 0);
                o[0] = apply_act_rt(o[0] + r2[0] + r1, p.act);
                o[1] = apply_act_rt(o[1] + r2[1], p.act);
                Io<TOUT>::store2(o, rs_y, v2, 0);
                Io<TOUT>::store(o[0], rs_y, v1, 0);
            }
    }
}

// ---- the same layer on the matrix cores, channel-interleaved input -------------------------------------------------------------------
// half2 mode of the 3-D models stores the layer's input as (K/8, Dy, Hy, Wy, 8) fp16 (rt_conv_plan_set_layouts): the 8 channels of a
// group at one voxel are ONE 16-byte slot.  Written as a GEMM per 2x2x2 output block, out[phase f][block] = sum_{j, k} W[f][j][k] *
// x[k][neighbour j of block], the contraction index of one v_mfma_f32_16x16x32_f16 is 32 input channels at a fixed neighbour j:
//   B operand (32 k x 16 blocks): lane (n = l % 16, g = l / 16) needs channels 8g .. 8g+7 of block n's neighbour j -- exactly one slot,
//                                 one 16-byte load, no LDS, no fp16 -> fp32 conversion, no per-channel 2-byte loads (the vector-ALU form
//                                 above issues 8 x K of those per block and was bound by the L1 / TA path at 0.12 of the HBM roof);
//   A operand (16 rows x 32 k):   rows = co * 8 + phase (one or two output channels), constant per (j, channel block): 4 VGPRs each,
//                                 packed by the plan in the MFMA's lane order (rt_capi.hip) and held in registers;
//   D (16 x 16):                  lane (n, q = l / 16) holds rows 4q .. 4q+3 = (co = q / 2, fz = q % 2, fy, fx) of block n: the two
//                                 x-phases of an output row are adjacent, stored as one 8-byte access; 16 lanes cover 128 bytes.
// A wave owns 16 consecutive blocks along x and kSmallIlIters such groups; 8 MFMAs (K = 32) per group.
constexpr int kSmallIlIters = 3;
constexpr float kSaLog2e = 1.44269504088896341f;      // deconv3d_s2_ilw_kernel<SA>: the softmax runs on exp2
typedef _Float16 f16x8_small __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) deconv3d_s2_il_kernel(Deconv3dSmallArgs p) {
    const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grid.x = x-groups-of-workgroups * Mz with the depth block running fastest and a contiguous range per XCD (as conv_mfma.hip.h:
    // RT_WG_TILE): the workgroups an XCD runs side by side are depth neighbours, so the input slice two of them share comes from its L2
    int lin = blockIdx.x;
    {
        const int nwg = gridDim.x, q_ = nwg >> 3, r_ = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lin = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + idx;
    }
    const int mz = lin % p.Mz, bx = lin / p.Mz;
    const int my0 = 2 * blockIdx.y, n = blockIdx.z;              // a wave works on TWO block rows: they share the input row between them
    const int KC = p.K / 32;
    const int ngroups = (int)cdiv((p.Wx + 1) / 2, 16);
    const buf_rsrc rs_x = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, 2));
    const buf_rsrc rs_w = make_buf(p.w);
    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride, 4));
    const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, (int64_t)n * p.y_bstride, 4), p.resid != nullptr);
    const unsigned gstride = (unsigned)(p.Dy * p.Hy * p.xp) * 16u;           // bytes between channel groups of the input
    const int co = q >> 1, fz = q & 1;
    const float bias = co < p.C ? p.bias[co] : 0.f;
    const bool has_r = p.resid != nullptr;                        // uniform
    const int nrows = (p.Hx + 1) / 2;                              // block rows

    // K = 32 (every network of the reference): the 8 weight operands live in registers for the whole workgroup
    f32x4 a0[8];
    if (KC == 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) a0[j] = buf_load4(rs_w, (unsigned)(j * 64 + lane) * 16u, 0);
    }
    for (int it = 0; it < kSmallIlIters; it++) {
        const int grp = (bx * 4 + wv) * kSmallIlIters + it;                    // wave-uniform
        if (grp >= ngroups) break;
        const int mx = grp * 16 + n16;
        f32x4 acc[2] = {{bias, bias, bias, bias}, {bias, bias, bias, bias}};
        for (int kc = 0; kc < KC; kc++) {
            // neighbours (jz, row, jx): rows my0 + by + {0, 1, 2}; block row 0 uses rows 0, 1, block row 1 rows 1, 2
            f32x4 b[2][3][2];
#pragma unroll
            for (int jz = 0; jz < 2; jz++)
#pragma unroll
                for (int ry = 0; ry < 3; ry++)
#pragma unroll
                    for (int jx = 0; jx < 2; jx++) {
                        const int iz = mz + p.bz + jz, iy = my0 + p.by + ry, ix = mx + p.bx + jx;
                        const bool ok = iz >= 0 && iz < p.Dy && iy >= 0 && iy < p.Hy && ix >= 0 && ix < p.Wy;
                        const unsigned vo = ok ? (unsigned)((iz * p.Hy + iy) * p.xp + ix) * 16u + (unsigned)(kc * 4 + q) * gstride : kBufOOB;
                        b[jz][ry][jx] = buf_load4(rs_x, vo, 0);
                    }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const f32x4 av = KC == 1 ? a0[j] : buf_load4(rs_w, (unsigned)((j * KC + kc) * 64 + lane) * 16u, 0);      // L1 / L2 resident
                const f16x8_small a = __builtin_bit_cast(f16x8_small, av);
#pragma unroll
                for (int row = 0; row < 2; row++)
                    acc[row] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(f16x8_small, b[j >> 2][row + ((j >> 1) & 1)][j & 1]), acc[row], 0, 0, 0);
            }
        }
        // rows 4q + r of the MFMA tile: output channel co, depth phase fz, (fy, fx) = (r >> 1, r & 1)
        const int oz = 2 * mz + fz, ox = 2 * mx;
#pragma unroll
        for (int row = 0; row < 2; row++) {
            if (my0 + row >= nrows) break;                         // uniform
#pragma unroll
            for (int fy = 0; fy < 2; fy++) {
                const int oy = 2 * (my0 + row) + fy;
                const bool row_ok = co < p.C && oz < p.Dx && oy < p.Hx;
                const unsigned off = (unsigned)(((oz * p.C + co) * p.Hx + oy) * p.yp + ox) * 4u;
                const unsigned v2 = (row_ok && ox + 1 < p.Wx) ? off : kBufOOB;        // both x-phases inside
                const unsigned v1 = (row_ok && ox + 1 == p.Wx) ? off : kBufOOB;       // only the even one (odd Wx)
                f32x2_t o = {acc[row][2 * fy], acc[row][2 * fy + 1]};
                if (has_r) {
                    const f32x2_t r2 = buf_load2(rs_r, v2, 0);
                    const float r1 = buf_load(rs_r, v1, 0);
                    o[0] += r2[0] + r1;
                    o[1] += r2[1];
                }
                o[0] = apply_act_rt(o[0], p.act);
                o[1] = apply_act_rt(o[1], p.act);
                buf_store2(o, rs_y, v2, 0);
                buf_store(o[0], rs_y, v1, 0);
            }
        }
    }
}

// ---- the same kernel WALKING DOWN THE DEPTH AXIS (round 5; K = 32) -----------------------------------------------------------------------
// deconv3d_s2_il_kernel gives a wave three 16-block groups of one depth block and lets it go: 131 k wave-lives per pair for NVSmall's last
// layer, each with its own prologue (weight operands, address set-up), every input slice requested by two depth blocks: 2.5-2.9 TB/s of
// algorithmic bytes (profiles/r04_traffic_3d.json: 11 % MFMA-busy, nothing else busy either).  Here a wave keeps its 16 blocks x 2 block rows
// and walks the depth blocks of a segment: the slice two consecutive depth blocks share stays in registers (6 instead of 12 16-byte loads per
// step), the next slice is requested before the current step's MFMAs (three register sets, the loop unrolled by three so that their roles
// are static), and prologue and weight operands are paid once per walk.  Same arithmetic in the same order: bit-identical.
// Grid: x = ceil(groups / 4) * segments, y = pairs of block rows, z = samples; p.Mz = depth blocks, seg_len blocks per segment.
//
// SA != 0 (1 = soft-argmax, 2 = soft-argmin; C = 1, one segment, no residual): the walk IS the reduction axis of the soft-argmax that follows
// the last layer of the 3-D models (disp_softargmax, softargmax_plugin.cpp:167-205), so the (Dx, 1, Hx, Wx) fp32 volume is never written:
// every lane keeps an online softmax (running maximum, sum of weights, weighted sum of the slice index) for its 2 x 2 x 2 outputs of one
// depth parity, the two parities are merged with one lane exchange at the end, and p.y receives the (1, Hx, Wx) map (plain pitch Wx,
// p.y_bstride = Hx * Wx).  NVSmall 1025 x 321: 127 MB per pair less written and read again.
template <int SA>
__global__ void __launch_bounds__(256) RT_WAVES_PER_EU(3) deconv3d_s2_ilw_kernel(Deconv3dSmallArgs p, int seg_len, int nseg) {
    const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (An XCD-aware order of the workgroups -- each XCD a contiguous run of the logical grid, so that neighbouring row pairs share their
    //  common input row in one L2 -- was measured in round 5: deconv3D_3 of NVSmall at batch 8 0.516 -> 0.529 ms.  Not kept.)
    const int seg = blockIdx.x % nseg, bxg = blockIdx.x / nseg;
    const int my0 = 2 * blockIdx.y, n = blockIdx.z;
    const int ngroups = (int)cdiv((p.Wx + 1) / 2, 16);
    const int grp = bxg * 4 + wv;                                  // wave-uniform
    if (grp >= ngroups) return;
    const int z0 = seg * seg_len, z1 = z0 + seg_len < p.Mz ? z0 + seg_len : p.Mz;
    const buf_rsrc rs_x = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, 2));
    const buf_rsrc rs_w = make_buf(p.w);
    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride, 4));
    const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, (int64_t)n * p.y_bstride, 4), p.resid != nullptr);
    const unsigned gstride = (unsigned)(p.Dy * p.Hy * p.xp) * 16u;           // bytes between channel groups of the input
    const int co = q >> 1, fz = q & 1;
    const float bias = co < p.C ? p.bias[co] : 0.f;
    const bool has_r = p.resid != nullptr;                        // uniform
    const int nrows = (p.Hx + 1) / 2;
    const int mx = grp * 16 + n16;

    f32x4 a0[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a0[j] = buf_load4(rs_w, (unsigned)(j * 64 + lane) * 16u, 0);

    // in-plane part of the six positions (rows my0 + by + {0, 1, 2}, columns mx + bx + {0, 1}) of this lane's channel group; the slice is a
    // wave-uniform scalar offset
    unsigned pos[3][2];
#pragma unroll
    for (int ry = 0; ry < 3; ry++)
#pragma unroll
        for (int jx = 0; jx < 2; jx++) {
            const int iy = my0 + p.by + ry, ix = mx + p.bx + jx;
            pos[ry][jx] = (iy >= 0 && iy < p.Hy && ix >= 0 && ix < p.Wy) ? (unsigned)(iy * p.xp + ix) * 16u + (unsigned)q * gstride : kBufOOB;
        }
    const unsigned zbytes = (unsigned)(p.Hy * p.xp) * 16u;
    auto load_slice = [&](f32x4 (&b)[3][2], int iz) __attribute__((always_inline)) {
        const bool ok = iz >= 0 && iz < p.Dy;                     // uniform
#pragma unroll
        for (int ry = 0; ry < 3; ry++)
#pragma unroll
            for (int jx = 0; jx < 2; jx++) b[ry][jx] = buf_load4(rs_x, ok ? pos[ry][jx] : kBufOOB, ok ? (unsigned)iz * zbytes : 0u);
    };
    // (SA) running maximum (a finite start: exp2(start - anything) = 0 without an inf - inf), sum of weights, weighted index sum
    float sa_m[2][4], sa_s[2][4], sa_w[2][4];
#pragma unroll
    for (int row = 0; row < 2; row++)
#pragma unroll
        for (int j = 0; j < 4; j++) { sa_m[row][j] = -1e30f; sa_s[row][j] = 0.f; sa_w[row][j] = 0.f; }
    auto block = [&](int mz, const f32x4 (&b0)[3][2], const f32x4 (&b1)[3][2]) __attribute__((always_inline)) {
        f32x4 acc[2] = {{bias, bias, bias, bias}, {bias, bias, bias, bias}};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const f16x8_small a = __builtin_bit_cast(f16x8_small, a0[j]);
#pragma unroll
            for (int row = 0; row < 2; row++) {
                const f32x4 bv = (j >> 2) ? b1[row + ((j >> 1) & 1)][j & 1] : b0[row + ((j >> 1) & 1)][j & 1];
                acc[row] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(f16x8_small, bv), acc[row], 0, 0, 0);
            }
        }
        const int oz = 2 * mz + fz, ox = 2 * mx;
        if constexpr (SA != 0) {
            // online softmax over the slices this lane sees (oz = fz, fz + 2, ...): exp2 on pre-scaled arguments; a slice beyond the kept
            // depth enters as -inf (weight 0)
            const bool z_ok = oz < p.Dx;
            const float fzv = (float)oz;
            // (one uniform branch per step: `apply_act_rt(x, p.act)` on the run-time activation is a scalar branch tree per VALUE -- 50 branches
            //  per step in the ISA -- and the layer the reference's networks put here has no activation)
            if (p.act != 0) {
#pragma unroll
                for (int row = 0; row < 2; row++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[row][j] = apply_act_rt(acc[row][j], p.act);
            }
#pragma unroll
            for (int row = 0; row < 2; row++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float v = acc[row][j] * (SA == 2 ? -kSaLog2e : kSaLog2e);
                    const float xv = z_ok ? v : -INFINITY;
                    const float mn = fmaxf(sa_m[row][j], xv);
                    const float c = __builtin_amdgcn_exp2f(sa_m[row][j] - mn), e = __builtin_amdgcn_exp2f(xv - mn);
                    sa_s[row][j] = sa_s[row][j] * c + e;
                    sa_w[row][j] = sa_w[row][j] * c + e * fzv;
                    sa_m[row][j] = mn;
                }
        } else {
#pragma unroll
            for (int row = 0; row < 2; row++) {
                if (my0 + row >= nrows) break;                         // uniform
#pragma unroll
                for (int fy = 0; fy < 2; fy++) {
                    const int oy = 2 * (my0 + row) + fy;
                    const bool row_ok = co < p.C && oz < p.Dx && oy < p.Hx;
                    const unsigned off = (unsigned)(((oz * p.C + co) * p.Hx + oy) * p.yp + ox) * 4u;
                    const unsigned v2 = (row_ok && ox + 1 < p.Wx) ? off : kBufOOB;        // both x-phases inside
                    const unsigned v1 = (row_ok && ox + 1 == p.Wx) ? off : kBufOOB;       // only the even one (odd Wx)
                    f32x2_t o = {acc[row][2 * fy], acc[row][2 * fy + 1]};
                    if (has_r) {
                        const f32x2_t r2 = buf_load2(rs_r, v2, 0);
                        const float r1 = buf_load(rs_r, v1, 0);
                        o[0] += r2[0] + r1;
                        o[1] += r2[1];
                    }
                    if (p.act != 0) {
                        o[0] = apply_act_rt(o[0], p.act);
                        o[1] = apply_act_rt(o[1], p.act);
                    }
                    buf_store2(o, rs_y, v2, 0);
                    buf_store(o[0], rs_y, v1, 0);
                }
            }
        }
    };

    f32x4 s0[3][2], s1[3][2], s2[3][2];
    load_slice(s0, z0 + p.bz);
    load_slice(s1, z0 + p.bz + 1);
    for (int mz = z0; mz < z1; mz += 3) {
        load_slice(s2, mz + p.bz + 2);             // (beyond the segment / the volume: zeros, never used)
        block(mz, s0, s1);
        if (mz + 1 >= z1) break;
        load_slice(s0, mz + p.bz + 3);
        block(mz + 1, s1, s2);
        if (mz + 2 >= z1) break;
        load_slice(s1, mz + p.bz + 4);
        block(mz + 2, s2, s0);
    }
    if constexpr (SA != 0) {
        // merge the two depth parities (lanes q = 0 and q = 1 of the same 16 blocks), then the q = 0 lane writes its 2 x 2 x 2 pixels
        const int ox = 2 * mx;
#pragma unroll
        for (int row = 0; row < 2; row++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float m1 = __shfl_xor(sa_m[row][j], 16), s1v = __shfl_xor(sa_s[row][j], 16), w1v = __shfl_xor(sa_w[row][j], 16);
                const float mn = fmaxf(sa_m[row][j], m1);
                const float c0 = __builtin_amdgcn_exp2f(sa_m[row][j] - mn), c1 = __builtin_amdgcn_exp2f(m1 - mn);
                const float st = sa_s[row][j] * c0 + s1v * c1, wt = sa_w[row][j] * c0 + w1v * c1;
                const int oy = 2 * (my0 + row) + (j >> 1), oxj = ox + (j & 1);
                const bool ok = q == 0 && oy < p.Hx && oxj < p.Wx;
                buf_store(wt / st, rs_y, ok ? (unsigned)(oy * p.Wx + oxj) * 4u : kBufOOB, 0);
            }
    }
}

// ---- ... and for fp32 engines: input (K/4, Dy, Hy, Wy, 4) fp32, 3-term fp16 split ---------------------------------------------------------
// The vector-ALU kernel at the top reads its fp32 input with one 4-byte load per channel and neighbour (NVSmall: 0.40 ms for 254 MB in +
// 127 MB out; PMC: 2.3 GB through the L2).  With the layer's input channel-interleaved in groups of 4 (written that way by
// deconv_s3p_kernel<true>), a lane's 8 channels of a neighbour are two 16-byte slots; they are split into fp16 high / low parts in
// registers (s3_split: 5 vector instructions per pair) and multiplied as in conv_split.hip.h -- A_hi B_hi into the main accumulator,
// A_lo B_hi + A_hi B_lo into the cross accumulator (scaled by 2^11) -- so the result has fp32-class accuracy like every other layer
// of an fp32 engine.  Same tile shape, workgroup order and epilogue as deconv3d_s2_il_kernel; weights: [hi / lo][j][KC][64 lanes][8].
__global__ void __launch_bounds__(256) deconv3d_s2_il4_kernel(Deconv3dSmallArgs p) {
    const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    {
        const int nwg = gridDim.x, q_ = nwg >> 3, r_ = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lin = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + idx;
    }
    const int mz = lin % p.Mz, bx = lin / p.Mz;
    const int my0 = 2 * blockIdx.y, n = blockIdx.z;
    const int KC = p.K / 32;
    const int ngroups = (int)cdiv((p.Wx + 1) / 2, 16);
    const buf_rsrc rs_x = make_buf(elem_ptr(p.x, (int64_t)n * p.x_bstride, 4));
    const buf_rsrc rs_w = make_buf(p.w);
    const buf_rsrc rs_y = make_buf(elem_ptr(p.y, (int64_t)n * p.y_bstride, 4));
    const buf_rsrc rs_r = make_buf(elem_ptr(p.resid, (int64_t)n * p.y_bstride, 4), p.resid != nullptr);
    const unsigned gstride = (unsigned)(p.Dy * p.Hy * p.xp) * 16u;           // bytes between groups of 4 channels
    const int co = q >> 1, fz = q & 1;
    const float bias = co < p.C ? p.bias[co] : 0.f;
    const bool has_r = p.resid != nullptr;
    const int nrows = (p.Hx + 1) / 2;

    f32x4 ah0[8], al0[8];                                          // K = 32: the weight operands stay in registers
    if (KC == 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            ah0[j] = buf_load4(rs_w, (unsigned)(j * 64 + lane) * 16u, 0);
            al0[j] = buf_load4(rs_w, (unsigned)((8 + j) * 64 + lane) * 16u, 0);
        }
    }
    for (int it = 0; it < kSmallIlIters; it++) {
        const int grp = (bx * 4 + wv) * kSmallIlIters + it;
        if (grp >= ngroups) break;
        const int mx = grp * 16 + n16;
        f32x4 acc_m[2] = {{bias, bias, bias, bias}, {bias, bias, bias, bias}};
        f32x4 acc_c[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int kc = 0; kc < KC; kc++) {
#pragma unroll
            for (int jz = 0; jz < 2; jz++) {                        // one depth neighbour at a time: 6 positions x (hi, lo) live
                f16x8_small bh[3][2], bl[3][2];
#pragma unroll
                for (int ry = 0; ry < 3; ry++)
#pragma unroll
                    for (int jx = 0; jx < 2; jx++) {
                        const int iz = mz + p.bz + jz, iy = my0 + p.by + ry, ix = mx + p.bx + jx;
                        const bool ok = iz >= 0 && iz < p.Dy && iy >= 0 && iy < p.Hy && ix >= 0 && ix < p.Wy;
                        const unsigned vo = (unsigned)((iz * p.Hy + iy) * p.xp + ix) * 16u + (unsigned)(kc * 8 + 2 * q) * gstride;
                        const S3Split s0 = s3_split(buf_load4(rs_x, ok ? vo : kBufOOB, 0));
                        const S3Split s1 = s3_split(buf_load4(rs_x, ok ? vo + gstride : kBufOOB, 0));
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            bh[ry][jx][e] = s0.hi[e]; bh[ry][jx][4 + e] = s1.hi[e];
                            bl[ry][jx][e] = s0.lo[e]; bl[ry][jx][4 + e] = s1.lo[e];
                        }
                    }
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int j = 4 * jz + jj;
                    const f32x4 avh = KC == 1 ? ah0[j] : buf_load4(rs_w, (unsigned)((j * KC + kc) * 64 + lane) * 16u, 0);
                    const f32x4 avl = KC == 1 ? al0[j] : buf_load4(rs_w, (unsigned)(((8 + j) * KC + kc) * 64 + lane) * 16u, 0);
                    const f16x8_small ah = __builtin_bit_cast(f16x8_small, avh), al = __builtin_bit_cast(f16x8_small, avl);
#pragma unroll
                    for (int row = 0; row < 2; row++) {
                        const f16x8_small h = bh[row + ((jj >> 1) & 1)][jj & 1], l = bl[row + ((jj >> 1) & 1)][jj & 1];
                        acc_m[row] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, h, acc_m[row], 0, 0, 0);
                        acc_c[row] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, h, acc_c[row], 0, 0, 0);
                        acc_c[row] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, l, acc_c[row], 0, 0, 0);
                    }
                }
            }
        }
        const int oz = 2 * mz + fz, ox = 2 * mx;
#pragma unroll
        for (int row = 0; row < 2; row++) {
            if (my0 + row >= nrows) break;                         // uniform
#pragma unroll
            for (int fy = 0; fy < 2; fy++) {
                const int oy = 2 * (my0 + row) + fy;
                const bool row_ok = co < p.C && oz < p.Dx && oy < p.Hx;
                const unsigned off = (unsigned)(((oz * p.C + co) * p.Hx + oy) * p.yp + ox) * 4u;
                const unsigned v2 = (row_ok && ox + 1 < p.Wx) ? off : kBufOOB;
                const unsigned v1 = (row_ok && ox + 1 == p.Wx) ? off : kBufOOB;
                f32x2_t o = {fmaf(acc_c[row][2 * fy], kSplitInv, acc_m[row][2 * fy]), fmaf(acc_c[row][2 * fy + 1], kSplitInv, acc_m[row][2 * fy + 1])};
                if (has_r) {
                    const f32x2_t r2 = buf_load2(rs_r, v2, 0);
                    const float r1 = buf_load(rs_r, v1, 0);
                    o[0] += r2[0] + r1;
                    o[1] += r2[1];
                }
                o[0] = apply_act_rt(o[0], p.act);
                o[1] = apply_act_rt(o[1], p.act);
                buf_store2(o, rs_y, v2, 0);
                buf_store(o[0], rs_y, v1, 0);
            }
        }
    }
}

}  // namespace rt
