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

}  // namespace rt
