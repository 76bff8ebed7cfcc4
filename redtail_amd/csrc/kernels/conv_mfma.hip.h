// Implicit-GEMM convolution on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// One kernel serves every dense contraction of the Stereo DNN hot path:
//   * TensorRT addConvolution 3x3 s1/s2, 5x5 s2          (reference: resnet18_2D_513x257_net.cpp:48-719)
//   * TensorRT addDeconvolution 3x3 s2 (4 output phases) (reference: resnet18_2D_513x257_net.cpp:722-763)
//   * Conv3DPlugin, TF semantics        (reference: lib/conv3d_plugin.cpp:187-216, conv_utils.cpp:27-72)
//   * Conv3DTransposePlugin             (reference: lib/conv3d_transpose_plugin.cpp:205-243)
// by expressing each as   y[co, oy, ox] = sum_{ci, r, s} x[chan(ci), oy*S + r - py, ox*S + s - px] * w[co, ci, r, s]
// over a rectangular KH x KW tap window, where `chan(ci)` is a per-z-slice table of plane offsets
// (-1 = zero plane).  The table is what turns the (D*C)-merged cuDNN trick of the reference
// (conv_utils.cpp:27-32,58-72) and its Pad/Slice plugins into plain bounds checks.
//
// GEMM view per workgroup:  M = TY x TX output pixels, N = 32*NBW output channels, K = Cin * KH*KW.
//   A operand = weights  (rows  = output channel)   -> lane l holds w[co = l&31][k = l>>5]
//   B operand = pixels   (cols  = output pixel)     -> lane l holds x[k = l>>5][px = l&31]
// so each accumulator register holds 32 consecutive output pixels of one channel across the lanes
// of a half-wave: global stores are 128-byte coalesced with no alignment requirement (W is odd in
// every network of the reference).
//
// Data flow: global -> registers (prefetch of chunk i+1 is in flight while chunk i is multiplied)
// -> LDS (input patch [CC][PR][PC] + weight slab [taps][CC][NB]) -> one ds_read_b32 per MFMA operand.
// v_mfma_f32_32x32x2_f32 issues every 64 cycles per SIMD, so two LDS reads per MFMA keep the matrix
// pipe fed (MI355X_MICROARCH.md, LDS table); the kernel is MFMA-bound, roofline = 157.3 TFLOP/s.
#pragma once
#include "common.hip.h"

namespace rt {

struct ConvArgs {
    const float* x;
    float* y;
    const float* w;        // packed [nblk][chunk][tap][CC][NB]
    const float* bias;     // [Cout] or nullptr
    const float* resid;    // same addressing as y, or nullptr
    const int* ch_off;     // [nz][CinPad] plane offsets (elements) relative to the sample base, -1 = zeros
    int CinPad;            // multiple of CC
    int Cout;
    int Hi, Wi;            // input plane (row pitch = Wi)
    int Ho, Wo;            // output grid of this launch
    int pad_y, pad_x;
    int nz;                // z-slices per sample (3-D depth positions / 1 for 2-D)
    int tiles_x;           // number of tiles along x (grid.x = tiles_x * tiles_y)
    int act;
    int64_t x_bstride;     // per-sample strides (elements)
    int64_t y_bstride;
    int64_t y_cstride;     // output channel stride
    int64_t y_zstride;     // output z-slice stride
    int64_t y_off;         // constant output offset (deconv phase origin)
    int y_ystride, y_xstride;
};

template <int KH, int KW, int S, int TY, int TXW, int NBW, int CC>
struct ConvCfg {
    static constexpr int TX = 32 * TXW;
    static constexpr int NB = 32 * NBW;
    static constexpr int TAPS = KH * KW;
    static constexpr int PR = (TY - 1) * S + KH;
    static constexpr int PC = (TX - 1) * S + KW;
    static constexpr int IN_ELEMS = CC * PR * PC;
    static constexpr int W_ELEMS = TAPS * CC * NB;
    static constexpr int NK_IN = (IN_ELEMS + 255) / 256;
    static constexpr int NK_W = (W_ELEMS / 4 + 255) / 256;
    static constexpr int WT = TY * TXW / 4;              // wave-tiles (32 px) per wave
    static_assert((TY * TXW) % 4 == 0, "tile must split evenly over 4 waves");
    static_assert(CC % 2 == 0, "MFMA 32x32x2 consumes channel pairs");
    static_assert(W_ELEMS % 4 == 0, "weight slab is copied as float4");
    static constexpr size_t LDS_BYTES = (size_t)(IN_ELEMS + W_ELEMS) * 4 + 4 * 512;
};

template <int KH, int KW, int S, int TY, int TXW, int NBW, int CC>
__global__ void __launch_bounds__(256) conv_mfma_f32_kernel(ConvArgs p) {
    using Cfg = ConvCfg<KH, KW, S, TY, TXW, NBW, CC>;
    constexpr int TX = Cfg::TX, NB = Cfg::NB, PR = Cfg::PR, PC = Cfg::PC, WT = Cfg::WT;

    __shared__ float sIn[CC * PR * PC];
    __shared__ __attribute__((aligned(16))) float sW[Cfg::W_ELEMS];
    __shared__ int sOff[512];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    const int tile = blockIdx.x;
    const int tx0 = (tile % p.tiles_x) * TX;
    const int ty0 = (tile / p.tiles_x) * TY;
    const int nblk = blockIdx.y;
    const int zi = blockIdx.z % p.nz;
    const int n = blockIdx.z / p.nz;

    const float* __restrict__ xb = p.x + (int64_t)n * p.x_bstride;
    const int nchunks = p.CinPad / CC;

    // plane-offset table of this z-slice -> LDS (CinPad <= 512 is checked by the launcher)
    for (int i = tid; i < p.CinPad; i += 256) sOff[i] = p.ch_off[(int64_t)zi * p.CinPad + i];

    // per-thread, chunk-invariant part of the input gather: in-plane offset or -1
    int poff[Cfg::NK_IN];
    int pch[Cfg::NK_IN];
#pragma unroll
    for (int k = 0; k < Cfg::NK_IN; k++) {
        const int idx = tid + 256 * k;
        const int c = idx / (PR * PC);
        const int rem = idx - c * (PR * PC);
        const int pr = rem / PC, pc = rem - pr * PC;
        const int iy = ty0 * S - p.pad_y + pr;
        const int ix = tx0 * S - p.pad_x + pc;
        const bool ok = idx < Cfg::IN_ELEMS && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        poff[k] = ok ? iy * p.Wi + ix : -1;
        pch[k] = c;
    }

    const f32x4* __restrict__ wsrc =
        reinterpret_cast<const f32x4*>(p.w + ((int64_t)nblk * nchunks) * Cfg::W_ELEMS);

    float rin[Cfg::NK_IN];
    f32x4 rw[Cfg::NK_W];

    f32x16 acc[WT][NBW];
#pragma unroll
    for (int i = 0; i < WT; i++)
#pragma unroll
        for (int b = 0; b < NBW; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][b][r] = 0.f;

    __syncthreads();   // sOff visible

    auto prefetch = [&](int ch) {
#pragma unroll
        for (int k = 0; k < Cfg::NK_IN; k++) {
            float v = 0.f;
            if (poff[k] >= 0) {
                const int off = sOff[ch * CC + pch[k]];
                if (off >= 0) v = xb[(int64_t)off + poff[k]];
            }
            rin[k] = v;
        }
        const f32x4* ws = wsrc + (int64_t)ch * (Cfg::W_ELEMS / 4);
#pragma unroll
        for (int k = 0; k < Cfg::NK_W; k++) {
            const int idx = tid + 256 * k;
            rw[k] = idx < Cfg::W_ELEMS / 4 ? ws[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    prefetch(0);
    for (int ch = 0; ch < nchunks; ch++) {
        __syncthreads();   // everyone finished reading the previous chunk from LDS
#pragma unroll
        for (int k = 0; k < Cfg::NK_IN; k++) {
            const int idx = tid + 256 * k;
            if (idx < Cfg::IN_ELEMS) sIn[idx] = rin[k];
        }
#pragma unroll
        for (int k = 0; k < Cfg::NK_W; k++) {
            const int idx = tid + 256 * k;
            if (idx < Cfg::W_ELEMS / 4) reinterpret_cast<f32x4*>(sW)[idx] = rw[k];
        }
        __syncthreads();
        if (ch + 1 < nchunks) prefetch(ch + 1);   // global loads fly while the MFMAs below run

#pragma unroll
        for (int r = 0; r < KH; r++) {
#pragma unroll
            for (int s = 0; s < KW; s++) {
#pragma unroll
                for (int cp = 0; cp < CC / 2; cp++) {
                    const int c = 2 * cp + half;
                    float a[NBW];
#pragma unroll
                    for (int b = 0; b < NBW; b++) a[b] = sW[((r * KW + s) * CC + c) * NB + b * 32 + l31];
#pragma unroll
                    for (int i = 0; i < WT; i++) {
                        const int t = wave + 4 * i;
                        const int ty = t / TXW, txw = t % TXW;
                        const float bv = sIn[(c * PR + ty * S + r) * PC + (txw * 32 + l31) * S + s];
#pragma unroll
                        for (int b = 0; b < NBW; b++)
                            acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b], bv, acc[i][b], 0, 0, 0);
                    }
                }
            }
        }
    }

    // epilogue: bias (+ residual) + activation, 128-byte coalesced stores per half-wave
    const int64_t ybase = (int64_t)n * p.y_bstride + (int64_t)zi * p.y_zstride + p.y_off;
#pragma unroll
    for (int i = 0; i < WT; i++) {
        const int t = wave + 4 * i;
        const int oy = ty0 + t / TXW;
        const int ox = tx0 + (t % TXW) * 32 + l31;
        const bool pix_ok = oy < p.Ho && ox < p.Wo;
        const int64_t pbase = ybase + (int64_t)oy * p.y_ystride + (int64_t)ox * p.y_xstride;
#pragma unroll
        for (int b = 0; b < NBW; b++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = nblk * NB + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (pix_ok && co < p.Cout) {
                    const int64_t addr = pbase + (int64_t)co * p.y_cstride;
                    float v = acc[i][b][r];
                    if (p.bias) v += p.bias[co];
                    if (p.resid) v += p.resid[addr];
                    p.y[addr] = apply_act_rt(v, p.act);
                }
            }
        }
    }
}

}  // namespace rt
