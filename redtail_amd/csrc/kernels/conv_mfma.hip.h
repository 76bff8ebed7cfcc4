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
// Data flow per chunk of CC input channels:
//   global --(register prefetch: loads of chunk i+1 fly under the MFMAs of chunk i)--> LDS
//   LDS image of the patch: [row][h][col][CC/2]  with channel = 2*j + h, so ONE ds_read_b128 hands a
//   lane its B operands for the CC/2 consecutive MFMA k-steps of a tap; the weight slab is
//   [tap][h][co][CC/2], so ONE ds_read_b128 hands it the matching A operands.  Both images make
//   every ds_read_b128 lane group hit 16 distinct 16-byte slots (bank-conflict free, stride 1).
//   Per tap a wave issues (NBW + WT) wide LDS reads for 4*NBW*WT MFMAs of 64 cycles each, and the
//   reads of tap t+1 are issued before the MFMAs of tap t: the kernel is MFMA-bound,
//   roofline = 157.3 TFLOP/s (MI355X_MICROARCH.md).
#pragma once
#include <type_traits>
#include "common.hip.h"

namespace rt {

// Optional per-z-slice overrides: lets ONE launch cover the output phases of a transposed convolution
// (each phase = its own padding, output origin and weight slab) and/or depth positions with their
// own gather-table row.
struct ZSlice {
    int pad_y, pad_x;
    int Ho, Wo;
    int ch_row;            // row of the ch_off table
    unsigned tap_mask;     // bit (u*KW + v): tap (u, v) of the launch window exists in this slice (others have zero weights)
    int64_t y_off;         // replaces ConvArgs::y_off + zi * y_zstride
    int64_t w_off;         // element offset of this slice's packed weights
    int64_t r_off;         // like y_off, for the residual tensor (differs when the output layout is transposed)
    int64_t r_off_il8;     // r_off for a channel-interleaved fp16 residual (C/8, H, W, 8): the pixel part of the offset counts 8 elements
    int64_t y_off_il8;     // y_off for a channel-interleaved fp16 output: (D, C/8, H, W, 8), or (C/8, D, H, W, 8) when the layout is transposed
    int64_t r_off_il4;     // r_off for a channel-interleaved fp32 residual, groups of 4 channels: plane part + 4 * pixel part
};

struct ConvArgs {
    const float* x;
    float* y;
    const float* w;        // packed [nblk][chunk][tap][h][NB][CC/2]
    const float* bias;     // [round_up(Cout, NB)], zero padded, never null
    const float* resid;    // same addressing as y, or nullptr
    unsigned long long* dbg;   // phase-timing buffer [workgroup][16]; only read when built with -DRT_KERNEL_TIMING
    const float* zeros;    // >= 16 bytes of zeros: target of every out-of-image / padded-channel gather
    const ZSlice* zs;      // [nz] or nullptr (uniform slices)
    const int* ch_off;     // [nz][CinPad] plane offsets (elements) relative to the sample base, -1 = zeros
    int CinPad;            // multiple of CC
    int Cout;
    int Hi, Wi;            // input plane
    int x_pitch;           // input row pitch in elements (>= Wi; internal tensors may be padded to 128 B)
    int Ho, Wo;            // output grid of this launch
    int pad_y, pad_x;
    int nz;                // z-slices per sample (3-D depth positions / 1 for 2-D)
    int tiles_x;           // number of tiles along x (grid.x = tiles_x * tiles_y)
    int act;
    int xcd_order;         // 1 = contiguous tile range per XCD (see kernel), 0 = dispatch order
    int64_t x_bstride;     // per-sample strides (elements)
    int64_t y_bstride;
    int64_t y_cstride;     // output channel stride
    int64_t y_zstride;     // output z-slice stride
    int64_t y_off;         // constant output offset (deconv phase origin)
    int y_ystride, y_xstride;
    int64_t r_cstride;     // residual: channel stride and per-sample stride (== y_cstride / y_bstride unless the
    int64_t r_bstride;     // launch writes a transposed layout, e.g. Conv3DTranspose + Transform in one pass)
    int r_il8;             // conv_f16mma_kernel: the residual tensor is channel-interleaved (see conv_f16.hip.h)
    // persistent kernels (conv_split.hip.h): the grid is not the tile count
    int batch;             // samples of this launch
    int cin_real;          // input channels that exist (CinPad - zero padding)
    int64_t x_cstride;     // input channel stride (2-D plans: Hi * x_pitch)
    // conv_s3_kernel: per gathered channel a shift along x (same [nz][CinPad] shape as ch_off, or nullptr): the channel is read
    // at column ix - shift and is zero for ix < shift -- the right-image half of a folded default cost volume
    const int* ch_shift;
    int w_exact;           // conv_s3_kernel: the weights came from an fp16 file -- their low parts are zero and are not multiplied
    // 3-D launches: the z-slices are folded into grid.x with z running FASTEST (grid.x = tiles * nz, grid.z = samples).  With the per-XCD
    // contiguous ranges of xcd_order the workgroups an XCD runs side by side then belong to neighbouring depth slices (and output phases)
    // of the same image tile: the slices a 3x3x3 window shares are served by that XCD's L2 instead of being fetched three times
    // (round 4, PMC: conv3D_2 of NVSmall fetched 836 MB for a 254 MB input with z outermost).
    int z_inner;
    // > 0: the blocks of 32 output channels are folded into grid.x as well and run fastest of all -- the nb_inner workgroups that read
    // the SAME input patch follow each other on one XCD, so that patch comes from HBM once instead of once per block (round 4, PMC:
    // the stride-2 Conv3D 32 -> 64 of NVSmall half2 fetched 839 MB for a 254 MB input at 6.6 TB/s of fabric traffic).  0: grid.y.
    int nb_inner;
    // conv_f16dw_kernel / conv_s3dw_kernel (depth-walking Conv3D, conv_f16dw.hip.h): output slices per depth segment, segments per tile
    // pair, 16-channel chunks per input slice (C / 16), image tiles per sample (tiles_x * tiles_y of 12 x 32 pixels)
    int dw_seg, dw_nseg, dw_cpc, dw_ntiles;
    // first layers (conv_s3_first_kernel, conv_f16_first_kernel): samples n >= x2_from are read from x2 (sample n - x2_from) -- the left and
    // the right image of a stereo pair are two bindings, and one launch over [left samples | right samples] serves both towers' first layer
    // (rt_conv_enqueue_twin_input).  x2 == nullptr: off.
    const float* x2;
    int x2_from;
};

// workgroup -> (tile of the output plane, z-slice, sample)
#define RT_WG_TILE_NB(p, tile, zi, n, nblk)                                                         \
    int tile = blockIdx.x;                                                                          \
    if ((p).xcd_order) {                          /* contiguous range per XCD (see conv_mfma_f32_kernel) */ \
        const int nwg_ = gridDim.x, q_ = nwg_ >> 3, r_ = nwg_ & 7;                                  \
        const int xcd_ = blockIdx.x & 7, idx_ = blockIdx.x >> 3;                                    \
        tile = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + idx_;             \
    }                                                                                               \
    int nblk = blockIdx.y;                                                                          \
    if ((p).nb_inner > 0) { nblk = tile % (p).nb_inner; tile /= (p).nb_inner; }                     \
    int zi, n;                                                                                      \
    if ((p).z_inner) { zi = tile % (p).nz; tile /= (p).nz; n = blockIdx.z; }                        \
    else { zi = blockIdx.z % (p).nz; n = blockIdx.z / (p).nz; }
#define RT_WG_TILE(p, tile, zi, n) RT_WG_TILE_NB(p, tile, zi, n, nblk_unused_); (void)nblk_unused_;

// Optional in-kernel phase timing (tools/time_phases.py builds a separate library with
// -DRT_KERNEL_TIMING): thread 0 stamps s_memtime at phase boundaries.  Compiles to nothing otherwise.
#ifdef RT_KERNEL_TIMING
#define RT_TSTAMP() do { if (dbgp && tid == 0 && dbi < 16) dbgp[dbi++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RT_TSTAMP() do { } while (0)
#endif

// Ablation switches for tools/ablate_conv.py (separate instrumented builds, -DRT_ABLATE=<mask>; the product
// build has RT_ABLATE == 0 and every switch folds away).  Results are wrong by construction.
#ifndef RT_ABLATE
#define RT_ABLATE 0
#endif
constexpr bool kAblGather = (RT_ABLATE & 1) != 0;    // no global input gathers
constexpr bool kAblWLoad = (RT_ABLATE & 2) != 0;     // no global weight loads
constexpr bool kAblLdsWr = (RT_ABLATE & 4) != 0;     // no LDS writes
constexpr bool kAblBarrier = (RT_ABLATE & 8) != 0;   // no workgroup barriers
constexpr bool kAblResid = (RT_ABLATE & 16) != 0;    // no residual loads
constexpr bool kAblStore = (RT_ABLATE & 32) != 0;    // no output stores
constexpr bool kAblLdsRd = (RT_ABLATE & 64) != 0;    // MFMA operands from registers instead of LDS
constexpr bool kAblXform = (RT_ABLATE & 128) != 0;   // Winograd kernel: no input transform
constexpr bool kAblMfma = (RT_ABLATE & 256) != 0;    // conv_f16mma_kernel: no matrix instructions
__device__ __forceinline__ void wg_barrier() { if (!kAblBarrier) __syncthreads(); }

template <int N> struct VecOf;
template <> struct VecOf<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int KH, int KW, int S, int TY, int TXW, int NBW, int CC, int NW = 4, bool WLDS = true>
struct ConvCfg {
    static constexpr int TX = 32 * TXW;
    static constexpr int NB = 32 * NBW;
    static constexpr int TAPS = KH * KW;
    static constexpr int CPG = CC / 2;                    // MFMA k-steps (channel pairs) per tap per chunk
    static constexpr int PR = (TY - 1) * S + KH;
    static constexpr int PC = (TX - 1) * S + KW;
    static constexpr int NPIX = PR * PC;                  // patch pixels
    static constexpr int IN_ELEMS = CC * NPIX;
    static constexpr int W_ELEMS = TAPS * CC * NB;
    // staging roles: with 4 waves, wave w gathers channel parity w&1 for pixel half w>>1; a
    // single-wave workgroup gathers both parities of all pixels itself
    static constexpr int NPAR = NW == 1 ? 2 : 1;          // parities gathered per wave
    static constexpr int NPART = NW == 1 ? 1 : NW / 2;    // pixel partitions
    static constexpr int NKP = ((NPIX + NPART - 1) / NPART + 63) / 64;   // patch pixels per lane
    static constexpr int NTHR = 64 * NW;
    static constexpr int NK_W = (W_ELEMS / 4 + NTHR - 1) / NTHR;
    static constexpr int WT = TY * TXW / NW;              // wave-tiles (32 px) per wave
    // LDS stages.  2 = double buffer with one barrier per chunk: measured slower (b8 197 vs 179 us, 5 instead of
    // 7 waves/SIMD for the extra 15.7 KB) -- barriers are not what bounds the kernel -- so 1 everywhere.
    static constexpr int NBUF = 1;
    // one 32x32 accumulator per wave: ask the register allocator for 5 waves/SIMD (<= 100 VGPR+AGPR)
    static constexpr int MINW = (WT * NBW == 1 && NW == 4 && TY == 4) ? 5 : (WT * NBW == 1 && NW == 8 ? 4 : 1);
    static_assert(NW == 1 || NW == 2 || NW == 4 || NW == 8, "1, 2, 4 or 8 waves per workgroup");
    static_assert((TY * TXW) % NW == 0, "tile must split evenly over the waves");
    static_assert(WLDS || true, "");
    static_assert(CC == 4 || CC == 8, "CC/2 k-steps are fetched by one ds_read_b64 / ds_read_b128");
    static_assert(W_ELEMS % 4 == 0, "weight slab is copied as float4");
};

// TIN / TOUT: storage type of the input and of the output + residual (float, or _Float16 in half2 mode)
// YIL (fp32 output, no residual): the output tensor is channel-interleaved (C/4, H, pitch, 4), see conv_wino.hip.h --
// accumulator rows 4q .. 4q+3 of a lane are 4 consecutive channels of its pixel: one 16-byte store instead of four.
template <int KH, int KW, int S, int TY, int TXW, int NBW, int CC, int NW, bool WLDS, typename TIN = float, typename TOUT = float, bool YIL = false>
__global__ void __launch_bounds__(64 * NW, (ConvCfg<KH, KW, S, TY, TXW, NBW, CC, NW, WLDS>::MINW))
conv_mfma_f32_kernel(ConvArgs p) {
    static_assert(!YIL || std::is_same<TOUT, float>::value, "interleaved output: fp32");
    constexpr unsigned ESX = Io<TIN>::ES, ESY = Io<TOUT>::ES;
    using Cfg = ConvCfg<KH, KW, S, TY, TXW, NBW, CC, NW, WLDS>;
    constexpr int TX = Cfg::TX, NB = Cfg::NB, PC = Cfg::PC, WT = Cfg::WT, CPG = Cfg::CPG;
    constexpr int NPIX = Cfg::NPIX, NKP = Cfg::NKP, TAPS = Cfg::TAPS, NPAR = Cfg::NPAR, NTHR = Cfg::NTHR;
    typedef typename VecOf<CPG>::type vec_t;

    // NBUF == 2: the chunk being staged never aliases the one being read and one barrier per chunk is enough
    constexpr int NBUF = Cfg::NBUF;
    __shared__ __attribute__((aligned(16))) float sIn[NBUF][Cfg::IN_ELEMS];            // [row][h][col][CPG]
    __shared__ __attribute__((aligned(16))) float sW[NBUF][WLDS ? Cfg::W_ELEMS : 4];   // [tap][h][co][CPG]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only), each
    // XCD has its own L2, and neighbouring tiles share input halo rows and partially written cache lines.
    // Give every XCD a contiguous range of tiles (bijective for any grid size) so that sharing stays in one L2.
    int tile = blockIdx.x;
    if (p.xcd_order) {
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
    // per-slice parameters (wave-uniform): uniform launch, or one entry of the ZSlice table
    int pad_y = p.pad_y, pad_x = p.pad_x, Ho = p.Ho, Wo = p.Wo, ch_row = zi;
    int64_t y_off = p.y_off + (int64_t)zi * p.y_zstride, w_off = 0, r_off = y_off;
    unsigned tap_mask = ~0u;       // transposed-conv phases narrower than the launch window skip the padded taps
    if (p.zs) {
        const ZSlice z = p.zs[zi];
        pad_y = z.pad_y; pad_x = z.pad_x; Ho = z.Ho; Wo = z.Wo; ch_row = z.ch_row; y_off = z.y_off; w_off = z.w_off;
        r_off = z.r_off;
        tap_mask = z.tap_mask;
    }

    const int act = p.act;

    // ---- staging roles (see ConvCfg) ------------------------------------------------------------------
    const int sh = NW == 1 ? 0 : (wv & 1), spart = NW == 1 ? 0 : (wv >> 1);
    const int* __restrict__ tab = p.ch_off + (int64_t)ch_row * p.CinPad + sh;
    unsigned voff[NKP]; // in-plane byte offset of the patch pixel, kBufOOB outside the image (buffer load returns 0)
    int lidx[NKP];      // LDS vec_t index of the patch pixel (parity 0 of this wave), or -1 if not owned
#pragma unroll
    for (int k = 0; k < NKP; k++) {
        const int pidx = spart * (NKP * 64) + lane + 64 * k;
        const int pr = pidx / PC, pc = pidx - pr * PC;
        const int iy = ty0 * S - pad_y + pr;
        const int ix = tx0 * S - pad_x + pc;
        const bool own = pidx < NPIX;
        voff[k] = (own && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) ? (unsigned)(iy * p.x_pitch + ix) * ESX : kBufOOB;
        lidx[k] = own ? (pr * 2 + sh) * PC + pc : -1;
    }

    const float* __restrict__ wsrc = p.w + w_off + ((int64_t)nblk * nchunks) * Cfg::W_ELEMS;

    vec_t rin[NPAR][NKP];
    f32x4 rw[WLDS ? Cfg::NK_W : 1];

    // Output addressing (residual loads and stores): sample/slice base in the resource, channel offset in an
    // SGPR, and ONE per-lane byte offset per wave-tile (pixel + the 4-channel shift of the upper half-wave).
    // Accumulator register r of 32-block b holds channel cb + (r&3) + 8*(r>>2) + 4*half.
    const int64_t ybase = (int64_t)n * p.y_bstride + y_off;
    const int64_t rbase = (int64_t)n * p.r_bstride + r_off;
    const int cs32 = (int)p.y_cstride, rs32 = (int)p.r_cstride;
    const bool tail8 = (p.Cout & 7) != 0;          // only then is channel validity lane dependent
    unsigned yvoff[WT];
#pragma unroll
    for (int i = 0; i < WT; i++) {
        const int t = wv + NW * i;
        const int oy = ty0 + t / TXW;
        const int ox = tx0 + (t % TXW) * 32 + l31;
        yvoff[i] = !(oy < Ho && ox < Wo) ? kBufOOB
                   : YIL ? (unsigned)((oy * p.y_ystride + ox) * 4 + 4 * half * cs32) * ESY
                         : (unsigned)(oy * p.y_ystride + ox * p.y_xstride + 4 * half * cs32) * ESY;
    }
    // residual (skip connection) values of this lane's outputs: requested first thing, so their (HBM) latency
    // overlaps the first gather, and added to the accumulators before the first MFMA -- no registers are held
    // for them afterwards and the epilogue has no loads.
    // TAIL = channel count not a multiple of 8 (the only case with lane-dependent channel validity); both
    // versions are behind one scalar branch so the common one carries no selects.
    const float* __restrict__ resid = p.resid;
    float rv[WT][NBW][16];
    auto prefetch_resid_t = [&](auto TAIL) {
#pragma unroll
        for (int b = 0; b < NBW; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int cs = nblk * NB + b * 32 + (r & 3) + 8 * (r >> 2);
                const buf_rsrc rs = make_buf(elem_ptr(resid, rbase, ESY), (resid != nullptr) & (cs < p.Cout));
                const unsigned so = (unsigned)(cs * rs32) * ESY;
#pragma unroll
                for (int i = 0; i < WT; i++) {
                    // same pixel, the residual's own channel stride for the upper half-wave's 4-channel shift
                    const unsigned rvo = yvoff[i] == kBufOOB ? kBufOOB : yvoff[i] + (unsigned)(4 * half * (rs32 - cs32)) * ESY;
                    const unsigned vo = (decltype(TAIL)::value && cs + 4 * half >= p.Cout) ? kBufOOB : rvo;
                    rv[i][b][r] = kAblResid ? (float)cs : Io<TOUT>::load(rs, vo, so);
                }
            }
    };
    auto prefetch_resid = [&]() {
        if (tail8) prefetch_resid_t(std::true_type{});
        else prefetch_resid_t(std::false_type{});
    };

    // accumulators start at the bias (accumulator register r holds channel (r&3) + 8*(r>>2) + 4*half of its
    // 32-block; p.bias is padded to a multiple of 64), so the epilogue has no loads of its own
    f32x16 acc[WT][NBW];
#pragma unroll
    for (int b = 0; b < NBW; b++) {
        const float* bsrc = p.bias + blockIdx.y * NB + b * 32 + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bsrc + 8 * q);
#pragma unroll
            for (int i = 0; i < WT; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][b][4 * q + e] = bv[e];
        }
    }

    // Loads only.  Address = sample base + plane offset (SGPR, from the gather table) + in-plane offset (VGPR,
    // fixed per lane for the whole kernel): no vector ALU work per load.  Out-of-image lanes carry kBufOOB and
    // padded / out-of-range planes (table entry -1) use an empty resource, so both read 0.
    const buf_rsrc rs_w = make_buf(wsrc);
    unsigned wvoff[WLDS ? Cfg::NK_W : 1];
#pragma unroll
    for (int k = 0; k < (WLDS ? Cfg::NK_W : 1); k++) {
        const int idx = tid + NTHR * k;
        wvoff[k] = idx < Cfg::W_ELEMS / 4 ? (unsigned)idx * 16u : kBufOOB;
    }
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int h = 0; h < NPAR; h++)
#pragma unroll
            for (int j = 0; j < CPG; j++) {
                const int off = tab[ch * CC + 2 * j + h];          // wave-uniform scalar load
                const buf_rsrc rs = make_buf(xb, off >= 0);
                const unsigned so = (unsigned)off * ESX;
#pragma unroll
                for (int k = 0; k < NKP; k++)
                    rin[h][k][j] = kAblGather ? (float)(off + (int)voff[k]) : Io<TIN>::load(rs, voff[k], so);
            }
        if (WLDS) {
            const unsigned so = (unsigned)ch * (unsigned)(Cfg::W_ELEMS * 4);
#pragma unroll
            for (int k = 0; k < Cfg::NK_W; k++) {
                if (kAblWLoad) rw[k] = f32x4{(float)wvoff[k], 1.f, 2.f, (float)ch};
                else rw[k] = buf_load4(rs_w, wvoff[k], so);
            }
        }
    };

    // per-lane LDS read bases (vec_t units)
    int a_base[NBW], b_base[WT];
#pragma unroll
    for (int b = 0; b < NBW; b++) a_base[b] = half * NB + b * 32 + l31;
#pragma unroll
    for (int i = 0; i < WT; i++) {
        const int t = wv + NW * i;
        b_base[i] = ((t / TXW) * S * 2 + half) * PC + ((t % TXW) * 32 + l31) * S;
    }
    const vec_t* __restrict__ gW4 = reinterpret_cast<const vec_t*>(wsrc);   // weights straight from L1/L2 (WLDS == false)

#ifdef RT_KERNEL_TIMING
    unsigned long long* dbgp = p.dbg ? p.dbg + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 : nullptr;
    int dbi = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // registers -> LDS for the chunk that was prefetched last
    auto stage_to_lds = [&](int buf) {
        if (kAblLdsWr) {        // keep the staged values alive without touching LDS
#pragma unroll
            for (int h = 0; h < NPAR; h++)
#pragma unroll
                for (int k = 0; k < NKP; k++) asm volatile("" ::"v"(rin[h][k]));
            if (WLDS) {
#pragma unroll
                for (int k = 0; k < Cfg::NK_W; k++) asm volatile("" ::"v"(rw[k]));
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < NPAR; h++)
#pragma unroll
            for (int k = 0; k < NKP; k++)
                if (lidx[k] >= 0) reinterpret_cast<vec_t*>(sIn[buf])[lidx[k] + h * PC] = rin[h][k];
        if (WLDS) {
#pragma unroll
            for (int k = 0; k < Cfg::NK_W; k++) {
                const int idx = tid + NTHR * k;
                if (idx < Cfg::W_ELEMS / 4) reinterpret_cast<f32x4*>(sW[buf])[idx] = rw[k];
            }
        }
    };
    // all MFMAs of one chunk; operands of tap t+1 are fetched before the MFMAs of tap t
    auto compute = [&](int ch, int buf) {
        const vec_t* sIn4 = reinterpret_cast<const vec_t*>(sIn[buf]);
        const vec_t* wq = WLDS ? reinterpret_cast<const vec_t*>(sW[buf]) : gW4 + (int64_t)ch * (TAPS * 2 * NB);
        vec_t a_cur[NBW], b_cur[WT], a_nxt[NBW], b_nxt[WT];
        auto lds_a = [&](int idx) { vec_t v; if (kAblLdsRd) { for (int e = 0; e < CPG; e++) v[e] = (float)(idx + e); } else v = wq[idx]; return v; };
        auto lds_b = [&](int idx) { vec_t v; if (kAblLdsRd) { for (int e = 0; e < CPG; e++) v[e] = (float)(idx - e); } else v = sIn4[idx]; return v; };
#pragma unroll
        for (int b = 0; b < NBW; b++) a_cur[b] = lds_a(a_base[b]);
#pragma unroll
        for (int i = 0; i < WT; i++) b_cur[i] = lds_b(b_base[i]);
#pragma unroll
        for (int t = 0; t < TAPS; t++) {
            if (t + 1 < TAPS) {
                const int r = (t + 1) / KW, s = (t + 1) % KW;
#pragma unroll
                for (int b = 0; b < NBW; b++) a_nxt[b] = lds_a(a_base[b] + (t + 1) * 2 * NB);
#pragma unroll
                for (int i = 0; i < WT; i++) b_nxt[i] = lds_b(b_base[i] + r * 2 * PC + s);
            }
            if ((tap_mask >> t) & 1u) {            // wave-uniform
#pragma unroll
                for (int j = 0; j < CPG; j++)
#pragma unroll
                    for (int i = 0; i < WT; i++)
#pragma unroll
                        for (int b = 0; b < NBW; b++)
                            acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[b][j], b_cur[i][j], acc[i][b], 0, 0, 0);
            }
            if (t + 1 < TAPS) {
#pragma unroll
                for (int b = 0; b < NBW; b++) a_cur[b] = a_nxt[b];
#pragma unroll
                for (int i = 0; i < WT; i++) b_cur[i] = b_nxt[i];
            }
        }
    };

    RT_TSTAMP();
    prefetch_resid();
    prefetch(0);
#pragma unroll
    for (int i = 0; i < WT; i++)
#pragma unroll
        for (int b = 0; b < NBW; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][b][r] += rv[i][b][r];
    stage_to_lds(0);
    wg_barrier();
    RT_TSTAMP();
    for (int ch = 0; ch < nchunks; ch++) {
        const int cur = NBUF == 2 ? (ch & 1) : 0, nxt = NBUF == 2 ? (cur ^ 1) : 0;
        const bool more = ch + 1 < nchunks;
        if (more) prefetch(ch + 1);        // global loads fly while the MFMAs below run
        compute(ch, cur);
        RT_TSTAMP();
        if (more) {
            if (NBUF == 1) wg_barrier();   // single stage: everyone must be done reading it first
            stage_to_lds(nxt);
            wg_barrier();
            RT_TSTAMP();
        }
    }

    RT_TSTAMP();
    // ---- epilogue: activation, 128-byte coalesced stores -------------------------------------------------------
    // (bias and residual went in with the accumulator init; one scalar branch per activation / tail case keeps the
    //  16-element loops free of selects)
    auto epilogue = [&](auto ACT, auto TAIL) {
        if constexpr (YIL) {
#pragma unroll
            for (int b = 0; b < NBW; b++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int cs = nblk * NB + b * 32 + 8 * q;
                    const buf_rsrc rs = make_buf(elem_ptr(p.y, ybase, ESY), cs < p.Cout);
                    const unsigned so = (unsigned)(cs * cs32) * ESY;
#pragma unroll
                    for (int i = 0; i < WT; i++) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = apply_act_fast(acc[i][b][4 * q + e], decltype(ACT)::value);
                        const unsigned vo = (cs + 4 * half >= p.Cout) ? kBufOOB : yvoff[i];      // Cout is a multiple of 4
                        if (!kAblStore || o[0] == 12345.678f)
                            buf_store4(o, rs, vo, so);
                    }
                }
            return;
        }
#pragma unroll
        for (int b = 0; b < NBW; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int cs = nblk * NB + b * 32 + (r & 3) + 8 * (r >> 2);
                const buf_rsrc rs = make_buf(elem_ptr(p.y, ybase, ESY), cs < p.Cout);
                const unsigned so = (unsigned)(cs * cs32) * ESY;
#pragma unroll
                for (int i = 0; i < WT; i++) {
                    const float v = apply_act_fast(acc[i][b][r], decltype(ACT)::value);
                    const unsigned vo = (decltype(TAIL)::value && cs + 4 * half >= p.Cout) ? kBufOOB : yvoff[i];
                    if (!kAblStore || v == 12345.678f) Io<TOUT>::store(v, rs, vo, so);
                }
            }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (tail8) {
        if (act == 1) epilogue(I1{}, std::true_type{});
        else if (act == 2) epilogue(I2{}, std::true_type{});
        else epilogue(I0{}, std::true_type{});
    } else {
        if (act == 1) epilogue(I1{}, std::false_type{});
        else if (act == 2) epilogue(I2{}, std::false_type{});
        else epilogue(I0{}, std::false_type{});
    }
#ifdef RT_KERNEL_TIMING
    __builtin_amdgcn_s_waitcnt(0);
#endif
    RT_TSTAMP();
#ifdef RT_KERNEL_TIMING
    if (dbgp && tid == 0) dbgp[15] = __builtin_amdgcn_s_memrealtime() - rt0;   // 100 MHz reference clock
#endif
}


// ------------------------------------------------------------------------------------------------------
// Direct (VALU) convolution for layers with one or two output channels, where a 32-wide MFMA column
// block would be >= 94 % padding: the last transposed convolutions of the networks (32 -> 1 at full
// resolution, reference resnet18_2D_513x257_net.cpp:758-763, nvsmall_1025x321_net.cpp:399-408).
// Same gather description as the MFMA kernel (window, plane table, ZSlice phases); one thread per output
// pixel, coalesced along x, weights [co][ci][tap] read through the scalar cache.  HBM-bound.
// grid = (ceil(Wo/256), Ho, batch*nz)
// ------------------------------------------------------------------------------------------------------
template <int COUT, int KH, int KW>
__global__ void __launch_bounds__(256) conv_direct_f32_kernel(ConvArgs p, int S, int cin_real) {
    constexpr int TAPS = KH * KW;
    constexpr int kMaxW = 4096;                       // floats of weights cached in LDS
    __shared__ float sw[kMaxW];
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const int oy = blockIdx.y;
    const int zi = blockIdx.z % p.nz, n = blockIdx.z / p.nz;
    int pad_y = p.pad_y, pad_x = p.pad_x, Ho = p.Ho, Wo = p.Wo, ch_row = zi;
    int64_t y_off = p.y_off + (int64_t)zi * p.y_zstride, w_off = 0;
    if (p.zs) {
        const ZSlice z = p.zs[zi];
        pad_y = z.pad_y; pad_x = z.pad_x; Ho = z.Ho; Wo = z.Wo; ch_row = z.ch_row; y_off = z.y_off; w_off = z.w_off;
    }
    if (oy >= Ho) return;                             // uniform per workgroup
    const float* __restrict__ w = p.w + w_off;        // [co][ci][tap]
    const int nw = COUT * cin_real * TAPS;            // launcher guarantees nw <= kMaxW
    for (int i = threadIdx.x; i < nw; i += 256) sw[i] = w[i];
    __syncthreads();
    const float* __restrict__ xb = p.x + (int64_t)n * p.x_bstride;
    const int* __restrict__ tab = p.ch_off + (int64_t)ch_row * p.CinPad;
    // window offsets of this output pixel, -1 outside the image
    int poff[TAPS];
#pragma unroll
    for (int u = 0; u < KH; u++)
#pragma unroll
        for (int v = 0; v < KW; v++) {
            const int iy = oy * S + u - pad_y, ix = ox * S + v - pad_x;
            poff[u * KW + v] = (ox < Wo && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) ? iy * p.x_pitch + ix : -1;
        }
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; co++) acc[co] = p.bias[co];
#pragma unroll 4
    for (int ci = 0; ci < cin_real; ci++) {
        const int off = tab[ci];                      // wave-uniform
        if (off < 0) continue;
        float xv[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; t++) xv[t] = *(poff[t] >= 0 ? xb + ((int64_t)off + poff[t]) : p.zeros);
#pragma unroll
        for (int t = 0; t < TAPS; t++)
#pragma unroll
            for (int co = 0; co < COUT; co++) acc[co] = fmaf(xv[t], sw[(co * cin_real + ci) * TAPS + t], acc[co]);
    }
    if (ox >= Wo) return;
    const int64_t base = (int64_t)n * p.y_bstride + y_off + (int64_t)oy * p.y_ystride + (int64_t)ox * p.y_xstride;
#pragma unroll
    for (int co = 0; co < COUT; co++) {
        if (co < p.Cout) {
            const int64_t addr = base + (int64_t)co * p.y_cstride;
            float v = acc[co];
            if (p.resid) v += p.resid[addr];
            p.y[addr] = apply_act_rt(v, p.act);
        }
    }
}

}  // namespace rt
