// How many non-MFMA instructions hide under one v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles on a SIMD) when they sit IN THE
// SAME WAVE's instruction stream, by kind of filler -- and what happens when two such waves share a SIMD?
//
// Round 2 (mfma_f16_valu.hip) only measured an MFMA-only wave next to a VALU-only wave (92 % of the sum) and DESIGN 4.0 concluded
// "matrix and vector issue never overlap".  MI355X_MICROARCH.md documents <= 5 single-issue fillers hidden per MFMA gap for ONE
// wave per SIMD.  This probe measures the cases conv_rbs.hip.h can be rebuilt on:
//
//   self   : one wave per SIMD, stream = [MFMA, K fillers] x 8 per iteration, 4 accumulators in rotation (no RAW between MFMAs)
//   pair   : two waves per SIMD, both run that stream
//   asym   : two waves per SIMD, waves 0-3 run [MFMA, K fillers], waves 4-7 run fillers only (8 per MFMA slot of the partner)
//
// Everything is inline asm (the order in the binary is the order written here).  Output: shader cycles per MFMA per wave
// (s_memtime deltas, mean over all waves), and the same per SIMD for the two-wave modes.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_interleave.hip -o /tmp/mfma_interleave && /tmp/mfma_interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum Kind { FMA = 0, EXP, CVT, CMPSEL, PKFMA, DSR128, DSW64, VMEMLD, MIXLDS, FMAMIX, SALU, SNOP, DSW2, VMEMST, NKIND };
static const char* kKindName[NKIND] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "v_cmp+v_cndmask", "v_pk_fma_f32", "ds_read_b128",
                                       "ds_write_b64", "global_load_dwordx4", "1 ds_read_b128 + (K-1) v_fma", "v_fma_mix_f32", "s_add_u32", "s_nop 0",
                                       "ds_write2_b64", "global_store_dwordx4"};

struct Regs {
    float x[8];
    f32x2 p[4];
    f32x4 q[4];
    unsigned h[5];
};

template <int KIND>
__device__ __forceinline__ void filler(int j, Regs& r, float m, float c, unsigned lds_addr, const f32x4* gp) {
    const int i = j & 3;
    if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.x[j & 7]) : "v"(m), "v"(c));
    else if constexpr (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r.x[j & 7]));
    else if constexpr (KIND == CVT) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.h[i]) : "v"(r.x[i]), "v"(r.x[i + 4]));
    else if constexpr (KIND == CMPSEL) {
        if (j & 1) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r.x[4 + i]) : "v"(m), "v"(c) : );
        else asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r.x[i]), "v"(c) : "vcc");
    } else if constexpr (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r.p[i]) : "v"(r.p[(i + 1) & 3]), "v"(r.p[(i + 2) & 3]));
    else if constexpr (KIND == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(r.q[i]) : "v"(lds_addr + 1024u * i));
    else if constexpr (KIND == DSW64) asm volatile("ds_write_b64 %0, %1" : : "v"(lds_addr + 512u * i), "v"(r.p[i]) : "memory");
    else if constexpr (KIND == VMEMLD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.q[i]) : "v"(gp + 64 * i));
    else if constexpr (KIND == MIXLDS) {
        if ((j & 7) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(r.q[i]) : "v"(lds_addr));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.x[j & 7]) : "v"(m), "v"(c));
    } else if constexpr (KIND == SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(r.h[4]) : : "scc");
    else if constexpr (KIND == SNOP) asm volatile("s_nop 0");
    else if constexpr (KIND == DSW2) asm volatile("ds_write2_b64 %0, %1, %2 offset1:4" : : "v"(lds_addr + 2048u * i), "v"(r.p[i]), "v"(r.p[(i + 1) & 3]) : "memory");
    else if constexpr (KIND == VMEMST) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gp + 64 * i), "v"(r.q[i]) : "memory");
    else if constexpr (KIND == FMAMIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,1]" : "+v"(r.x[j & 7]) : "v"(m), "v"(r.h[i]));
}

// NW = waves per SIMD, MODE 0: every wave runs [MFMA, K fillers]; MODE 1: waves >= 4 run fillers only (FK per partner MFMA slot)
template <int KIND, int K, int NW, int MODE, int FK>
__global__ void __launch_bounds__(NW * 256) probe(unsigned long long* cyc, float* out, const f32x4* g, int iters) {
    __shared__ f32x4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += NW * 256) lds[i] = f32x4{1.f, 2.f, 3.f, (float)i};
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(blockIdx.x * 0.002f + j); }
    Regs r;
    for (int j = 0; j < 8; j++) r.x[j] = 0.001f * (j + lane);
    for (int j = 0; j < 4; j++) { r.p[j] = f32x2{0.5f + j, 0.25f}; r.q[j] = f32x4{0.f, 0.f, 0.f, 0.f}; r.h[j] = 0x3c003c00u; }
    r.h[4] = 0;
    const float m = 0.9999f, c = 1e-3f;
    const unsigned lds_addr = (unsigned)(size_t)lds + lane * 16u;     // conflict-free 16-byte lanes
    const f32x4* gp = g + (size_t)blockIdx.x * 1024 + threadIdx.x;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0 || wave < 4) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
#pragma unroll
                for (int j = 0; j < K; j++) filler<KIND>(u * K + j, r, m, c, lds_addr, gp);
            }
            if (KIND == DSR128 || KIND == MIXLDS || KIND == DSW64 || KIND == DSW2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (KIND == VMEMLD || KIND == VMEMST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int j = 0; j < FK; j++) filler<KIND>(u * FK + j, r, m, c, lds_addr, gp);
            }
            if (KIND == DSR128 || KIND == MIXLDS || KIND == DSW64 || KIND == DSW2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (KIND == VMEMLD || KIND == VMEMST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * (NW * 4) + wave] = c1 - c0;
    float s = 0.f;
    for (int j = 0; j < 4; j++)
        for (int e = 0; e < 16; e++) s += acc[j][e];
    for (int j = 0; j < 8; j++) s += r.x[j];
    for (int j = 0; j < 4; j++) s += r.p[j][0] + r.p[j][1] + r.q[j][0] + r.q[j][3] + (float)r.h[j];
    s += (float)r.h[4];
    if (s == 12345.f) out[threadIdx.x] = s;
}

static unsigned long long* d_cyc;
static float* d_out;
static f32x4* d_g;
static const int kIters = 1500, kBlocks = 256;

template <int KIND, int K, int NW, int MODE, int FK>
void run() {
    const int nw = NW * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((probe<KIND, K, NW, MODE, FK>), dim3(kBlocks), dim3(NW * 256), 0, 0, d_cyc, d_out, d_g, kIters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, K, NW, MODE, FK>), dim3(kBlocks), dim3(NW * 256), 0, 0, d_cyc, d_out, d_g, kIters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(kBlocks * nw);
    (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < kBlocks; b++)
        for (int w = 0; w < nw; w++) (w < 4 ? lo : hi) += (double)h[b * nw + w];
    const double n_mfma = (double)kIters * 8;
    lo /= kBlocks * 4.0;
    hi = NW == 2 ? hi / (kBlocks * 4.0) : 0;
    if (MODE == 0 && NW == 1)
        printf("self  %-30s K=%d : %6.1f cycles per MFMA (wave)                         wall %.3f ms\n", kKindName[KIND], K, lo / n_mfma, ms);
    else if (MODE == 0)
        printf("pair  %-30s K=%d : %6.1f / %6.1f cycles per own MFMA (waves 0-3 / 4-7) = %6.1f per SIMD-MFMA   wall %.3f ms\n", kKindName[KIND], K,
               lo / n_mfma, hi / n_mfma, (lo > hi ? lo : hi) / (2 * n_mfma), ms);
    else
        printf("asym  %-30s K=%d + partner %d fillers per slot : MFMA waves %6.1f cycles per MFMA, filler waves done after %6.1f cycles per slot   wall %.3f ms\n",
               kKindName[KIND], K, FK, lo / n_mfma, hi / n_mfma, ms);
    fflush(stdout);
}

template <int KIND>
void sweep_self() {
    run<KIND, 0, 1, 0, 0>(); run<KIND, 1, 1, 0, 0>(); run<KIND, 2, 1, 0, 0>(); run<KIND, 3, 1, 0, 0>(); run<KIND, 4, 1, 0, 0>();
    run<KIND, 5, 1, 0, 0>(); run<KIND, 6, 1, 0, 0>(); run<KIND, 8, 1, 0, 0>(); run<KIND, 12, 1, 0, 0>();
}
template <int KIND>
void sweep_pair() {
    run<KIND, 0, 2, 0, 0>(); run<KIND, 2, 2, 0, 0>(); run<KIND, 3, 2, 0, 0>(); run<KIND, 4, 2, 0, 0>(); run<KIND, 5, 2, 0, 0>();
    run<KIND, 6, 2, 0, 0>(); run<KIND, 8, 2, 0, 0>();
}

template <int KIND>
void sweep_asym() {
    run<KIND, 0, 2, 1, 1>(); run<KIND, 0, 2, 1, 2>(); run<KIND, 0, 2, 1, 4>(); run<KIND, 0, 2, 1, 8>();
    run<KIND, 4, 2, 1, 4>();     // MFMA + 4 of them in the MFMA wave, 4 per slot in the partner
}
int main(int argc, char** argv) {
    (void)hipMalloc(&d_cyc, kBlocks * 8 * 8);
    (void)hipMalloc(&d_out, 4096);
    (void)hipMalloc(&d_g, (size_t)kBlocks * 2048 * 16);
    (void)hipMemset(d_g, 0, (size_t)kBlocks * 2048 * 16);
    if (argc > 1) {      // round 3, second question: do SALU / s_nop / LDS reads of EITHER wave cost the SIMD issue time next to MFMAs?
        (void)hipMalloc(&d_cyc, kBlocks * 8 * 8); (void)hipMalloc(&d_out, 4096); (void)hipMalloc(&d_g, (size_t)kBlocks * 2048 * 16);
        sweep_self<SALU>(); sweep_self<SNOP>(); sweep_self<DSW2>(); sweep_self<VMEMST>();
        sweep_pair<SALU>(); sweep_pair<SNOP>(); sweep_pair<DSR128>();
        sweep_asym<SALU>(); sweep_asym<SNOP>(); sweep_asym<DSR128>(); sweep_asym<FMA>(); sweep_asym<DSW2>(); sweep_asym<VMEMLD>(); sweep_asym<VMEMST>();
        // mixed partner: MFMA + 4 v_fma in one wave, the other wave the same plus one LDS read (MIXLDS K = 5)
        run<MIXLDS, 5, 2, 0, 0>(); run<MIXLDS, 6, 2, 0, 0>();
        return 0;
    }
    printf("[MFMA, K fillers] x 8 per iteration, %d iterations, %d workgroups (one per CU); v_mfma_f32_32x32x16_f16 = 32 cycles per SIMD\n", kIters, kBlocks);
    sweep_self<FMA>(); sweep_self<EXP>(); sweep_self<CVT>(); sweep_self<CMPSEL>(); sweep_self<PKFMA>(); sweep_self<FMAMIX>();
    sweep_self<DSR128>(); sweep_self<DSW64>(); sweep_self<VMEMLD>(); sweep_self<MIXLDS>();
    sweep_pair<FMA>(); sweep_pair<EXP>(); sweep_pair<MIXLDS>(); sweep_pair<DSW64>();
    // the round-2 set-up and its relatives: MFMA wave (with K own fillers) next to a filler-only wave
    run<FMA, 0, 2, 1, 2>(); run<FMA, 0, 2, 1, 4>(); run<FMA, 0, 2, 1, 8>(); run<FMA, 0, 2, 1, 12>();
    run<FMA, 2, 2, 1, 2>(); run<FMA, 2, 2, 1, 4>(); run<FMA, 4, 2, 1, 2>(); run<FMA, 4, 2, 1, 4>();
    run<EXP, 0, 2, 1, 2>(); run<EXP, 0, 2, 1, 4>(); run<DSW64, 0, 2, 1, 2>(); run<DSW64, 0, 2, 1, 4>();
    return 0;
}
