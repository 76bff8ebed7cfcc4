// Round 4: is a packed-fp32 VALU instruction (v_pk_add_f32 -- what the SLP vectoriser makes of two adjacent fp32 adds) exact when ANOTHER
// wave of the same SIMD issues fp16 MFMAs (v_mfma_f32_32x32x16_f16)?  profiles/r04_race.txt: the exact-fp32 Winograd kernel's interleaved
// epilogue (SLP-packed ELU arithmetic) deviated in the high half of a pair on lanes 48-63 only beside the split-fp16 kernels, and not at
// all when built with -fno-slp-vectorize.  Waves 4-7 of a workgroup repeat a short packed sequence on fixed inputs and compare every result
// with the scalar form; waves 0-3 (one per SIMD) issue MFMAs (mode bit 0) or nothing.
//     hipcc --offload-arch=gfx950 -O3 tools/micro/pk_f32_beside_mfma.hip -o /tmp/pk && /tmp/pk
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512) probe(unsigned long long* bad, float* sink, int iters, int mode, int variant) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x16 acc;
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        f16x8 a, b;
        for (int j = 0; j < 8; j++) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(blockIdx.x * 0.002f + j); }
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        float s = 0.f;
        for (int r = 0; r < 16; r++) s += acc[r];
        if (s == 12345.f) sink[threadIdx.x] = s;
        return;
    }
    // victim waves: x0, x1 per lane; expected results from scalar instructions computed once up front
    const float x0 = -0.25f - 0.01f * lane, x1 = -1.5f + 0.02f * lane, y0 = 0.75f + lane, y1 = -2.f * lane;
    unsigned long long nbad_lo = 0, nbad_hi = 0;
    for (int it = 0; it < iters; it++) {
        f32x2 r;
        float e0, e1;
        if (variant == 0) {                     // exp pair, then packed add of the inline constant -1.0 to both halves (the ELU epilogue)
            asm volatile("v_exp_f32 %0, %3\n\tv_exp_f32 %1, %4\n\tv_pk_add_f32 %2, %5, -1.0 op_sel_hi:[1,0]"
                         : "=&v"(e0), "=&v"(e1), "=&v"(r) : "v"(x0), "v"(x1), "v"(f32x2{__builtin_amdgcn_exp2f(x0), __builtin_amdgcn_exp2f(x1)}));
            const float w0 = __builtin_amdgcn_exp2f(x0) - 1.0f, w1 = __builtin_amdgcn_exp2f(x1) - 1.0f;
            nbad_lo += r[0] != w0; nbad_hi += r[1] != w1;
        } else if (variant == 1) {              // plain packed add of two register pairs
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(r) : "v"(f32x2{x0, x1}), "v"(f32x2{y0, y1}));
            nbad_lo += r[0] != x0 + y0; nbad_hi += r[1] != x1 + y1;
        } else if (variant == 2) {              // cross-half form of the Winograd output transform
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(f32x2{x0, y1}));
            nbad_lo += r[0] != x0 + y1; nbad_hi += r[1] != y1 + x0;
        } else {                                // a transcendental right before the packed add (its result is not read)
            asm volatile("v_exp_f32 %0, %2\n\tv_pk_add_f32 %1, %3, %4 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(e0), "=&v"(r) : "v"(x0), "v"(f32x2{x0, x1}), "v"(f32x2{y0, y1}));
            nbad_lo += r[0] != x0 - y0; nbad_hi += r[1] != x1 - y1;
        }
    }
    if (nbad_lo) atomicAdd(&bad[0], nbad_lo);
    if (nbad_hi) atomicAdd(&bad[1 + (lane >> 4)], nbad_hi);         // high half, per lane quarter
}

int main() {
    unsigned long long* bad; float* sink;
    (void)hipMalloc(&bad, 64); (void)hipMalloc(&sink, 4096);
    const int iters = 200000;
    for (int variant = 0; variant < 4; variant++)
        for (int mode = 0; mode < 2; mode++) {
            (void)hipMemset(bad, 0, 64);
            hipLaunchKernelGGL(probe, dim3(512), dim3(512), 0, 0, bad, sink, iters, mode, variant);
            (void)hipDeviceSynchronize();
            unsigned long long h[5];
            (void)hipMemcpy(h, bad, 40, hipMemcpyDeviceToHost);
            printf("variant %d, %-22s: %d x %d packed results per lane checked: wrong low halves %llu, wrong high halves by lane quarter %llu %llu %llu %llu\n", variant,
                   mode ? "MFMA waves beside" : "alone", 512 * 4, iters, h[0], h[1], h[2], h[3], h[4]);
        }
    return 0;
}
