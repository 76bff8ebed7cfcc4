// The device form of s3_split (conv_split.hip.h: v_cvt_pk_f16_f32 + v_fma_mix_f32 + v_fma_mixlo/hi_f16) against its definition
//     hi = fp16(v), lo = fp16((v - float(hi)) * 2^11)
// for ALL 2^32 fp32 bit patterns (NaNs compared as NaNs).  Prints the number of differing hi / lo halves; 0 0 is the pass.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I redtail_amd/csrc/kernels -I redtail_amd/csrc -I include tools/micro/split_mix.hip -o /tmp/split_mix && /tmp/split_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include "conv_split.hip.h"

__device__ static inline bool same_h(_Float16 a, _Float16 b) {
    const unsigned short x = __builtin_bit_cast(unsigned short, a), y = __builtin_bit_cast(unsigned short, b);
    const bool nx = (x & 0x7fff) > 0x7c00, ny = (y & 0x7fff) > 0x7c00;
    return (nx && ny) || x == y;
}

__global__ void check(unsigned long long* bad, unsigned* first) {
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;      // 2^30 threads, 4 patterns each
    rt::f32x4 v;
    for (int j = 0; j < 4; j++) v[j] = __builtin_bit_cast(float, (unsigned)(4 * t + j));
    const rt::S3Split s = rt::s3_split(v);
    for (int j = 0; j < 4; j++) {
        volatile float x = v[j];                                   // the definition, kept away from the pattern matcher
        const _Float16 h = (_Float16)x;
        const float r = x - (float)h;
        const _Float16 l = (_Float16)(r * rt::kSplitScale);
        if (!same_h(h, s.hi[j])) { atomicAdd(&bad[0], 1ull); atomicMin(&first[0], (unsigned)(4 * t + j)); }
        if (!same_h(l, s.lo[j])) { atomicAdd(&bad[1], 1ull); atomicMin(&first[1], (unsigned)(4 * t + j)); }
    }
}

int main() {
    unsigned long long* bad; unsigned* first;
    (void)hipMalloc(&bad, 16); (void)hipMalloc(&first, 8);
    (void)hipMemset(bad, 0, 16); (void)hipMemset(first, 0xff, 8);
    hipLaunchKernelGGL(check, dim3(1u << 22), dim3(256), 0, 0, bad, first);
    unsigned long long hb[2]; unsigned hf[2];
    (void)hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
    printf("s3_split, all 2^32 inputs: differing hi %llu (first 0x%08x), differing lo %llu (first 0x%08x)\n", hb[0], hf[0], hb[1], hf[1]);
    return (hb[0] || hb[1]) ? 1 : 0;
}
