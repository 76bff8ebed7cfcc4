// Round 5: what does an LDS-DMA load (`buffer_load_dwordx4 ... lds`) do for a lane whose buffer offset is OUT OF RANGE -- write zeros
// to its LDS slot or leave the slot alone?  And for a lane switched off in EXEC?  conv_f16dw_kernel (conv_f16dw.hip.h) zero-fills its
// patch buffers once and never relies on either answer; this probe records the answer for the next kernel that wants to.
//     hipcc --offload-arch=gfx950 -O3 tools/micro/lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) probe(const float* src, float* out) {
    __shared__ __attribute__((aligned(16))) f32x4 lds[128];
    const int lane = threadIdx.x;
    lds[lane] = f32x4{-1.f, -1.f, -1.f, -1.f};
    lds[64 + lane] = f32x4{-2.f, -2.f, -2.f, -2.f};
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 64 * 16, 0x00020000);
    // piece 0: odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&lds[0], 16,
                                             (lane & 1) ? 0x80000000u : (unsigned)lane * 16u, 0u, 0, 0);
    // piece 1: lanes >= 32 switched off in EXEC
    if (lane < 32)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&lds[64], 16, (unsigned)lane * 16u, 0u, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 2; i++) {
        const f32x4 v = lds[64 * i + lane];
        out[(64 * i + lane) * 4 + 0] = v[0]; out[(64 * i + lane) * 4 + 1] = v[1]; out[(64 * i + lane) * 4 + 2] = v[2]; out[(64 * i + lane) * 4 + 3] = v[3];
    }
}

int main() {
    float h[64 * 4], *src, *out, r[128 * 4];
    for (int i = 0; i < 256; i++) h[i] = 100.f + i;
    (void)hipMalloc(&src, sizeof(h)); (void)hipMalloc(&out, sizeof(r));
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out);
    (void)hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    int oob_zero = 0, oob_kept = 0, oob_other = 0, in_ok = 0, off_kept = 0, off_other = 0, on_ok = 0;
    for (int l = 0; l < 64; l++) {
        const float v = r[l * 4];
        if (l & 1) { if (v == 0.f) oob_zero++; else if (v == -1.f) oob_kept++; else oob_other++; }
        else in_ok += v == 100.f + 4 * l;
        const float w = r[(64 + l) * 4];
        if (l >= 32) { if (w == -2.f) off_kept++; else off_other++; }
        else on_ok += w == 100.f + 4 * l;
    }
    printf("LDS-DMA, 32 out-of-range lanes: %d wrote zeros, %d left the slot alone, %d wrote something else (in-range lanes correct: %d / 32)\n",
           oob_zero, oob_kept, oob_other, in_ok);
    printf("LDS-DMA, 32 lanes off in EXEC: %d left the slot alone, %d wrote something (active lanes correct: %d / 32)\n", off_kept, off_other, on_ok);
    return 0;
}
