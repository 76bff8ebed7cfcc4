// Issue/throughput cost of vector memory instructions on one gfx950 CU (L1/L2-resident data):
// cycles per wave-instruction for dword / dwordx2 / dwordx4 loads, aligned and row-misaligned, and dword stores.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int W, int MIS>
__global__ void __launch_bounds__(256) load_kernel(const float* g, float* out, unsigned long long* cyc, int iters) {
    // every wave walks its own 8 KB window (L1 resident after the first pass), lane stride W floats
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* base = g + ((size_t)blockIdx.x * 4 + wave) * 4096 + MIS;
    float s = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float* p = base + ((u * 64 * W) & 2047) + lane * W;
            if (W == 1) s += *p;
            if (W == 2) { f32x2 v = *reinterpret_cast<const f32x2*>(p); s += v[0] + v[1]; }
            if (W == 4) { f32x4 v = *reinterpret_cast<const f32x4*>(p); s += v[0] + v[1] + v[2] + v[3]; }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

__global__ void __launch_bounds__(256) store_kernel(float* g, unsigned long long* cyc, int iters, int mis) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* base = g + ((size_t)blockIdx.x * 4 + wave) * 4096 + mis;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int u = 0; u < 8; u++) base[((u * 64) & 2047) + lane] = (float)it;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

float* g; float* out; unsigned long long* cyc; unsigned long long h[4096];
template <typename F> void report(const char* name, int wgs_per_cu, int iters, F launch) {
    const int grid = 256 * wgs_per_cu;
    launch(grid); (void)hipDeviceSynchronize();
    launch(grid); (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < grid; i++) c += h[i];
    c /= grid;
    // per CU: wgs_per_cu workgroups x 4 waves x iters*8 instructions in c cycles
    printf("%-34s %d WG/CU: %6.1f cycles per wave-instruction per CU\n", name, wgs_per_cu, c / (wgs_per_cu * 4.0 * iters * 8));
}

int main() {
    (void)hipMalloc(&g, (size_t)256 * 8 * 4 * 4096 * 4 + 64); (void)hipMemset(g, 0, (size_t)256 * 8 * 4 * 4096 * 4 + 64);
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 4096 * 8);
    const int iters = 2000;
    for (int w : {2, 8}) {
        report("load dword aligned", w, iters, [&](int grid) { hipLaunchKernelGGL((load_kernel<1, 0>), dim3(grid), dim3(256), 0, 0, g, out, cyc, iters); });
        report("load dword +1 float (2 lines)", w, iters, [&](int grid) { hipLaunchKernelGGL((load_kernel<1, 1>), dim3(grid), dim3(256), 0, 0, g, out, cyc, iters); });
        report("load dwordx2 aligned", w, iters, [&](int grid) { hipLaunchKernelGGL((load_kernel<2, 0>), dim3(grid), dim3(256), 0, 0, g, out, cyc, iters); });
        report("load dwordx4 aligned", w, iters, [&](int grid) { hipLaunchKernelGGL((load_kernel<4, 0>), dim3(grid), dim3(256), 0, 0, g, out, cyc, iters); });
        report("load dwordx4 +1 float", w, iters, [&](int grid) { hipLaunchKernelGGL((load_kernel<4, 1>), dim3(grid), dim3(256), 0, 0, g, out, cyc, iters); });
        report("store dword aligned", w, iters, [&](int grid) { hipLaunchKernelGGL(store_kernel, dim3(grid), dim3(256), 0, 0, g, cyc, iters, 0); });
        report("store dword +1 float", w, iters, [&](int grid) { hipLaunchKernelGGL(store_kernel, dim3(grid), dim3(256), 0, 0, g, cyc, iters, 1); });
    }
    return 0;
}
