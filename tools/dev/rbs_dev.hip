// Development harness for the streaming residual-block kernels: conv_s3rbs_kernel (round 2) against conv_s3rbs2_kernel (round 3) on
// one tower block of ResNet-18 2D at 1257x369 (629 x 185 x 32 channels), interleaved tensors.  Checks both against an fp64 evaluation on
// sampled pixels, compares them with each other everywhere, and times back-to-back launches (one stream, and two streams at once --
// the throughput set-up has two launches co-resident).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iredtail_amd/csrc -Iinclude tools/dev/rbs_dev.hip -o tools/build/rbs_dev
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "kernels/conv_rbs.hip.h"
#include "conv_rbs2.hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void split_f16(float w, uint16_t& hi, uint16_t& lo) {
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
    std::memcpy(&hi, &h, 2);
    std::memcpy(&lo, &l, 2);
}
// conv_s3_kernel's slab order (rt_capi.hip pack_into, split3): [chunk of 16 ci][tap][hi/lo][k-group][co][8 halfs]
static std::vector<float> pack(const std::vector<float>& w) {       // w[co][ci][3][3], 32 x 32
    std::vector<float> out((size_t)2 * 9 * 2 * 2 * 32 * 4, 0.f);
    uint16_t* dst = reinterpret_cast<uint16_t*>(out.data());
    for (int co = 0; co < 32; co++)
        for (int ci = 0; ci < 32; ci++)
            for (int t = 0; t < 9; t++) {
                uint16_t hi, lo;
                split_f16(w[((size_t)co * 32 + ci) * 9 + t], hi, lo);
                const int ch = ci / 16, kg = (ci % 16) / 8, e = ci % 8;
                const size_t slab = (size_t)ch * 9 + t;
                dst[(((slab * 2 + 0) * 2 + kg) * 32 + co) * 8 + e] = hi;
                dst[(((slab * 2 + 1) * 2 + kg) * 32 + co) * 8 + e] = lo;
            }
    return out;
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 629, H = argc > 2 ? atoi(argv[2]) : 185;
    const int seg = argc > 3 ? atoi(argv[3]) : 32, iters = argc > 4 ? atoi(argv[4]) : 40, batch = argc > 5 ? atoi(argv[5]) : 1;
    const int pitch = (W + 31) / 32 * 32;
    const size_t plane = (size_t)H * pitch, sample = 32 * plane;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x(sample * batch, 0.f), w1(32 * 32 * 9), w2(32 * 32 * 9), b1(64, 0.f), b2(64, 0.f);
    // interleaved (C/4, H, pitch, 4)
    auto xi = [&](int n, int c, int yy, int xx) -> float& { return x[n * sample + ((size_t)(c / 4) * plane + (size_t)yy * pitch + xx) * 4 + c % 4]; };
    for (int n = 0; n < batch; n++)
        for (int c = 0; c < 32; c++)
            for (int yy = 0; yy < H; yy++)
                for (int xx = 0; xx < W; xx++) xi(n, c, yy, xx) = nd(rng) * (1.f + 3.f * (c % 3 == 0));
    for (auto& v : w1) v = nd(rng) * 0.06f;
    for (auto& v : w2) v = nd(rng) * 0.06f;
    for (int c = 0; c < 32; c++) { b1[c] = nd(rng) * 0.1f; b2[c] = nd(rng) * 0.1f; }
    const std::vector<float> p1 = pack(w1), p2 = pack(w2);

    float *dx, *dy0, *dy1, *dw1, *dw2, *db1, *db2;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dy0, x.size() * 4)); CK(hipMalloc(&dy1, x.size() * 4));
    CK(hipMalloc(&dw1, p1.size() * 4)); CK(hipMalloc(&dw2, p2.size() * 4)); CK(hipMalloc(&db1, 256)); CK(hipMalloc(&db2, 256));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw1, p1.data(), p1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw2, p2.data(), p2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db1, b1.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db2, b2.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemset(dy0, 0xff, x.size() * 4)); CK(hipMemset(dy1, 0xff, x.size() * 4));

    rt::RBArgs a;
    std::memset(&a, 0, sizeof(a));
    a.c.x = dx; a.c.w = dw2; a.c.bias = db2; a.c.Hi = H; a.c.Wi = W; a.c.Ho = H; a.c.Wo = W; a.c.x_pitch = pitch;
    a.c.tiles_x = (W + 29) / 30; a.c.act = 1; a.c.xcd_order = 1; a.c.x_bstride = (int64_t)sample; a.c.y_bstride = (int64_t)sample;
    a.c.y_cstride = (int64_t)plane; a.c.y_ystride = pitch; a.c.y_xstride = 1; a.c.x_cstride = (int64_t)plane; a.c.batch = batch;
    a.w1 = dw1; a.bias1 = db1; a.act1 = 1; a.cmid = 32; a.seg = seg;
    const dim3 grid((unsigned)(a.c.tiles_x * ((H + seg - 1) / seg)), 1, batch);
    printf("%d x %d, %d-row segments, batch %d: grid %u x %u workgroups\n", W, H, seg, batch, grid.x, grid.z);

    auto launch = [&](int which, float* y, hipStream_t st) {
        rt::RBArgs b = a;
        b.c.y = y;
        if (which == 0) hipLaunchKernelGGL(rt::conv_s3rbs_kernel, grid, dim3(512), 0, st, b);
        else hipLaunchKernelGGL(rt::conv_s3rbs2_kernel, grid, dim3(512), 0, st, b);
    };
    launch(0, dy0, 0); launch(1, dy1, 0);
    CK(hipDeviceSynchronize());
    std::vector<float> y0(x.size()), y1(x.size());
    CK(hipMemcpy(y0.data(), dy0, x.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dy1, x.size() * 4, hipMemcpyDeviceToHost));
    auto yi = [&](const std::vector<float>& y, int n, int c, int yy, int xx) { return y[n * sample + ((size_t)(c / 4) * plane + (size_t)yy * pitch + xx) * 4 + c % 4]; };
    double dmax = 0;
    size_t bad = 0;
    for (int n = 0; n < batch; n++)
        for (int c = 0; c < 32; c++)
            for (int yy = 0; yy < H; yy++)
                for (int xx = 0; xx < W; xx++) {
                    const float u = yi(y0, n, c, yy, xx), v = yi(y1, n, c, yy, xx);
                    const double d = std::fabs((double)u - v);
                    if (!(d <= 1e-4)) { if (bad < 8) printf("  mismatch n%d c%d y%d x%d: old %g new %g\n", n, c, yy, xx, u, v); bad++; }
                    if (d > dmax) dmax = d;
                }
    printf("old vs new kernel: max |diff| %.3g, %zu elements differ by more than 1e-4\n", dmax, bad);
    // fp64 evaluation on sampled pixels (all channels), borders included
    auto elu = [](double v) { return v > 0 ? v : std::exp(v) - 1.0; };
    double e0 = 0, e1 = 0;
    std::uniform_int_distribution<int> ux(0, W - 1), uy(0, H - 1);
    for (int sidx = 0; sidx < 160; sidx++) {
        int xx = ux(rng), yy = uy(rng), n = sidx % batch;
        if (sidx < 16) { xx = (sidx & 1) ? W - 1 - (sidx >> 3) : (sidx >> 3); yy = (sidx & 2) ? H - 1 - ((sidx >> 2) & 1) : ((sidx >> 2) & 1); }
        if (sidx >= 16 && sidx < 48) { yy = ((sidx - 16) * seg / 4 + (sidx & 3)) % H; xx = (30 * (sidx - 16) + (sidx & 1) * 29) % W; }
        double t[3][3][32];
        for (int dy = -1; dy <= 1; dy++)
            for (int dxx = -1; dxx <= 1; dxx++)
                for (int cm = 0; cm < 32; cm++) {
                    const int ty = yy + dy, tx = xx + dxx;
                    double acc = 0;
                    if (ty < 0 || ty >= H || tx < 0 || tx >= W) { t[dy + 1][dxx + 1][cm] = 0; continue; }
                    for (int ci = 0; ci < 32; ci++)
                        for (int u = 0; u < 3; u++)
                            for (int v = 0; v < 3; v++) {
                                const int iy = ty + u - 1, ix = tx + v - 1;
                                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                                acc += (double)w1[((size_t)cm * 32 + ci) * 9 + u * 3 + v] * xi(n, ci, iy, ix);
                            }
                    t[dy + 1][dxx + 1][cm] = elu(acc + b1[cm]);
                }
        for (int co = 0; co < 32; co++) {
            double acc = 0;
            for (int cm = 0; cm < 32; cm++)
                for (int u = 0; u < 3; u++)
                    for (int v = 0; v < 3; v++) acc += (double)w2[((size_t)co * 32 + cm) * 9 + u * 3 + v] * t[u][v][cm];
            const double ref = elu(acc + b2[co] + xi(n, co, yy, xx));
            e0 = std::fmax(e0, std::fabs(ref - yi(y0, n, co, yy, xx)));
            e1 = std::fmax(e1, std::fabs(ref - yi(y1, n, co, yy, xx)));
        }
    }
    printf("max |y - fp64| on 160 sampled pixels x 32 channels: old %.3g, new %.3g\n", e0, e1);

#ifdef RT_KERNEL_TIMING
    {   // phase stamps of wave 0 / wave 4 of every workgroup: [wg][role][16]
        unsigned long long* dbg;
        const size_t nst = (size_t)grid.x * grid.z * 2 * 16;
        CK(hipMalloc(&dbg, nst * 8));
        for (int which = 0; which < 2; which++) {
            CK(hipMemset(dbg, 0, nst * 8));
            a.c.dbg = dbg;
            for (int i = 0; i < 3; i++) launch(which, dy0, 0);
            CK(hipDeviceSynchronize());
            a.c.dbg = nullptr;
            std::vector<unsigned long long> h(nst);
            CK(hipMemcpy(h.data(), dbg, nst * 8, hipMemcpyDeviceToHost));
            for (int role = 0; role < 2; role++) {
                double d[16] = {0};
                int cnt = 0;
                double life = 0, mhz = 0;
                for (size_t wg = 0; wg < (size_t)grid.x * grid.z; wg++) {
                    const unsigned long long* st = &h[(wg * 2 + role) * 16];
                    if (!st[0] || !st[14]) continue;
                    cnt++;
                    for (int i = 1; i <= 12; i++) if (st[i] && st[i - 1]) d[i] += (double)(st[i] - st[i - 1]);
                    if (st[13] && st[7] && st[8]) { d[13] += (double)(st[13] - st[7]); d[14] += (double)(st[8] - st[13]); }   // step 5: stream, barrier wait
                    life += (double)(st[14] - st[0]);
                    mhz += (double)(st[14] - st[0]) / ((double)st[15] / 100.0);
                }
                printf("%s kernel, %s wave: lifetime %.0f cycles (%.0f MHz); phases:", which ? "new" : "old", role ? "conv2" : "conv1", life / cnt, mhz / cnt);
                for (int i = 1; i <= 14; i++) printf(" %d:%.0f", i, d[i] / cnt);
                printf("\n");
            }
        }
    }
#endif
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    for (int which = 0; which < 2; which++) {
        for (int i = 0; i < 30; i++) launch(which, dy0, s1);             // clock spin-up
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(ea, s1));
        for (int i = 0; i < iters; i++) launch(which, dy0, s1);
        CK(hipEventRecord(eb, s1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, ea, eb));
        const float one = ms * 1e3f / iters;
        // two streams: both towers' launches co-resident
        hipEvent_t ec, ed;
        CK(hipEventCreate(&ec)); CK(hipEventCreate(&ed));
        CK(hipEventRecord(ea, s1)); CK(hipEventRecord(ec, s2));
        for (int i = 0; i < iters; i++) { launch(which, dy0, s1); launch(which, dy1, s2); }
        CK(hipEventRecord(eb, s1)); CK(hipEventRecord(ed, s2));
        CK(hipDeviceSynchronize());
        float m1, m2;
        CK(hipEventElapsedTime(&m1, ea, eb)); CK(hipEventElapsedTime(&m2, ec, ed));
        printf("%s kernel: %.2f us per launch alone; two streams: %.2f us per launch pair (%.2f us per block)\n", which ? "new" : "old", one,
               std::fmax(m1, m2) * 1e3f / iters, std::fmax(m1, m2) * 1e3f / iters / 2);
    }
    return 0;
}
