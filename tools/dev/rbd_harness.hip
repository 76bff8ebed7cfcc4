// Development harness of conv_s3rbd_kernel (conv_rbd.hip.h): the tower block alone, against an fp64 host evaluation and beside
// conv_s3rbs_kernel (the block it replaces), on ResNet-18 2D's tower geometry (32 channels, 185 x 629, pitch 640, two images per launch).
// Never part of the product; built and run by tools/r06/rbd_dev.sh.
//     rbd_harness [check] [time] [phases]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../redtail_amd/csrc/kernels/conv_rbs.hip.h"
#include "../../redtail_amd/csrc/kernels/conv_rbd.hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static const int H = 185, W = 629, P = 640, C = 32, NB = 2;

static void split16(float v, uint16_t& hi, uint16_t& lo) {
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)((v - (float)h) * 2048.f);
    memcpy(&hi, &h, 2); memcpy(&lo, &l, 2);
}
static float join16(uint16_t hi, uint16_t lo) {
    _Float16 h, l; memcpy(&h, &hi, 2); memcpy(&l, &lo, 2);
    return (float)h + (float)l / 2048.f;
}
// row of the A operand that holds output channel co (conv_rbd.hip.h: lane (pixel, kg) owns channels 8 kg .. + 7 and 16 + 8 kg .. + 7)
static int rbd_row(int co) { return 8 * (2 * (co >> 4) + ((co >> 2) & 1)) + 4 * ((co >> 3) & 1) + (co & 3); }

// [chunk * 9 + tap][hi / lo][k-group][row][8 halfs]
static std::vector<uint16_t> pack_slab(const std::vector<float>& w, bool perm) {
    std::vector<uint16_t> s((size_t)18 * 2 * 64 * 8, 0);
    for (int co = 0; co < 32; co++)
        for (int ci = 0; ci < 32; ci++)
            for (int t = 0; t < 9; t++) {
                uint16_t hi, lo;
                split16(w[((size_t)co * 32 + ci) * 9 + t], hi, lo);
                const int ch = ci / 16, kg = (ci % 16) / 8, e = ci % 8, row = perm ? rbd_row(co) : co;
                const size_t slab = (size_t)ch * 9 + t;
                s[(((slab * 2 + 0) * 2 + kg) * 32 + row) * 8 + e] = hi;
                s[(((slab * 2 + 1) * 2 + kg) * 32 + row) * 8 + e] = lo;
            }
    return s;
}

int main(int argc, char** argv) {
    bool do_check = false, do_time = false, do_phases = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "check")) do_check = true;
        if (!strcmp(argv[i], "time")) do_time = true;
        if (!strcmp(argv[i], "phases")) do_phases = true;
    }
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x((size_t)NB * C * H * W), w1(32 * 32 * 9), w2(32 * 32 * 9), b1(64, 0.f), b2(64, 0.f);
    for (auto& v : x) v = nd(rng);
    for (auto& v : w1) v = nd(rng) / sqrtf(288.f);
    for (auto& v : w2) v = nd(rng) / sqrtf(288.f);
    for (int i = 0; i < 32; i++) { b1[i] = 0.3f * nd(rng); b2[i] = 0.3f * nd(rng); }
    // tensors: split (C/8, H, P, [8 hi | 8 lo]) and fp32 (C/4, H, P, 4)
    const size_t samp = (size_t)C * H * P;          // elements (4 bytes each) per sample in either layout
    std::vector<uint16_t> xs(2 * samp * NB, 0);
    std::vector<float> x4(samp * NB, 0.f), xe((size_t)NB * C * H * W);
    for (int n = 0; n < NB; n++)
        for (int c = 0; c < C; c++)
            for (int y = 0; y < H; y++)
                for (int xx = 0; xx < W; xx++) {
                    const float v = x[(((size_t)n * C + c) * H + y) * W + xx];
                    uint16_t hi, lo;
                    split16(v, hi, lo);
                    const size_t rec = (size_t)n * 2 * samp + ((((size_t)(c / 8) * H + y) * P + xx) * 16);
                    xs[rec + (c % 8)] = hi; xs[rec + 8 + (c % 8)] = lo;
                    xe[(((size_t)n * C + c) * H + y) * W + xx] = join16(hi, lo);
                    x4[(size_t)n * samp + (((size_t)(c / 4) * H + y) * P + xx) * 4 + (c % 4)] = v;
                }
    const std::vector<uint16_t> s1p = pack_slab(w1, true), s2p = pack_slab(w2, true), s1 = pack_slab(w1, false), s2 = pack_slab(w2, false);
    void *dxs, *dx4, *dy, *dy2, *ds1p, *ds2p, *ds1, *ds2, *db1, *db2;
    CK(hipMalloc(&dxs, samp * NB * 4)); CK(hipMalloc(&dx4, samp * NB * 4)); CK(hipMalloc(&dy, samp * NB * 4)); CK(hipMalloc(&dy2, samp * NB * 4));
    CK(hipMalloc(&ds1p, s1p.size() * 2)); CK(hipMalloc(&ds2p, s2p.size() * 2)); CK(hipMalloc(&ds1, s1.size() * 2)); CK(hipMalloc(&ds2, s2.size() * 2));
    CK(hipMalloc(&db1, 256)); CK(hipMalloc(&db2, 256));
    CK(hipMemcpy(dxs, xs.data(), samp * NB * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx4, x4.data(), samp * NB * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds1p, s1p.data(), s1p.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ds2p, s2p.data(), s2p.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds1, s1.data(), s1.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ds2, s2.data(), s2.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db1, b1.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db2, b2.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemset(dy, 0xff, samp * NB * 4)); CK(hipMemset(dy2, 0xff, samp * NB * 4));
    unsigned long long* ddbg = nullptr;
    CK(hipMalloc(&ddbg, 1024 * 2 * 16 * 8)); CK(hipMemset(ddbg, 0, 1024 * 2 * 16 * 8));

    auto args = [&](bool newk, void* xin, void* yout, int seg) {
        rt::RBArgs a;
        memset(&a, 0, sizeof(a));
        a.c.x = (const float*)xin; a.c.y = (float*)yout; a.c.resid = (const float*)xin;
        a.c.w = (const float*)(newk ? ds2p : ds2); a.c.bias = (const float*)db2;
        a.w1 = (const float*)(newk ? ds1p : ds1); a.bias1 = (const float*)db1;
        a.c.dbg = ddbg;
        a.c.CinPad = 32; a.c.Cout = 32; a.c.Hi = H; a.c.Wi = W; a.c.x_pitch = P; a.c.Ho = H; a.c.Wo = W; a.c.pad_y = 1; a.c.pad_x = 1; a.c.nz = 1;
        a.c.tiles_x = (W + 29) / 30; a.c.act = 1; a.c.xcd_order = 1;
        a.c.x_bstride = (int64_t)samp; a.c.y_bstride = (int64_t)samp; a.c.y_cstride = (int64_t)H * P; a.c.y_ystride = P; a.c.y_xstride = 1;
        a.c.r_cstride = (int64_t)H * P; a.c.r_bstride = (int64_t)samp; a.c.r_il8 = 1; a.c.batch = NB; a.c.cin_real = 32; a.c.x_cstride = (int64_t)H * P;
        a.act1 = 1; a.cmid = 32; a.seg = seg;
        return a;
    };
    auto grid_of = [&](int seg) { return dim3((unsigned)(((W + 29) / 30) * ((H + seg - 1) / seg)), 1u, (unsigned)NB); };

    if (do_check) {
        // fp64 evaluation from the values the kernel sees (xe = hi + lo / 2048), rows in parallel
        std::vector<double> t((size_t)NB * C * H * W), yr((size_t)NB * C * H * W);
        auto elu = [](double v) { return v > 0 ? v : std::exp(v) - 1.0; };
        auto conv = [&](const std::vector<double>* tin, const std::vector<float>* xin, const std::vector<float>& w, const std::vector<float>& b,
                        std::vector<double>& out, bool skip) {
            const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
            std::vector<std::thread> th;
            for (unsigned ti = 0; ti < nt; ti++)
                th.emplace_back([&, ti] {
                    for (int job = ti; job < NB * H; job += nt) {
                        const int n = job / H, y = job % H;
                        for (int co = 0; co < C; co++)
                            for (int xx = 0; xx < W; xx++) {
                                double acc = b[co];
                                for (int ci = 0; ci < C; ci++)
                                    for (int dy = 0; dy < 3; dy++) {
                                        const int iy = y + dy - 1;
                                        if (iy < 0 || iy >= H) continue;
                                        for (int dx = 0; dx < 3; dx++) {
                                            const int ix = xx + dx - 1;
                                            if (ix < 0 || ix >= W) continue;
                                            const size_t idx = (((size_t)n * C + ci) * H + iy) * W + ix;
                                            acc += (double)w[((size_t)co * 32 + ci) * 9 + dy * 3 + dx] * (tin ? (*tin)[idx] : (double)(*xin)[idx]);
                                        }
                                    }
                                const size_t o = (((size_t)n * C + co) * H + y) * W + xx;
                                if (skip) acc += (double)xe[o];
                                out[o] = elu(acc);
                            }
                    }
                });
            for (auto& t_ : th) t_.join();
        };
        conv(nullptr, &xe, w1, b1, t, false);
        conv(&t, nullptr, w2, b2, yr, true);
        for (int seg : {64, 32, 16, 188}) {
            for (int ysplit = 1; ysplit >= 0; ysplit--) {
                CK(hipMemset(dy, 0xff, samp * NB * 4));
                rt::RBArgs a = args(true, dxs, dy, seg);
                if (ysplit) hipLaunchKernelGGL(rt::conv_s3rbd_kernel<true>, grid_of(seg), dim3(512), 0, 0, a);
                else hipLaunchKernelGGL(rt::conv_s3rbd_kernel<false>, grid_of(seg), dim3(512), 0, 0, a);
                CK(hipGetLastError()); CK(hipDeviceSynchronize());
                std::vector<uint16_t> yh(2 * samp * NB);
                CK(hipMemcpy(yh.data(), dy, samp * NB * 4, hipMemcpyDeviceToHost));
                const float* yf = reinterpret_cast<const float*>(yh.data());
                double maxe = 0, maxr = 0; size_t bad = 0;
                for (int n = 0; n < NB; n++)
                    for (int c = 0; c < C; c++)
                        for (int y = 0; y < H; y++)
                            for (int xx = 0; xx < W; xx++) {
                                float got;
                                if (ysplit) {
                                    const size_t rec = (size_t)n * 2 * samp + ((((size_t)(c / 8) * H + y) * P + xx) * 16);
                                    got = join16(yh[rec + (c % 8)], yh[rec + 8 + (c % 8)]);
                                } else {
                                    got = yf[(size_t)n * samp + (((size_t)(c / 4) * H + y) * P + xx) * 4 + (c % 4)];
                                }
                                const double ref = yr[(((size_t)n * C + c) * H + y) * W + xx];
                                const double e = std::fabs((double)got - ref);
                                if (!(e <= 2e-5 * std::max(1.0, std::fabs(ref)))) {
                                    if (bad < 8) fprintf(stderr, "  seg %d split %d: n %d c %d y %d x %d got %g ref %g\n", seg, ysplit, n, c, y, xx, got, ref);
                                    bad++;
                                }
                                if (e == e) { maxe = std::max(maxe, e); maxr = std::max(maxr, std::fabs(ref)); }
                            }
                printf("check seg %3d y_split %d: max |err| %.3g (max |ref| %.3g), %zu of %zu outside 2e-5\n", seg, ysplit, maxe, maxr, bad, (size_t)NB * C * H * W);
            }
        }
    }

    if (do_time) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](const char* name, auto launch, int reps) {
            for (int i = 0; i < 5; i++) launch((hipStream_t)0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; i++) launch((hipStream_t)0);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-44s %8.2f us per launch\n", name, ms * 1e3 / reps);
        };
        for (int seg : {64, 32, 16}) {
            rt::RBArgs an = args(true, dxs, dy, seg), ao = args(false, dx4, dy2, seg);
            const dim3 g = grid_of(seg);
            char nm[96];
            snprintf(nm, sizeof nm, "rbd (new), split out, seg %d, %u wgs", seg, g.x * g.z);
            timeit(nm, [&](hipStream_t st) { hipLaunchKernelGGL(rt::conv_s3rbd_kernel<true>, g, dim3(512), 0, st, an); }, 50);
            snprintf(nm, sizeof nm, "rbd (new), fp32 out, seg %d", seg);
            timeit(nm, [&](hipStream_t st) { hipLaunchKernelGGL(rt::conv_s3rbd_kernel<false>, g, dim3(512), 0, st, an); }, 50);
            snprintf(nm, sizeof nm, "rbs (old), seg %d", seg);
            timeit(nm, [&](hipStream_t st) { hipLaunchKernelGGL(rt::conv_s3rbs_kernel<false>, g, dim3(512), 0, st, ao); }, 50);
        }
        // in company: 4 streams, each a chain of launches (64-row segments: 126 workgroups per launch)
        {
            hipStream_t st[4];
            for (auto& s : st) CK(hipStreamCreate(&s));
            void* ys[4];
            for (auto& y_ : ys) CK(hipMalloc(&y_, samp * NB * 4));
            for (int newk = 1; newk >= 0; newk--) {
                const int seg = 64, reps = 40;
                const dim3 g = grid_of(seg);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; i++)
                    for (int k = 0; k < 4; k++) {
                        rt::RBArgs a = args(newk, newk ? dxs : dx4, ys[k], seg);
                        if (newk) hipLaunchKernelGGL(rt::conv_s3rbd_kernel<true>, g, dim3(512), 0, st[k], a);
                        else hipLaunchKernelGGL(rt::conv_s3rbs_kernel<false>, g, dim3(512), 0, st[k], a);
                    }
                for (int k = 0; k < 4; k++) CK(hipStreamSynchronize(st[k]));
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("%s in company (4 streams x %d launches, seg 64): %8.2f us per launch\n", newk ? "rbd (new)" : "rbs (old)", reps, ms * 1e3 / (4 * reps));
            }
        }
    }

#ifdef RT_KERNEL_TIMING
    if (do_phases) {
        for (int newk = 1; newk >= 0; newk--) {
            const int seg = 64;
            const dim3 g = grid_of(seg);
            rt::RBArgs a = args(newk, newk ? dxs : dx4, dy, seg);
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(ddbg, 0, 1024 * 2 * 16 * 8));
                if (newk) hipLaunchKernelGGL(rt::conv_s3rbd_kernel<true>, g, dim3(512), 0, 0, a);
                else hipLaunchKernelGGL(rt::conv_s3rbs_kernel<false>, g, dim3(512), 0, 0, a);
                CK(hipDeviceSynchronize());
            }
            const int nwg = g.x * g.z;
            std::vector<unsigned long long> d((size_t)nwg * 2 * 16);
            CK(hipMemcpy(d.data(), ddbg, d.size() * 8, hipMemcpyDeviceToHost));
            for (int role = 0; role < 2; role++) {
                printf("%s %s wave: mean shader cycles between stamps over %d workgroups\n", newk ? "rbd (new)" : "rbs (old)", role ? "conv2" : "conv1", nwg);
                for (int i = 1; i < (newk ? 12 : 14); i++) {
                    double sum = 0; int cnt = 0;
                    for (int wgi = 0; wgi < nwg; wgi++) {
                        const unsigned long long* q = d.data() + ((size_t)wgi * 2 + role) * 16;
                        if (q[i] && q[i - 1]) { sum += (double)(q[i] - q[i - 1]); cnt++; }
                    }
                    if (cnt) printf("   stamp %2d -> %2d  %9.1f\n", i - 1, i, sum / cnt);
                }
                double life = 0, real = 0;
                for (int wgi = 0; wgi < nwg; wgi++) { const unsigned long long* q = d.data() + ((size_t)wgi * 2 + role) * 16; life += (double)(q[14] - q[0]); real += (double)q[15]; }
                printf("   workgroup lifetime %9.1f cycles = %.2f us (constant 100 MHz counter): shader clock %.0f MHz\n", life / nwg, real / nwg / 100.0, life / real * 100.0);
                if (newk && role == 0) {
                    std::vector<double> st, du; unsigned long long t_min = ~0ull, t_max = 0;
                    for (int wgi = 0; wgi < nwg; wgi++) { const unsigned long long* q = d.data() + ((size_t)wgi * 2) * 16; t_min = std::min(t_min, q[12]); t_max = std::max(t_max, q[13]); }
                    for (int wgi = 0; wgi < nwg; wgi++) { const unsigned long long* q = d.data() + ((size_t)wgi * 2) * 16; st.push_back((double)(q[12] - t_min) / 100.0); du.push_back((double)(q[13] - q[12]) / 100.0); }
                    std::sort(st.begin(), st.end()); std::sort(du.begin(), du.end());
                    printf("   kernel span %.2f us; workgroup starts p50 %.2f p90 %.2f max %.2f us after the first; lifetimes min %.2f p50 %.2f p90 %.2f max %.2f us\n",
                           (double)(t_max - t_min) / 100.0, st[nwg / 2], st[nwg * 9 / 10], st[nwg - 1], du[0], du[nwg / 2], du[nwg * 9 / 10], du[nwg - 1]);
                }
            }
        }
    }
#endif
    return 0;
}
