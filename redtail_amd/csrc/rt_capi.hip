// C-ABI of the MI355X kernels (include/rt_stereo.h): argument checking, launch geometry, weight
// re-packing and gather-table construction.  No CPU compute path exists here on purpose: if there is
// no HIP device every entry point fails with a non-zero status.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/rt_stereo.h"
// Every device build of this file goes through redtail_amd/build.py (DEVICE_FLAGS): compiler-formed packed fp32 math (the SLP vectoriser
// of -O3) beside co-resident fp16-MFMA waves computed wrong values in round 3 / 4 (profiles/r04_race.txt: 1662 of 4000 launches; 0 with
// -fno-slp-vectorize) and the hardware mechanism is not established -- so the flag is part of the source's contract, not of one build
// script: it travels with a define, and a compile without the pair stops here.  tools/check_no_packed_f32.py disassembles the product.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_BUILT_NO_SLP)
#error "rt_capi.hip must be compiled with -fno-slp-vectorize -DRT_BUILT_NO_SLP (redtail_amd/build.py: DEVICE_FLAGS)"
#endif
#include "kernels/common.hip.h"
#include "kernels/conv_mfma.hip.h"
#include "kernels/conv_wino.hip.h"
#include "kernels/deconv3d_small.hip.h"
#include "kernels/imgproc.hip.h"
#include "kernels/conv_f16.hip.h"
#include "kernels/conv_f16_first.hip.h"
#include "kernels/conv_f16r4.hip.h"
#include "kernels/conv_f16dw.hip.h"
#include "kernels/deconv_f16p.hip.h"
#include "kernels/conv_split.hip.h"
#include "kernels/conv_rbs.hip.h"
#include "kernels/conv_rbd.hip.h"
#include "kernels/conv_rbh.hip.h"
#include "kernels/deconv_s3p.hip.h"
#include "kernels/fold_factor.hip.h"
#include "kernels/cost_volume.hip.h"
#include "kernels/corr_mfma.hip.h"
#include "kernels/elementwise.hip.h"
#include "kernels/layout.hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail((int)e, "%s: HIP error %d (%s: %s)", what, (int)e, hipGetErrorName(e), hipGetErrorString(e));
}

#define RT_HIP(call)                                        \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(e_, #call);   \
    } while (0)

#define RT_LAUNCH_CHECK(what)                               \
    do {                                                    \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) return hip_fail(e_, what);    \
    } while (0)

#define RT_REQUIRE(cond, ...)                               \
    do {                                                    \
        if (!(cond)) return fail(RT_E_BADARG, __VA_ARGS__); \
    } while (0)

inline hipStream_t S(rtStream s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline unsigned ew_blocks(int64_t n, int vec) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(rt::cdiv(n, 256 * (int64_t)vec), 256 * 16));
}

int launch_ew(const void* a, const void* b, void* y, int64_t n, int act, int dtype, bool add, hipStream_t st) {
    if (n == 0) return 0;
    RT_REQUIRE(a && y && (!add || b) && n > 0, "elementwise: null pointer or negative size");
    const bool al = aligned16(a) && aligned16(y) && (!add || aligned16(b));
    if (dtype == RT_F32) {
        auto pa = static_cast<const float*>(a);
        auto pb = static_cast<const float*>(b);
        auto py = static_cast<float*>(y);
        if (al) {
            if (add) hipLaunchKernelGGL((rt::ew_f32_kernel<4, true>), dim3(ew_blocks(n, 4)), dim3(256), 0, st, pa, pb, py, n, act);
            else hipLaunchKernelGGL((rt::ew_f32_kernel<4, false>), dim3(ew_blocks(n, 4)), dim3(256), 0, st, pa, pb, py, n, act);
        } else {
            if (add) hipLaunchKernelGGL((rt::ew_f32_kernel<1, true>), dim3(ew_blocks(n, 1)), dim3(256), 0, st, pa, pb, py, n, act);
            else hipLaunchKernelGGL((rt::ew_f32_kernel<1, false>), dim3(ew_blocks(n, 1)), dim3(256), 0, st, pa, pb, py, n, act);
        }
    } else if (dtype == RT_F16) {
        auto pa = static_cast<const _Float16*>(a);
        auto pb = static_cast<const _Float16*>(b);
        auto py = static_cast<_Float16*>(y);
        if (al) {
            if (add) hipLaunchKernelGGL((rt::ew_f16_kernel<8, true>), dim3(ew_blocks(n, 8)), dim3(256), 0, st, pa, pb, py, n, act);
            else hipLaunchKernelGGL((rt::ew_f16_kernel<8, false>), dim3(ew_blocks(n, 8)), dim3(256), 0, st, pa, pb, py, n, act);
        } else {
            if (add) hipLaunchKernelGGL((rt::ew_f16_kernel<1, true>), dim3(ew_blocks(n, 1)), dim3(256), 0, st, pa, pb, py, n, act);
            else hipLaunchKernelGGL((rt::ew_f16_kernel<1, false>), dim3(ew_blocks(n, 1)), dim3(256), 0, st, pa, pb, py, n, act);
        }
    } else {
        return fail(RT_E_UNSUPPORTED, "elementwise: dtype %d not supported", dtype);
    }
    RT_LAUNCH_CHECK("elementwise kernel");
    return 0;
}

size_t dsize(int dtype) { return dtype == RT_F16 ? 2 : 4; }

template <typename T>
int launch_copy_rows(const void* src, void* dst, int64_t rows, int64_t len, int64_t sstride, int64_t dstride,
                     int64_t ztail, hipStream_t st) {
    if (rows == 0 || len + ztail == 0) return 0;
    RT_REQUIRE(rows <= 65535, "copy: more than 65535 rows");
    const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(rt::cdiv(len + ztail, 256), 2048));
    hipLaunchKernelGGL((rt::copy_rows_kernel<T>), dim3(bx, (unsigned)rows), dim3(256), 0, st, static_cast<const T*>(src),
                       static_cast<T*>(dst), len, sstride, dstride, ztail);
    RT_LAUNCH_CHECK("copy_rows kernel");
    return 0;
}

int copy_rows(const void* src, void* dst, int64_t rows, int64_t len, int64_t sstride, int64_t dstride, int64_t ztail,
              int dtype, hipStream_t st) {
    if (dtype == RT_F32) return launch_copy_rows<float>(src, dst, rows, len, sstride, dstride, ztail, st);
    if (dtype == RT_F16) return launch_copy_rows<_Float16>(src, dst, rows, len, sstride, dstride, ztail, st);
    return fail(RT_E_UNSUPPORTED, "copy: dtype %d not supported", dtype);
}

}  // namespace

// =================================================================================================
// library / device
// =================================================================================================
extern "C" const char* rt_last_error_string(void) { return g_err.c_str(); }

extern "C" const char* rt_backend_name(void) {
    static thread_local char name[320];
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        snprintf(name, sizeof(name), "hip:none");
        return name;
    }
    snprintf(name, sizeof(name), "hip:%s (%s)", prop.gcnArchName, prop.name);
    return name;
}

extern "C" int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int rt_set_device(int ordinal) { RT_HIP(hipSetDevice(ordinal)); return 0; }
extern "C" int rt_malloc(void** dptr, size_t bytes) {
    RT_REQUIRE(dptr, "rt_malloc: null out pointer");
    RT_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}
extern "C" int rt_free(void* dptr) { if (dptr) RT_HIP(hipFree(dptr)); return 0; }
extern "C" int rt_memcpy_h2d(void* dst, const void* src, size_t bytes, rtStream s) {
    if (bytes) RT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(s)));
    return 0;
}
extern "C" int rt_memcpy_d2h(void* dst, const void* src, size_t bytes, rtStream s) {
    if (bytes) RT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(s)));
    return 0;
}
extern "C" int rt_memcpy_d2d(void* dst, const void* src, size_t bytes, rtStream s) {
    if (bytes) RT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(s)));
    return 0;
}
extern "C" int rt_memset(void* dst, int value, size_t bytes, rtStream s) {
    if (bytes) RT_HIP(hipMemsetAsync(dst, value, bytes, S(s)));
    return 0;
}
extern "C" int rt_stream_create(rtStream* s) {
    RT_REQUIRE(s, "rt_stream_create: null");
    hipStream_t st;
    RT_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *s = st;
    return 0;
}
extern "C" int rt_stream_destroy(rtStream s) { RT_HIP(hipStreamDestroy(S(s))); return 0; }
extern "C" int rt_stream_sync(rtStream s) { RT_HIP(hipStreamSynchronize(S(s))); return 0; }
extern "C" int rt_stream_wait_event(rtStream s, void* ev) { RT_HIP(hipStreamWaitEvent(S(s), (hipEvent_t)ev, 0)); return 0; }
extern "C" int rt_event_create(void** ev) {
    RT_REQUIRE(ev, "rt_event_create: null");
    hipEvent_t e;
    RT_HIP(hipEventCreate(&e));
    *ev = e;
    return 0;
}
extern "C" int rt_event_create_ordering(void** ev) {
    RT_REQUIRE(ev, "rt_event_create_ordering: null");
    hipEvent_t e;
    RT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = e;
    return 0;
}
extern "C" int rt_event_destroy(void* ev) { RT_HIP(hipEventDestroy((hipEvent_t)ev)); return 0; }
extern "C" int rt_event_record(void* ev, rtStream s) { RT_HIP(hipEventRecord((hipEvent_t)ev, S(s))); return 0; }
extern "C" int rt_event_elapsed_ms(void* a, void* b, float* ms) {
    RT_REQUIRE(ms, "rt_event_elapsed_ms: null");
    RT_HIP(hipEventSynchronize((hipEvent_t)b));
    RT_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return 0;
}

// hipGraph capture / replay of a stream's launches (the executor's graph mode, engine.cpp: ContextImpl::run)
struct rtGraph {
#ifndef HIPEMU
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
#endif
};
extern "C" int rt_graph_begin_capture(rtStream s) {
#ifdef HIPEMU
    (void)s;
    return fail(RT_E_UNSUPPORTED, "rt_graph_begin_capture: no graphs on the emulator");
#else
    RT_REQUIRE(s, "rt_graph_begin_capture: the NULL stream cannot be captured");
    RT_HIP(hipStreamBeginCapture(S(s), hipStreamCaptureModeRelaxed));
    return 0;
#endif
}
extern "C" int rt_graph_end_capture(rtStream s, rtGraph** out) {
    RT_REQUIRE(out, "rt_graph_end_capture: null");
    *out = nullptr;
#ifdef HIPEMU
    (void)s;
    return fail(RT_E_UNSUPPORTED, "rt_graph_end_capture: no graphs on the emulator");
#else
    hipGraph_t g = nullptr;
    RT_HIP(hipStreamEndCapture(S(s), &g));
    RT_REQUIRE(g, "rt_graph_end_capture: the capture produced no graph");
    hipGraphExec_t e = nullptr;
    const hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    if (rc != hipSuccess) { (void)hipGraphDestroy(g); return fail((int)rc, "hipGraphInstantiate: %s", hipGetErrorString(rc)); }
    rtGraph* r = new rtGraph;
    r->graph = g; r->exec = e;
    *out = r;
    return 0;
#endif
}
extern "C" int rt_graph_launch(rtGraph* g, rtStream s) {
    RT_REQUIRE(g, "rt_graph_launch: null graph");
#ifdef HIPEMU
    (void)s;
    return fail(RT_E_UNSUPPORTED, "rt_graph_launch: no graphs on the emulator");
#else
    RT_HIP(hipGraphLaunch(g->exec, S(s)));
    return 0;
#endif
}
extern "C" int rt_graph_destroy(rtGraph* g) {
    if (!g) return 0;
#ifndef HIPEMU
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
#endif
    delete g;
    return 0;
}

// =================================================================================================
// element-wise
// =================================================================================================
extern "C" int rt_elu(const void* x, void* y, int64_t n, int dtype, rtStream s) {
    return launch_ew(x, nullptr, y, n, RT_ACT_ELU, dtype, false, S(s));
}
extern "C" int rt_add_act(const void* a, const void* b, void* y, int64_t n, int act, int dtype, rtStream s) {
    RT_REQUIRE(act >= 0 && act <= 2, "rt_add_act: bad activation %d", act);
    return launch_ew(a, b, y, n, act, dtype, true, S(s));
}
extern "C" int rt_activation(const void* x, void* y, int64_t n, int act, int dtype, rtStream s) {
    RT_REQUIRE(act >= 0 && act <= 2, "rt_activation: bad activation %d", act);
    return launch_ew(x, nullptr, y, n, act, dtype, false, S(s));
}

// =================================================================================================
// cost volumes / soft-argmax
// =================================================================================================
namespace {
template <bool FUSED, bool ISMIN, bool H2 = false, typename T = float>
int launch_corr(const float* l, const float* r, float* out, int batch, int C, int H, int W, int D, int64_t obs,
                hipStream_t st, int in_pitch = 0, int out_pitch = 0) {
    const int ip = in_pitch ? in_pitch : W, op = out_pitch ? out_pitch : W;
    const int dt = std::min(16, rt::round_up((int)rt::cdiv(D, 4), 4));
    if (FUSED && D > 4 * dt) return fail(RT_E_UNSUPPORTED, "fused corr+softargmax supports max_disp <= 64 (got %d)", D);
    dim3 grid((unsigned)rt::cdiv(W, rt::kCorrTX), (unsigned)rt::cdiv(H, rt::kCorrRY), (unsigned)batch);
    for (int d_base = 0; d_base < D; d_base += 4 * dt) {
        switch (dt) {
            case 4: hipLaunchKernelGGL((rt::corr_f32_kernel<4, FUSED, ISMIN, H2, T>), grid, dim3(256), 0, st, l, r, out, C, H, W, D, d_base, obs, ip, op); break;
            case 8: hipLaunchKernelGGL((rt::corr_f32_kernel<8, FUSED, ISMIN, H2, T>), grid, dim3(256), 0, st, l, r, out, C, H, W, D, d_base, obs, ip, op); break;
            case 12: hipLaunchKernelGGL((rt::corr_f32_kernel<12, FUSED, ISMIN, H2, T>), grid, dim3(256), 0, st, l, r, out, C, H, W, D, d_base, obs, ip, op); break;
            default: hipLaunchKernelGGL((rt::corr_f32_kernel<16, FUSED, ISMIN, H2, T>), grid, dim3(256), 0, st, l, r, out, C, H, W, D, d_base, obs, ip, op); break;
        }
        RT_LAUNCH_CHECK("corr cost volume kernel");
    }
    return 0;
}
}  // namespace

namespace { int env_int(const char* name, int dflt); }      // (development knobs, defined with the convolution plans below)

extern "C" int rt_corr_cost_volume(const void* left, const void* right, void* cv, int batch, int C, int H, int W,
                                   int D, int dtype, int format, rtStream s) {
    return rt_corr_cost_volume_flags(left, right, cv, batch, C, H, W, D, dtype, format, 0u, s);
}

extern "C" int rt_corr_cost_volume_flags(const void* left, const void* right, void* cv, int batch, int C, int H, int W,
                                         int D, int dtype, int format, unsigned flags, rtStream s) {
    RT_REQUIRE(left && right && cv, "rt_corr_cost_volume: null pointer");
    RT_REQUIRE(batch > 0 && C > 0 && H > 0 && W > 0 && D > 0, "rt_corr_cost_volume: bad dims");
    // the two combinations the reference plugin accepts (lib/cost_volume_plugin.cpp:60-66): fp32 NCHW, fp16 NC2HW2
    if (dtype == RT_F16 && format == RT_NC2HW2)
        return launch_corr<false, false, true>(static_cast<const float*>(left), static_cast<const float*>(right),
                                               static_cast<float*>(cv), batch, C, H, W, D, (int64_t)((D + 1) / 2) * H * W, S(s));
    if (dtype != RT_F32 || format != RT_NCHW)
        return fail(RT_E_UNSUPPORTED, "rt_corr_cost_volume: fp32 NCHW or fp16 NC2HW2 (dtype %d format %d)", dtype, format);
    // maps of a network's size: the Gram band on the matrix cores (corr_mfma_planar_kernel, 3-term fp16 split as in the engines'
    // correlation: |x| < 65504, see rt_check_range); small maps, D > 64, C > 32 and RT_CONV_EXACT_FP32 keep the fp32 fmaf kernel
    if (C >= 16 && C <= 32 && W >= 64 && D <= 64 && (int64_t)std::max(C, D) * H * W < (1ll << 29) && !(flags & RT_CONV_EXACT_FP32) && env_int("RT_CONV_EXACT_FP32", 0) == 0 &&
        env_int("RT_NO_CORR_MFMA_PLANAR", 0) == 0) {
        rt::CorrPlanarArgs a;
        a.left = static_cast<const float*>(left); a.right = static_cast<const float*>(right); a.out = static_cast<float*>(cv);
        a.C = C; a.H = H; a.W = W; a.D = D;
        a.blocks_x = (int)rt::cdiv(W, 32); a.batch = batch;
        const int64_t tasks = (int64_t)a.blocks_x * H * batch;
        RT_REQUIRE(rt::cdiv(tasks, 4) < (1ll << 31), "rt_corr_cost_volume: grid too large");
        const dim3 grid((unsigned)(rt::cdiv(rt::cdiv(tasks, 4), 8) * 8));        // a multiple of 8: the kernel renumbers workgroups per XCD
        hipLaunchKernelGGL(rt::corr_mfma_planar_kernel, grid, dim3(256), 0, S(s), a);
        RT_LAUNCH_CHECK("corr_mfma_planar_kernel");
        return 0;
    }
    return launch_corr<false, false>(static_cast<const float*>(left), static_cast<const float*>(right),
                                     static_cast<float*>(cv), batch, C, H, W, D, (int64_t)D * H * W, S(s));
}

extern "C" int rt_corr_softargmax_pitched(const void* left, const void* right, void* out, int batch, int C, int H,
                                          int W, int D, int is_min, int in_pitch, int out_pitch, int64_t out_bstride,
                                          int dtype, rtStream s) {
    RT_REQUIRE(left && right && out, "rt_corr_softargmax_pitched: null pointer");
    RT_REQUIRE(batch > 0 && C > 0 && H > 0 && W > 0 && D > 0, "rt_corr_softargmax_pitched: bad dims");
    RT_REQUIRE((in_pitch == 0 || in_pitch >= W) && (out_pitch == 0 || out_pitch >= W), "rt_corr_softargmax_pitched: pitch smaller than the row");
    if (dtype != RT_F32 && dtype != RT_F16) return fail(RT_E_UNSUPPORTED, "rt_corr_softargmax_pitched: bad dtype");
    if (out_bstride == 0) out_bstride = (int64_t)H * (out_pitch ? out_pitch : W);
    auto l = static_cast<const float*>(left);
    auto r = static_cast<const float*>(right);
    auto o = static_cast<float*>(out);
    if (dtype == RT_F16)        // fp16 NCHW feature maps and output (half2 mode of the executor)
        return is_min ? launch_corr<true, true, false, _Float16>(l, r, o, batch, C, H, W, D, out_bstride, S(s), in_pitch, out_pitch)
                      : launch_corr<true, false, false, _Float16>(l, r, o, batch, C, H, W, D, out_bstride, S(s), in_pitch, out_pitch);
    return is_min ? launch_corr<true, true>(l, r, o, batch, C, H, W, D, out_bstride, S(s), in_pitch, out_pitch)
                  : launch_corr<true, false>(l, r, o, batch, C, H, W, D, out_bstride, S(s), in_pitch, out_pitch);
}

// Correlation + soft-argmax on channel-interleaved (C/4, H, pitch, 4) fp32 feature maps, on the matrix cores (corr_mfma.hip.h)
extern "C" int rt_corr_softargmax_il(const void* left, const void* right, void* out, int batch, int C, int H, int W, int D,
                                     int is_min, int in_pitch, int out_pitch, int64_t out_bstride, rtStream s) {
    return rt_corr_softargmax_il_slot(left, right, out, batch, C, H, W, D, is_min, in_pitch, out_pitch, out_bstride, 1, s);
}
// ... writing the map as lane 0 of the 16-byte pixel slots of a channel-interleaved group (out_slot = 4): the 33rd channel of
// conv2D_1's input when the concatenation it belongs to is kept interleaved (resnet18_2D_513x257_net.cpp:601-615)
extern "C" int rt_corr_softargmax_il_slot(const void* left, const void* right, void* out, int batch, int C, int H, int W, int D,
                                          int is_min, int in_pitch, int out_pitch, int64_t out_bstride, int out_slot, rtStream s) {
    RT_REQUIRE(left && right && out, "rt_corr_softargmax_il: null pointer");
    RT_REQUIRE(out_slot == 1 || out_slot == 4, "rt_corr_softargmax_il_slot: out_slot is 1 (plane) or 4 (lane 0 of an interleaved group)");
    RT_REQUIRE(out_slot == 1 || (reinterpret_cast<uintptr_t>(out) % 16 == 0 && out_bstride % 4 == 0), "rt_corr_softargmax_il_slot: 16-byte slots need a 16-byte aligned output");
    RT_REQUIRE(batch > 0 && C > 0 && H > 0 && W > 0 && D > 0, "rt_corr_softargmax_il: bad dims");
    if (C % 4 != 0 || C > 32 || D > 64) return fail(RT_E_UNSUPPORTED, "rt_corr_softargmax_il: C must be a multiple of 4 up to 32 and max_disp <= 64 (C %d, D %d)", C, D);
    RT_REQUIRE((in_pitch == 0 || in_pitch >= W) && (out_pitch == 0 || out_pitch >= W), "rt_corr_softargmax_il: pitch smaller than the row");
    rt::CorrMfmaArgs a;
    a.left = static_cast<const float*>(left); a.right = static_cast<const float*>(right); a.out = static_cast<float*>(out);
    a.C = C; a.H = H; a.W = W; a.D = D;
    a.in_pitch = in_pitch ? in_pitch : W; a.out_pitch = out_pitch ? out_pitch : W;
    a.in_bstride = (int64_t)C * H * a.in_pitch;
    a.out_bstride = out_bstride ? out_bstride : (int64_t)H * a.out_pitch * out_slot;
    a.out_slot = out_slot;
    RT_REQUIRE(a.in_bstride < (1ll << 29), "rt_corr_softargmax_il: sample exceeds 2 GB (32-bit buffer offsets)");
    a.blocks_x = (int)rt::cdiv(W, 32); a.batch = batch;
    const int64_t tasks = (int64_t)a.blocks_x * H * batch;
    RT_REQUIRE(rt::cdiv(tasks, 4) < (1ll << 31), "rt_corr_softargmax_il: grid too large");
    dim3 grid((unsigned)rt::cdiv(tasks, 4));
    if (is_min) hipLaunchKernelGGL((rt::corr_softargmax_mfma_kernel<true>), grid, dim3(256), 0, S(s), a);
    else hipLaunchKernelGGL((rt::corr_softargmax_mfma_kernel<false>), grid, dim3(256), 0, S(s), a);
    RT_LAUNCH_CHECK("corr_softargmax_mfma_kernel");
    return 0;
}

// half2 mode: fp16 channel-interleaved feature maps (C/8, H, pitch, 8), fp16 map out (plane, out_pitch) -- corr_softargmax_mfma_f16_kernel
extern "C" int rt_corr_softargmax_il8_f16(const void* left, const void* right, void* out, int batch, int C, int H, int W, int D,
                                          int is_min, int in_pitch, int out_pitch, int64_t out_bstride, int out_slot, rtStream s) {
    RT_REQUIRE(left && right && out, "rt_corr_softargmax_il8_f16: null pointer");
    RT_REQUIRE(out_slot == 1 || out_slot == 8, "rt_corr_softargmax_il8_f16: out_slot is 1 (plane) or 8 (lane 0 of an interleaved group of 8)");
    RT_REQUIRE(out_slot == 1 || (reinterpret_cast<uintptr_t>(out) % 16 == 0 && out_bstride % 8 == 0), "rt_corr_softargmax_il8_f16: 16-byte slots need a 16-byte aligned output");
    RT_REQUIRE(batch > 0 && C > 0 && H > 0 && W > 0 && D > 0, "rt_corr_softargmax_il8_f16: bad dims");
    if (C % 8 != 0 || C > 32 || D > 64) return fail(RT_E_UNSUPPORTED, "rt_corr_softargmax_il8_f16: C must be a multiple of 8 up to 32 and max_disp <= 64 (C %d, D %d)", C, D);
    RT_REQUIRE((in_pitch == 0 || in_pitch >= W) && (out_pitch == 0 || out_pitch >= W), "rt_corr_softargmax_il8_f16: pitch smaller than the row");
    rt::CorrMfmaArgs a;
    a.left = static_cast<const float*>(left); a.right = static_cast<const float*>(right); a.out = static_cast<float*>(out);
    a.C = C; a.H = H; a.W = W; a.D = D;
    a.in_pitch = in_pitch ? in_pitch : W; a.out_pitch = out_pitch ? out_pitch : W;
    a.in_bstride = (int64_t)C * H * a.in_pitch;
    a.out_bstride = out_bstride ? out_bstride : (int64_t)H * a.out_pitch * out_slot;
    a.out_slot = out_slot;
    RT_REQUIRE(a.in_bstride < (1ll << 30), "rt_corr_softargmax_il8_f16: sample exceeds 2 GB (32-bit buffer offsets)");
    a.blocks_x = (int)rt::cdiv(W, 32); a.batch = batch;
    const int64_t tasks = (int64_t)a.blocks_x * H * batch;
    RT_REQUIRE(rt::cdiv(tasks, 4) < (1ll << 31), "rt_corr_softargmax_il8_f16: grid too large");
    dim3 grid((unsigned)rt::cdiv(tasks, 4));
    if (is_min) hipLaunchKernelGGL((rt::corr_softargmax_mfma_f16_kernel<true>), grid, dim3(256), 0, S(s), a);
    else hipLaunchKernelGGL((rt::corr_softargmax_mfma_f16_kernel<false>), grid, dim3(256), 0, S(s), a);
    RT_LAUNCH_CHECK("corr_softargmax_mfma_f16_kernel");
    return 0;
}

extern "C" int rt_corr_softargmax(const void* left, const void* right, void* out, int batch, int C, int H, int W,
                                  int D, int is_min, int64_t out_bstride, int dtype, rtStream s) {
    RT_REQUIRE(left && right && out, "rt_corr_softargmax: null pointer");
    RT_REQUIRE(batch > 0 && C > 0 && H > 0 && W > 0 && D > 0, "rt_corr_softargmax: bad dims");
    if (dtype != RT_F32) return fail(RT_E_UNSUPPORTED, "rt_corr_softargmax: only fp32 in this build");
    if (out_bstride == 0) out_bstride = (int64_t)H * W;
    // maps of a network's size: the Gram band on the matrix cores from planar maps (corr_softargmax_mfma_kernel<.., PLANAR>, the 3-term
    // fp16 split of the engines' correlation: |x| < 65504).  Small maps, D > 64, C > 32 and RT_CONV_EXACT_FP32 keep the fp32 fmaf kernel --
    // as does rt_corr_softargmax_pitched, the entry exact-fp32 engines call.
    if (C >= 16 && C <= 32 && W >= 64 && D <= 64 && (int64_t)C * H * W < (1ll << 29) && env_int("RT_CONV_EXACT_FP32", 0) == 0 &&
        env_int("RT_NO_CORR_MFMA_PLANAR", 0) == 0) {
        rt::CorrMfmaArgs a;
        a.left = static_cast<const float*>(left); a.right = static_cast<const float*>(right); a.out = static_cast<float*>(out);
        a.C = C; a.H = H; a.W = W; a.D = D;
        a.in_pitch = W; a.out_pitch = W;
        a.in_bstride = (int64_t)C * H * W;
        a.out_bstride = out_bstride;
        a.out_slot = 1;
        a.blocks_x = (int)rt::cdiv(W, 32); a.batch = batch;
        const int64_t tasks = (int64_t)a.blocks_x * H * batch;
        RT_REQUIRE(rt::cdiv(tasks, 4) < (1ll << 31), "rt_corr_softargmax: grid too large");
        dim3 grid((unsigned)rt::cdiv(tasks, 4));
        if (is_min) hipLaunchKernelGGL((rt::corr_softargmax_mfma_kernel<true, true>), grid, dim3(256), 0, S(s), a);
        else hipLaunchKernelGGL((rt::corr_softargmax_mfma_kernel<false, true>), grid, dim3(256), 0, S(s), a);
        RT_LAUNCH_CHECK("corr_softargmax_mfma_kernel<planar>");
        return 0;
    }
    auto l = static_cast<const float*>(left);
    auto r = static_cast<const float*>(right);
    auto o = static_cast<float*>(out);
    return is_min ? launch_corr<true, true>(l, r, o, batch, C, H, W, D, out_bstride, S(s))
                  : launch_corr<true, false>(l, r, o, batch, C, H, W, D, out_bstride, S(s));
}

extern "C" int rt_cost_volume(const void* left, const void* right, void* cv, int batch, int C, int H, int W, int D,
                              int dtype, rtStream s) {
    RT_REQUIRE(left && right && cv, "rt_cost_volume: null pointer");
    RT_REQUIRE(batch > 0 && C > 0 && H > 0 && W > 0 && D > 0, "rt_cost_volume: bad dims");
    RT_REQUIRE(H <= 65535 && (int64_t)batch * C <= 65535, "rt_cost_volume: grid too large");
    if (dtype != RT_F32) return fail(RT_E_UNSUPPORTED, "rt_cost_volume: only fp32 (as the reference, kernels.cu:140)");
    if (C % 4 == 0 && reinterpret_cast<uintptr_t>(cv) % 16 == 0 && (int64_t)H * W + 4 < 0x7fffffff && C <= 65535 && batch <= 65535 &&
        env_int("RT_NO_CV_X4", 0) == 0) {
        // 16-byte stores: four consecutive plane elements per thread, aligned in the output (cost_volume_f32x4_kernel)
        dim3 g4((unsigned)rt::cdiv(rt::cdiv((int64_t)H * W + 3, 4), 256), (unsigned)C, (unsigned)batch);
        hipLaunchKernelGGL(rt::cost_volume_f32x4_kernel, g4, dim3(256), 0, S(s), static_cast<const float*>(left),
                           static_cast<const float*>(right), static_cast<float*>(cv), C, H, W, D);
        RT_LAUNCH_CHECK("cost volume kernel");
        return 0;
    }
    dim3 grid((unsigned)rt::cdiv(W, 256), (unsigned)H, (unsigned)(batch * C));
    hipLaunchKernelGGL(rt::cost_volume_f32_kernel, grid, dim3(256), 0, S(s), static_cast<const float*>(left),
                       static_cast<const float*>(right), static_cast<float*>(cv), C, H, W, D);
    RT_LAUNCH_CHECK("cost volume kernel");
    return 0;
}

extern "C" int rt_softargmax(const void* vol, void* out, int batch, int D, int H, int W, int is_min, int dtype,
                             rtStream s) {
    RT_REQUIRE(vol && out, "rt_softargmax: null pointer");
    RT_REQUIRE(batch > 0 && D > 0 && H > 0 && W > 0 && batch <= 65535, "rt_softargmax: bad dims");
    if (dtype != RT_F32 && dtype != RT_F16) return fail(RT_E_UNSUPPORTED, "rt_softargmax: dtype %d", dtype);
    const int64_t hw = (int64_t)H * W;
    dim3 grid((unsigned)rt::cdiv(hw, 256), (unsigned)batch);
    if (dtype == RT_F16) {          // kHALF volumes, NCHW (softargmax_plugin.cpp:51-54)
        auto v = static_cast<const _Float16*>(vol);
        auto o = static_cast<_Float16*>(out);
        if (is_min) hipLaunchKernelGGL((rt::softargmax_f32_kernel<true, _Float16>), grid, dim3(256), 0, S(s), v, o, D, hw);
        else hipLaunchKernelGGL((rt::softargmax_f32_kernel<false, _Float16>), grid, dim3(256), 0, S(s), v, o, D, hw);
        RT_LAUNCH_CHECK("softargmax kernel (fp16)");
        return 0;
    }
    if (is_min)
        hipLaunchKernelGGL((rt::softargmax_f32_kernel<true>), grid, dim3(256), 0, S(s), static_cast<const float*>(vol),
                           static_cast<float*>(out), D, hw);
    else
        hipLaunchKernelGGL((rt::softargmax_f32_kernel<false>), grid, dim3(256), 0, S(s), static_cast<const float*>(vol),
                           static_cast<float*>(out), D, hw);
    RT_LAUNCH_CHECK("softargmax kernel");
    return 0;
}

// =================================================================================================
// layout glue
// =================================================================================================
extern "C" int rt_permute4d(const void* x, void* y, int batch, int d0, int d1, int d2, int d3, const int order[4],
                            int dtype, rtStream s) {
    RT_REQUIRE(x && y && order, "rt_permute4d: null pointer");
    RT_REQUIRE(batch > 0 && batch <= 65535 && d0 > 0 && d1 > 0 && d2 > 0 && d3 > 0, "rt_permute4d: bad dims");
    int seen = 0;
    for (int i = 0; i < 4; i++) {
        RT_REQUIRE(order[i] >= 0 && order[i] < 4, "rt_permute4d: bad order");
        seen |= 1 << order[i];
    }
    RT_REQUIRE(seen == 15, "rt_permute4d: order is not a permutation");
    const int in_dims[4] = {d0, d1, d2, d3};
    const int64_t in_str[4] = {(int64_t)d1 * d2 * d3, (int64_t)d2 * d3, (int64_t)d3, 1};
    const int64_t total = in_str[0] * d0;
    const int o[4] = {in_dims[order[0]], in_dims[order[1]], in_dims[order[2]], in_dims[order[3]]};
    if (order[3] == 3 && (dtype == RT_F32 || dtype == RT_F16)) {
        // the innermost dimension (with order[2] == 2 the innermost two) stays in place: contiguous runs move as they are
        const bool two = order[2] == 2;
        int64_t inner = two ? (int64_t)d2 * d3 : d3;
        const int o2 = two ? 1 : o[2];
        int64_t s0 = in_str[order[0]], s1 = in_str[order[1]], s2 = two ? 0 : in_str[order[2]], tot = total;
        const int64_t esz = dtype == RT_F32 ? 4 : 2;
        const int64_t runs = (int64_t)o[0] * o[1] * o2;
        // 16-byte units when every run starts on a 16-byte boundary in both tensors
        const bool v16 = (inner * esz) % 16 == 0 && (total * esz) % 16 == 0 && (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0;
        if (v16) { const int64_t f = 16 / esz; inner /= f; s0 /= f; s1 /= f; s2 /= f; tot /= f; }
        const int64_t chunks = rt::cdiv(inner, 256 * rt::kPermU);
        if (chunks * runs <= 0x7fffffff && chunks <= 0x7fffffff) {
            dim3 g((unsigned)(chunks * runs), (unsigned)batch);
            if (v16)
                hipLaunchKernelGGL((rt::permute_runs_kernel<rt::f32x4>), g, dim3(256), 0, S(s), static_cast<const rt::f32x4*>(x), static_cast<rt::f32x4*>(y),
                                   o[1], o2, s0, s1, s2, inner, (int)chunks, tot);
            else if (dtype == RT_F32)
                hipLaunchKernelGGL((rt::permute_runs_kernel<float>), g, dim3(256), 0, S(s), static_cast<const float*>(x), static_cast<float*>(y),
                                   o[1], o2, s0, s1, s2, inner, (int)chunks, tot);
            else
                hipLaunchKernelGGL((rt::permute_runs_kernel<_Float16>), g, dim3(256), 0, S(s), static_cast<const _Float16*>(x), static_cast<_Float16*>(y),
                                   o[1], o2, s0, s1, s2, inner, (int)chunks, tot);
            RT_LAUNCH_CHECK("permute kernel");
            return 0;
        }
    }
    const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(rt::cdiv(total, 256), 4096));
    dim3 grid(bx, (unsigned)batch);
    if (dtype == RT_F32)
        hipLaunchKernelGGL((rt::permute4d_kernel<float>), grid, dim3(256), 0, S(s), static_cast<const float*>(x),
                           static_cast<float*>(y), o[0], o[1], o[2], o[3], in_str[order[0]], in_str[order[1]],
                           in_str[order[2]], in_str[order[3]], total);
    else if (dtype == RT_F16)
        hipLaunchKernelGGL((rt::permute4d_kernel<_Float16>), grid, dim3(256), 0, S(s), static_cast<const _Float16*>(x),
                           static_cast<_Float16*>(y), o[0], o[1], o[2], o[3], in_str[order[0]], in_str[order[1]],
                           in_str[order[2]], in_str[order[3]], total);
    else
        return fail(RT_E_UNSUPPORTED, "rt_permute4d: dtype %d", dtype);
    RT_LAUNCH_CHECK("permute kernel");
    return 0;
}

extern "C" int rt_convert_format(const void* x, void* y, int batch, int C, int64_t inner, int src_kind, int dst_kind, rtStream s) {
    RT_REQUIRE(x && y && batch > 0 && C > 0 && inner > 0, "rt_convert_format: bad arguments");
    RT_REQUIRE(src_kind >= 0 && src_kind <= 2 && dst_kind >= 0 && dst_kind <= 2, "rt_convert_format: kinds are 0 (fp32 NCHW), 1 (fp16 NCHW), 2 (fp16 NC2HW2)");
    RT_REQUIRE(batch <= 65535 && (C + 1) / 2 <= 65535, "rt_convert_format: grid too large");
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(rt::cdiv(inner, 256), 4096)), (unsigned)((C + 1) / 2), (unsigned)batch);
    hipLaunchKernelGGL(rt::convert_format_kernel, grid, dim3(256), 0, S(s), x, y, C, inner, src_kind, dst_kind);
    RT_LAUNCH_CHECK("convert_format_kernel");
    return 0;
}

extern "C" int rt_pad_d(const void* x, void* y, int batch, int D, int64_t inner, int pad_end, int dtype, rtStream s) {
    RT_REQUIRE(x && y && batch > 0 && D > 0 && inner > 0 && pad_end >= 0, "rt_pad_d: bad arguments");
    return copy_rows(x, y, batch, D * inner, D * inner, (D + pad_end) * inner, pad_end * inner, dtype, S(s));
}

extern "C" int rt_slice_d(const void* x, void* y, int batch, int D, int64_t inner, int start, int end, int dtype,
                          rtStream s) {
    RT_REQUIRE(x && y && batch > 0 && inner > 0, "rt_slice_d: bad arguments");
    RT_REQUIRE(0 <= start && start < end && end <= D, "rt_slice_d: bad range [%d,%d) of %d", start, end, D);
    const char* src = static_cast<const char*>(x) + (size_t)start * inner * dsize(dtype);
    return copy_rows(src, y, batch, (end - start) * inner, D * inner, (end - start) * inner, 0, dtype, S(s));
}

extern "C" int rt_concat_channels(const void* x, void* y, int batch, int C, int Ctot, int c_off, int64_t inner,
                                  int dtype, rtStream s) {
    RT_REQUIRE(x && y && batch > 0 && C > 0 && inner > 0, "rt_concat_channels: bad arguments");
    RT_REQUIRE(c_off >= 0 && c_off + C <= Ctot, "rt_concat_channels: channel range out of bounds");
    char* dst = static_cast<char*>(y) + (size_t)c_off * inner * dsize(dtype);
    return copy_rows(x, dst, batch, C * inner, C * inner, Ctot * inner, 0, dtype, S(s));
}

// =================================================================================================
// image front-end / back-end (sample_app/main.cpp:83-98, 324-330)
// =================================================================================================
extern "C" int rt_preprocess_bgr8(const void* src, int src_h, int src_w, void* dst, int dst_h, int dst_w, int batch,
                                  rtStream s) {
    RT_REQUIRE(src && dst, "rt_preprocess_bgr8: null pointer");
    RT_REQUIRE(src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0 && batch > 0 && batch <= 65535 && dst_h <= 65535,
               "rt_preprocess_bgr8: bad dims");
    if (dst_h > src_h || dst_w > src_w)
        return fail(RT_E_UNSUPPORTED, "rt_preprocess_bgr8: INTER_AREA up-scaling (%dx%d -> %dx%d) is not implemented", src_w, src_h, dst_w, dst_h);
    if ((float)src_w / dst_w > 6.f || (float)src_h / dst_h > 6.f)
        return fail(RT_E_UNSUPPORTED, "rt_preprocess_bgr8: scale factors above 6 are not implemented");
    dim3 grid((unsigned)rt::cdiv(dst_w, 256), (unsigned)dst_h, (unsigned)batch);
    hipLaunchKernelGGL(rt::preprocess_bgr8_kernel, grid, dim3(256), 0, S(s), static_cast<const unsigned char*>(src), src_h,
                       src_w, static_cast<float*>(dst), dst_h, dst_w);
    RT_LAUNCH_CHECK("preprocess_bgr8_kernel");
    return 0;
}

extern "C" int rt_disparity_to_u16(const void* disp, void* out, int64_t n, float scale, rtStream s) {
    RT_REQUIRE(disp && out && n > 0, "rt_disparity_to_u16: bad arguments");
    hipLaunchKernelGGL(rt::disparity_u16_kernel, dim3((unsigned)rt::cdiv(n, 256)), dim3(256), 0, S(s),
                       static_cast<const float*>(disp), static_cast<unsigned short*>(out), n, scale);
    RT_LAUNCH_CHECK("disparity_u16_kernel");
    return 0;
}

// =================================================================================================
// convolutions
// =================================================================================================
namespace {


struct SubConv {                    // one launch of conv_mfma_f32_kernel
    int KH = 1, KW = 1, S = 1, NBW = 1, CC = 8;
    int TY = 4, TXW = 2, NW = 4, WLDS = 1;      // workgroup tile: TY rows x 32*TXW pixels, NW waves
    int CinPad = 0, Cout = 0;
    int Hi = 0, Wi = 0, Ho = 0, Wo = 0, pad_y = 0, pad_x = 0, nz = 1;
    int x_pitch = 0;                    // 0 = dense (Wi)
    std::vector<rt::ZSlice> zs_host;    // host copy of the ZSlice table (re-pitching)
    int64_t y_cstride = 0, y_zstride = 0, y_off = 0;
    int64_t r_cstride = 0;              // residual channel stride, 0 = y_cstride
    int y_ystride = 0, y_xstride = 1;
    float* w_dev = nullptr;
    int* choff_dev = nullptr;
    int w_exact = 0;                    // weights were given as fp16: split low parts are zero (conv_s3_kernel skips their MFMAs)
    int* shift_dev = nullptr;           // per gathered channel x-shift (folded cost volume), same shape as the gather table
    rt::ZSlice* zs_dev = nullptr;       // per-slice overrides (transposed-conv phases), or null
    int direct = 0, cin_real = 0;       // direct = VALU kernel for Cout <= 2
    int wino = 0;                       // Winograd F(2x2,3x3) kernel (stride-1 3x3 windows)
    int x_f16 = 0, y_f16 = 0;           // storage type of input / output + residual (half2 mode), set by rt_conv_plan_set_io_types
    int f16mma = 0;                     // conv_f16mma_kernel: fp16 operands on the matrix cores (both tensors fp16)
    int x_il8 = 0, y_il8 = 0, r_il8 = 0; // ... with channel-interleaved (C/8,H,pitch,8) input / output / residual tensors
    int dp4 = 0;                        // deconv_f16p_kernel: all four output phases of a stride-2 transposed 3x3 window per workgroup (f16mma, interleaved in / out)
    int f16first = 0;                   // conv_f16_first_kernel: 5x5 stride-2 first layer, fp32 image -> fp16 tensor on fp16 operands
    int split3 = 0;                     // conv_s3_kernel: fp32 tensors, 3-term fp16 split on the fp16 matrix pipe (general form)
    int rb = 0;                         // conv_s3rb_kernel: fused residual block (two 3x3 convolutions), rtConvPlan::rb_*
    int s3first = 0;                    // conv_s3_first_kernel: 5x5 stride-2 first layer (<= 3 input channels), split fp16, row-as-contraction
    int s3p = 0;                        // conv_s3p_kernel: fp32 tensors, 3-term fp16 split on the fp16 matrix pipe, persistent (3x3 s1, Cin, Cout <= 32)
    int small3d = 0;                    // deconv3d_s2_small_kernel (stride-2 transposed 3x3x3 (1) / 3x3 (2), <= 2 output channels)
    rt::Deconv3dSmallArgs s3{};         // its geometry (pointers filled at enqueue)
    std::vector<float> small_w;         // ... its weights as [K][COUT][phase 8][neighbour 8] before structural zeros are dropped (3-D form)
    void* small_il_dev = nullptr;       // deconv3d_s2_il_kernel / _il4_kernel: the MFMA A operands (set when the input becomes channel-interleaved)
    int small_il_f32 = 0;               // ... built for the fp32 form (hi and lo slabs)
};

}  // namespace

// The first Conv3D over a folded default cost volume, factored into two 2-D convolutions and one combining pass (fold_factor.hip.h)
struct FoldFactor {
    rtConvPlan* pl = nullptr;          // conv3x3 F -> 3K on the left feature map: A_first, A_middle, A_last
    rtConvPlan* pr = nullptr;          // conv3x3 F -> 3K on [0 | right feature map]: C'_0, C'_1, C'_2
    float* wedge_dev = nullptr;        // [j][dy][c][k]: the taps dx = +1 of the right half (edge term)
    int F = 0, K = 0, D = 0, H = 0, W = 0;
    int64_t a_elems = 0, c_elems = 0, t_elems = 0, e_elems = 0;      // per sample
    // Scratch (A, C', T, E: rt_conv_plan_workspace_bytes) belongs to the CALLER: an execution context passes its own workspace
    // (rt_conv_enqueue_ws; IPlugin::getWorkspaceSize / enqueue's `workspace`, as the reference's plugins receive theirs,
    // lib/conv3d_plugin.cpp:179-185).  rt_conv_enqueue without one -- the operator-level tests and tools -- falls back to blocks owned by
    // the plan, one per STREAM it is enqueued on (launches of one stream are ordered; two streams never share scratch).
    std::mutex mu;
    std::map<rtStream, std::pair<void*, size_t>> own;      // the fallback blocks: one per stream the plan was enqueued on without a workspace
};

struct rtConvPlan {
    std::vector<SubConv> subs;
    FoldFactor* ff = nullptr;
    // launch-time knobs of the environment (A/B and test switches), read once at the plan's first enqueue: getenv walks the
    // whole environment, and five look-ups per launch were a third of the host's time per launch
    mutable std::once_flag env_once;   // the launch-time knobs below are read once per plan, by whichever context launches it first
    int softarg = 0;              // rt_conv_plan_set_softarg: 1 / 2 = the launch ends in a soft-argmax / soft-argmin over the output depth
    mutable int opt_xcd = 1, opt_trace = 0, opt_rb_tiles = 0, opt_rbs_seg = 0, opt_s3p_grid = 0, opt_ksplit = -1, opt_r4 = -1, opt_zinner = 1, opt_nbinner = 1, opt_dw = -1, opt_dw_nseg = 0, opt_small_walk = -1, opt_fold_u = 2, opt_f16p_walk = -1, opt_f16p_classes = 1;
    float* bias_dev = nullptr;
    float* zeros_dev = nullptr;
    int act = 0, has_resid = 0, dtype = RT_F32;
    int out_dims[4] = {0, 0, 0, 1};
    int64_t x_bstride = 0, y_bstride = 0;
    int64_t r_bstride = 0;                        // residual per-sample stride, 0 = y_bstride
    int is2d = 0, cin = 0, hin = 0, win = 0;      // 2-D plans can be re-pitched (rt_conv_plan_set_pitch)
    std::vector<float> w_canon;                   // 2-D plans: weights as given (KCRS / (Cin,Cout,R,S)), for re-packing
    rtConv2dDesc desc2d{};                        // ... and the descriptor they came with
    int is_deconv = 0;
    int in_pitch = 0, out_pitch = 0;
    // fused residual block (rt_resblock_plan_create): the first convolution's split weights, bias, activation
    float* rb_w1_dev = nullptr;
    float* rb_bias1_dev = nullptr;
    int rb_act1 = 0, rb_cmid = 0;
    // ... and, for the tower form (32 -> 32 -> 32), both convolutions' slabs with the output channels in conv_s3rbd_kernel's row order,
    // and whether the block's input / output is a pre-split tensor (rt_resblock_plan_set_split)
    float* rbd_w1_dev = nullptr;
    float* rbd_w2_dev = nullptr;
    int x_split = 0, y_split = 0;
    // ... and as fp16 operands, same row order, for the half2 form of the block (conv_f16rbd_kernel: both tensors fp16)
    float* rbh_w1_dev = nullptr;
    float* rbh_w2_dev = nullptr;
    int w_f16 = 0;                                // the weights were given as fp16 (trt_weights_fp16.bin)
    // 3-D plans: what rt_conv_plan_set_io_types / _supports_il8 need to know (Conv3D: w_canon holds the weights as (K, V*C, R, S))
    int is_conv3d = 0, is_deconv3d = 0;
    int c3d_C = 0, c3d_cin = 0, c3d_dchw = 0, c3d_fold = 0;
    int c3d_dw = 0;                               // 3x3x3, stride 1, pad 1, C % 16 == 0, no folded pad / cost volume: the depth-walking kernels apply (conv_f16dw.hip.h)
    rtConv3dDesc desc3d{};                        // transposed 3-D plans: the descriptor and input dims they were created with (the launches are
    int in_dims3[3] = {0, 0, 0};                  // rebuilt when the plan moves between the split-fp16 and the fp16-operand kernel)
    int flags = 0;                                // RT_CONV_* option bits of the descriptor(s) the plan was created with: every later re-planning
                                                  // (set_io_types, set_layouts, set_pitch) runs under the same options (ExactScope)
};

namespace {

// 1-D decomposition of a transposed convolution (scatter: o = i*s + r - p) into `s` dense
// correlations, one per output phase phi = o mod s:  i = m + u - pad,  r = tap[u]  (o = m*s + phi).
struct Phase1D {
    int K = 0;          // taps in the window (0: this phase receives no contribution)
    int pad = 0;
    int tap[8] = {0};
};
Phase1D phase1d(int s, int p, int k, int phi) {
    Phase1D ph;
    int r0 = -1;
    for (int r = k - 1; r >= 0; r--)
        if (((r - (phi + p)) % s + s) % s == 0) { r0 = r; break; }
    if (r0 < 0) return ph;
    // base = (phi + p - r0) / s  (exact);  i = m + base + u
    ph.pad = -((phi + p - r0) / s);
    for (int r = r0; r >= 0; r -= s) ph.tap[ph.K++] = r;
    return ph;
}

// ---- 3-term fp16 split (conv_split.hip.h) ---------------------------------------------------------------------------
// w = wh + wl * 2^-11 with wh = fp16(w), wl = fp16((w - wh) * 2^11)
void split_f16(float w, uint16_t& hi, uint16_t& lo) {
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)((w - (float)h) * rt::kSplitScale);
    std::memcpy(&hi, &h, 2);
    std::memcpy(&lo, &l, 2);
}

// Appends the weights of one launch (or of one phase of a multi-phase launch) to `packed`, in the exact
// order of the kernel's LDS slab: [nblk][chunk][tap][h][NB][CC/2] with gathered channel
// ci = chunk*CC + 2*j + h; wfun(co, ci, u, v) = weight of output channel co, gathered channel ci, window
// tap (u, v).  The direct (Cout <= 2) kernel takes plain [co][ci][tap].  Returns the element offset.
template <typename F>
int64_t pack_into(std::vector<float>& packed, const SubConv& sc, int cin_real, F wfun) {
    const int64_t base = (int64_t)packed.size();
    const int taps = sc.KH * sc.KW;
    if (sc.direct) {
        packed.resize(base + (size_t)sc.Cout * cin_real * taps, 0.f);
        for (int co = 0; co < sc.Cout; co++)
            for (int ci = 0; ci < cin_real; ci++)
                for (int u = 0; u < sc.KH; u++)
                    for (int v = 0; v < sc.KW; v++)
                        packed[base + ((size_t)co * cin_real + ci) * taps + u * sc.KW + v] = wfun(co, ci, u, v);
        return base;
    }
    if (sc.split3) {
        // split fp16 slabs of conv_s3_kernel: [nblk of 32 co][chunk of 16 ci][tap][hi / lo][k-group of 8][co % 32][8 halfs],
        // two halfs per float of `packed`; the returned offset counts 16-byte slots
        const int nblk = (int)rt::cdiv(sc.Cout, 32), nch = sc.CinPad / 16;
        const size_t slots = (size_t)nblk * nch * taps * 2 * 2 * 32;
        packed.resize(base + slots * 4, 0.f);
        uint16_t* dst = reinterpret_cast<uint16_t*>(packed.data() + base);
        for (int co = 0; co < sc.Cout; co++)
            for (int ci = 0; ci < cin_real; ci++)
                for (int u = 0; u < sc.KH; u++)
                    for (int v = 0; v < sc.KW; v++) {
                        uint16_t hi, lo;
                        split_f16(wfun(co, ci, u, v), hi, lo);
                        const int nb = co / 32, cc = co % 32, ch = ci / 16, kg = (ci % 16) / 8, e = ci % 8, t = u * sc.KW + v;
                        const size_t slab = ((size_t)nb * nch + ch) * taps + t;
                        dst[(((slab * 2 + 0) * 2 + kg) * 32 + cc) * 8 + e] = hi;
                        dst[(((slab * 2 + 1) * 2 + kg) * 32 + cc) * 8 + e] = lo;
                    }
        return base / 4;
    }
    if (sc.wino) {
        // U = G g G^T per (co, ci), G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; slab order of conv_wino_f32_kernel:
        // [nblk of 32 co][chunk of 8 ci][k-step j][k4][p/4][co % 32][p % 4],  ci = chunk*8 + 4*j + k4, p = 4*pr + pc
        static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
        const int nblk = (int)rt::cdiv(sc.Cout, 32), nch = sc.CinPad / 8;
        packed.resize(base + (size_t)nblk * nch * 4096, 0.f);
        for (int co = 0; co < sc.Cout; co++)
            for (int ci = 0; ci < cin_real; ci++) {
                double g[3][3], tmp[4][3];
                for (int u = 0; u < 3; u++)
                    for (int v = 0; v < 3; v++) g[u][v] = wfun(co, ci, u, v);
                for (int a = 0; a < 4; a++)
                    for (int v = 0; v < 3; v++) tmp[a][v] = G[a][0] * g[0][v] + G[a][1] * g[1][v] + G[a][2] * g[2][v];
                const int nb = co / 32, cc = co % 32, ch = ci / 8, j = (ci % 8) / 4, k4 = ci % 4;
                for (int pr = 0; pr < 4; pr++)
                    for (int pc = 0; pc < 4; pc++) {
                        const double u = tmp[pr][0] * G[pc][0] + tmp[pr][1] * G[pc][1] + tmp[pr][2] * G[pc][2];
                        const int pp = 4 * pr + pc;
                        packed[base + ((size_t)nb * nch + ch) * 4096 + ((((size_t)j * 4 + k4) * 4 + pp / 4) * 32 + cc) * 4 + pp % 4] = (float)u;
                    }
            }
        return base;
    }
    const int NB = 32 * sc.NBW, cpg = sc.CC / 2;
    const int nblk = (int)rt::cdiv(sc.Cout, NB), nch = sc.CinPad / sc.CC;
    packed.resize(base + (size_t)nblk * nch * taps * sc.CC * NB, 0.f);
    for (int nb = 0; nb < nblk; nb++)
        for (int ch = 0; ch < nch; ch++)
            for (int u = 0; u < sc.KH; u++)
                for (int v = 0; v < sc.KW; v++)
                    for (int h = 0; h < 2; h++)
                        for (int nn = 0; nn < NB; nn++) {
                            const int co = nb * NB + nn;
                            if (co >= sc.Cout) continue;
                            float* dst = &packed[base + (((((size_t)nb * nch + ch) * taps + (u * sc.KW + v)) * 2 + h) * NB + nn) * cpg];
                            for (int j = 0; j < cpg; j++) {
                                const int ci = ch * sc.CC + 2 * j + h;
                                if (ci < cin_real) dst[j] = wfun(co, ci, u, v);
                            }
                        }
    return base;
}

int upload_weights(SubConv& sc, const std::vector<float>& packed) {
    RT_HIP(hipMalloc((void**)&sc.w_dev, packed.size() * sizeof(float)));
    RT_HIP(hipMemcpy(sc.w_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

template <typename F>
int upload_packed(SubConv& sc, int cin_real, F wfun) {
    std::vector<float> packed;
    pack_into(packed, sc, cin_real, wfun);
    return upload_weights(sc, packed);
}

// fp16 operand slabs of conv_f16mma_kernel: [nblk][chunk of 16 ci][tap][h][co % 32][8 halfs], ci = chunk*16 + 8*h + e.
// Appends one slab set (all nblk x chunks) and returns its offset in 16-byte slots.
template <typename F>
int64_t pack_f16_into(std::vector<uint16_t>& packed, const SubConv& sc, int cin_real, F wfun) {
    const int64_t base = (int64_t)packed.size() / 8;
    const int taps = sc.KH * sc.KW, nblk = (int)rt::cdiv(sc.Cout, 32), nch = sc.CinPad / 16;
    packed.resize(packed.size() + (size_t)nblk * nch * taps * 2 * 32 * 8, 0);
    for (int co = 0; co < sc.Cout; co++)
        for (int ci = 0; ci < cin_real; ci++)
            for (int u = 0; u < sc.KH; u++)
                for (int v = 0; v < sc.KW; v++) {
                    const _Float16 hv = (_Float16)wfun(co, ci, u, v);
                    uint16_t bits;
                    std::memcpy(&bits, &hv, 2);
                    const int nb = co / 32, cc = co % 32, ch = ci / 16, h = (ci % 16) / 8, e = ci % 8;
                    packed[(size_t)base * 8 + ((((((size_t)nb * nch + ch) * taps + (u * sc.KW + v)) * 2 + h) * 32 + cc) * 8) + e] = bits;
                }
    return base;
}

int upload_zslices(SubConv& sc, const std::vector<rt::ZSlice>& zs) {
    RT_HIP(hipMalloc((void**)&sc.zs_dev, zs.size() * sizeof(rt::ZSlice)));
    RT_HIP(hipMemcpy(sc.zs_dev, zs.data(), zs.size() * sizeof(rt::ZSlice), hipMemcpyHostToDevice));
    return 0;
}

int upload_table(SubConv& sc, const std::vector<int>& table) {
    RT_HIP(hipMalloc((void**)&sc.choff_dev, table.size() * sizeof(int)));
    RT_HIP(hipMemcpy(sc.choff_dev, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

// bias is always materialised, zero padded to a multiple of 64 channels (the epilogue reads float4s);
// the same allocation carries 64 trailing zeros that out-of-image gathers are redirected to.
int upload_bias(rtConvPlan* plan, const float* bias, int n) {
    const int padded = rt::round_up(n, 64);
    std::vector<float> b(padded + 64, 0.f);
    if (bias) std::memcpy(b.data(), bias, n * sizeof(float));
    RT_HIP(hipMalloc((void**)&plan->bias_dev, b.size() * sizeof(float)));
    RT_HIP(hipMemcpy(plan->bias_dev, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    plan->zeros_dev = plan->bias_dev + padded;
    return 0;
}

bool window_supported(int KH, int KW, int S) {
    if (S == 1) return (KH == 3 && KW == 3) || (KH <= 2 && KW <= 2 && KH >= 1 && KW >= 1);
    if (S == 2) return (KH == 3 && KW == 3) || (KH == 5 && KW == 5);
    return false;
}

// Workgroup tile selection.  Candidates (TY rows x 32*TXW pixels, NW waves):
//   {4,2,4}: most reuse per staged byte;  {4,1,4} / {2,2,4}: one 32-pixel wave-tile per wave -> fewer
//   registers (5 waves/SIMD), twice the workgroups;  {2,1,2}: smallest, for layers that cannot fill 256 CUs.
// Measured on MI355X (tools/layer_profile.py): the whole network is fastest with one wave-tile per wave,
// RT_CONV_VARIANT / RT_CONV_NBW override the heuristic for A/B measurements.
// Development knobs (RT_* environment variables: A/B switches of tests, tools and DESIGN.md's measurements) are honoured ONLY when the
// process opts in with RT_DEV_KNOBS=1 -- tests/conftest.py, tools/ and the profile scripts do; a stray RT_NO_FUSION in a user's
// environment does not change what the shipped library computes.  Options a user may want are API: rtConv2dDesc::flags,
// rt_conv_enqueue_hint, IBuilder / IExecutionContext setters, rtNetOptions.
bool dev_knobs() {
    static const bool on = [] { const char* e = getenv("RT_DEV_KNOBS"); return e && atoi(e) != 0; }();
    return on;
}
int env_int(const char* name, int dflt) {
    if (!dev_knobs()) return dflt;
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Kernel families that were measured and rejected (the persistent split kernel, the per-tile fused residual block, 8-row /
// 8-wave tiles, the Winograd kernel on interleaved tensors) are compiled only with -DRT_EXPERIMENTAL -- the emulator build of the CPU
// test tier has it, the product library does not -- and the development knobs that select them read as their defaults otherwise.
#ifdef RT_EXPERIMENTAL
inline int exp_knob(const char* name, int dflt) { return env_int(name, dflt); }
#else
inline int exp_knob(const char*, int dflt) { return dflt; }
#endif
// RT_CONV_EXACT_FP32 as an API option: set from the descriptor's flags for the duration of one *_plan_create call
thread_local int tl_exact_fp32 = 0;
int exact_fp32() { return tl_exact_fp32 || env_int("RT_CONV_EXACT_FP32", 0) != 0; }
struct ExactScope {
    int prev;
    explicit ExactScope(int flags) : prev(tl_exact_fp32) { tl_exact_fp32 = (flags & RT_CONV_EXACT_FP32) ? 1 : prev; }
    ~ExactScope() { tl_exact_fp32 = prev; }
};

void set_tile(SubConv& sc, int variant) {
    switch (variant) {
        case 0: sc.TY = 4; sc.TXW = 2; sc.NW = 4; break;
        case 4: sc.TY = 2; sc.TXW = 2; sc.NW = 4; break;
        case 9: sc.TY = 2; sc.TXW = 1; sc.NW = 2; break;
        case 8: sc.TY = 8; sc.TXW = 1; sc.NW = 8; break;
        case 10: sc.TY = 8; sc.TXW = 1; sc.NW = 4; break;
        default: sc.TY = 4; sc.TXW = 1; sc.NW = 4; break;     // 6
    }
    sc.WLDS = 1;
}

int64_t count_wgs(const SubConv& sc) {
    return rt::cdiv(sc.Ho, sc.TY) * rt::cdiv(sc.Wo, 32 * sc.TXW) * rt::cdiv(sc.Cout, 32 * sc.NBW) * sc.nz;
}

// the direct kernel caches all weights of a slice in LDS and exists for a fixed set of windows
void check_direct(SubConv& sc, int cin_real) {
    const bool win = (sc.KH == sc.KW && (sc.KH == 1 || sc.KH == 2 || sc.KH == 3 || sc.KH == 5)) ||
                     (sc.KH == 1 && sc.KW == 2) || (sc.KH == 2 && sc.KW == 1);
    if (!win || (int64_t)sc.Cout * cin_real * sc.KH * sc.KW > 4096) sc.direct = 0;
}

void choose_tiling(SubConv& sc, bool allow_wino = true) {
    sc.direct = (sc.Cout <= 2 && env_int("RT_CONV_NO_DIRECT", 0) == 0) ? 1 : 0;    // re-checked against the LDS weight cache below
    sc.CC = 8;
    // one 32-channel block per workgroup (NBW = 1): the 64-channel block halves the patch staging per FLOP but
    // also halves the workgroup count, and every layer of the Stereo DNN graphs with Cout >= 64 runs at
    // 1/4 .. 1/8 resolution where workgroups are scarce (measured: 128->128 @47x158 45.8 -> 29.0 us,
    // 64->128 stride 2 28.6 -> 19.4 us, whole network +6 %).  The 2-wave {2,1,2} tile is slower everywhere.
    sc.NBW = 1;
    if (sc.KH == 5) sc.CC = 4;
    const int forced = env_int("RT_CONV_VARIANT", -1);
    set_tile(sc, forced >= 0 ? forced : 6);
    // Winograd F(2x2,3x3) for the stride-1 3x3 windows with at least one full pair of 16-channel blocks
    sc.wino = (allow_wino && !sc.direct && sc.KH == 3 && sc.KW == 3 && sc.S == 1 && sc.Cout >= 24 && env_int("RT_CONV_NO_WINO", 0) == 0) ? 1 : 0;
    if (sc.wino) { sc.CC = 8; sc.NBW = 1; sc.TY = exp_knob("RT_WINO_WAVES", 4); sc.TXW = 1; sc.NW = sc.TY; }
    const int nbw = env_int("RT_CONV_NBW", 0);
    if (nbw == 1 || (nbw == 2 && sc.KH != 5)) sc.NBW = nbw;
    // fp32 tensors on the fp16 matrix pipe (3-term split, conv_split.hip.h) for every window that kernel is built for;
    // RT_CONV_EXACT_FP32=1 keeps the fp32 fmaf-chain kernels
    const bool s3_win = (sc.KH == 3 && sc.KW == 3 && (sc.S == 1 || sc.S == 2)) || (sc.S == 1 && sc.KH <= 2 && sc.KW <= 2);
    sc.split3 = (!sc.direct && s3_win && !exact_fp32() && env_int("RT_NO_S3", 0) == 0) ? 1 : 0;
    if (sc.split3) {
        sc.wino = 0; sc.CC = 16; sc.NBW = 1; sc.TY = 4; sc.TXW = 1; sc.NW = 4;
        // 3x3 stride 1: 8-row tiles (8 waves, weights staged once per 8 rows, less halo) are built and opt-in (RT_S3_ROWS=8):
        // measured on the 32->32 @629x185 layer 13.9 vs 14.5 us alone, but 2055 vs 2108 pairs/s in the network (4 contexts)
        if (sc.KH == 3 && sc.KW == 3 && sc.S == 1 && exp_knob("RT_S3_ROWS", 4) == 8) sc.TY = sc.NW = 8;
    }
}

// The persistent form is opt-in (RT_S3P=1).  Measured on MI355X, ResNet-18 2D 1257x369 (profiles/README.md, round 2): alone it
// runs the 32->32 layer in 13.5 us against 14.1 us for the general kernel, but one workgroup per CU holding 135 KB of LDS cannot
// share a CU with the other stream's / other contexts' launches, so the network is slower with it (1830 vs 2081 pairs/s).
bool s3p_eligible(const SubConv& sc, int cin) {
    return !sc.direct && sc.KH == 3 && sc.KW == 3 && sc.S == 1 && cin <= 32 && sc.Cout <= 32 && sc.nz == 1 &&
           !exact_fp32() && exp_knob("RT_S3P", 0) != 0;
}

// conv_s3p_kernel's LDS image of the layer's weights: [tap][chunk of 16 ci][hi / lo][k-group of 8][co % 32][8 halfs]
template <typename F>
int upload_s3p(SubConv& sc, int cin_real, F wfun) {
    std::vector<uint16_t> packed((size_t)rt::S3PCfg<8>::W_SLOTS * 8, 0);
    for (int co = 0; co < sc.Cout; co++)
        for (int ci = 0; ci < cin_real; ci++)
            for (int t = 0; t < 9; t++) {
                uint16_t hi, lo;
                split_f16(wfun(co, ci, t / 3, t % 3), hi, lo);
                const int c = ci / 16, kg = (ci % 16) / 8, e = ci % 8;
                packed[(((((size_t)t * 2 + c) * 2 + 0) * 2 + kg) * 32 + co) * 8 + e] = hi;
                packed[(((((size_t)t * 2 + c) * 2 + 1) * 2 + kg) * 32 + co) * 8 + e] = lo;
            }
    if (sc.w_dev) (void)hipFree(sc.w_dev);
    sc.w_dev = nullptr;
    RT_HIP(hipMalloc((void**)&sc.w_dev, packed.size() * 2));
    RT_HIP(hipMemcpy(sc.w_dev, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    sc.s3p = 1; sc.split3 = 0; sc.wino = 0;
    sc.CinPad = 32; sc.CC = 16; sc.NBW = 1; sc.TXW = 1; sc.TY = sc.NW = 8;
    return 0;
}

// conv_s3_first_kernel: per 32-channel block the 10 A operands [r][hi / lo][k-half][co][8 halfs], k = 3*s + c
bool s3first_eligible(const SubConv& sc, int cin, int has_resid) {
    return !sc.direct && sc.KH == 5 && sc.KW == 5 && sc.S == 2 && cin <= 3 && !has_resid && sc.nz == 1 &&
           !exact_fp32() && env_int("RT_NO_S3", 0) == 0;
}
template <typename F>
int upload_s3first(SubConv& sc, int cin_real, F wfun) {
    const int nblk = (int)rt::cdiv(sc.Cout, 32);
    std::vector<uint16_t> packed((size_t)nblk * rt::S3FirstCfg::W_SLOTS * 8, 0);
    for (int co = 0; co < sc.Cout; co++)
        for (int c = 0; c < cin_real; c++)
            for (int r = 0; r < 5; r++)
                for (int sx = 0; sx < 5; sx++) {
                    uint16_t hi, lo;
                    split_f16(wfun(co, c, r, sx), hi, lo);
                    const int k = 3 * sx + c, h = k / 8, e = k % 8;
                    const size_t blk = (size_t)(co / 32) * rt::S3FirstCfg::W_SLOTS * 8;
                    packed[blk + ((((size_t)r * 2 + 0) * 2 + h) * 32 + co % 32) * 8 + e] = hi;
                    packed[blk + ((((size_t)r * 2 + 1) * 2 + h) * 32 + co % 32) * 8 + e] = lo;
                }
    if (sc.w_dev) (void)hipFree(sc.w_dev);
    sc.w_dev = nullptr;
    RT_HIP(hipMalloc((void**)&sc.w_dev, packed.size() * 2));
    RT_HIP(hipMemcpy(sc.w_dev, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    sc.s3first = 1; sc.s3p = 0; sc.split3 = 0; sc.wino = 0;
    sc.NBW = 1; sc.TXW = 1; sc.TY = sc.NW = 4;
    return 0;
}

// deconv3d_s2_small_kernel: weights packed [.. ][phase][neighbour]; when every pair outside rt::SmallTaps' pattern is zero (the padding of
// every network of the reference; PAT: rt::SmallTaps) keep only the pairs that carry a tap.  Returns whether it did (RT_NO_SMALL_SPARSE:
// development knob).
template <bool Z, int PAT>
bool drop_structural_zeros(std::vector<float>& packed) {
    using Taps = rt::SmallTaps<Z, PAT>;
    constexpr int NJ = Taps::NJ;
    if (env_int("RT_NO_SMALL_SPARSE", 0) != 0) return false;
    const size_t rows = packed.size() / (NJ * NJ);
    for (size_t r = 0; r < rows; r++)
        for (int f = 0; f < NJ; f++)
            for (int j = 0; j < NJ; j++)
                if (!Taps::valid(f, j) && packed[(r * NJ + f) * NJ + j] != 0.f) return false;
    std::vector<float> dense(rows * Taps::NV);
    for (size_t r = 0; r < rows; r++)
        for (int f = 0; f < NJ; f++)
            for (int j = 0; j < NJ; j++)
                if (Taps::valid(f, j)) dense[r * Taps::NV + Taps::index(f, j)] = packed[(r * NJ + f) * NJ + j];
    packed.swap(dense);
    return true;
}

// conv_s3_kernel with ks contraction groups per workgroup (clamped to what the instantiation's LDS and register budget allow)
template <int KH, int KW, int S, bool XIL, bool YIL, int NW = 4, typename TIN = float, typename TOUT = float>
void launch_s3(dim3 grid, int ks, int64_t per_cu, hipStream_t st, const rt::ConvArgs& a) {
    using Cfg = rt::S3Cfg<KH, KW, S, NW>;
    ks = Cfg::max_groups(ks);                                  // per_cu: a sample's workgroups per CU; co-resident ones share 160 KB
    while (ks > 1 && std::min<int64_t>(per_cu, 4) * ks * Cfg::GRP_BYTES > 160 * 1024) ks = Cfg::max_groups(ks - 1);
#ifndef HIPEMU
    // dynamic LDS beyond the default limit: a function attribute is per device (the one-process multi-GPU app), set once each
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !((done.load(std::memory_order_acquire) >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt::conv_s3_kernel<KH, KW, S, XIL, YIL, NW, TIN, TOUT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::KS_MAX * Cfg::GRP_BYTES);
        done.fetch_or(1ull << dev, std::memory_order_release);
    }
#endif
    hipLaunchKernelGGL((rt::conv_s3_kernel<KH, KW, S, XIL, YIL, NW, TIN, TOUT>), grid, dim3(64 * NW * ks), (size_t)ks * Cfg::GRP_BYTES, st, a);
}

// compute units of the CALLING THREAD's current device (the one-process, one-thread-per-device app runs plans of several devices)
int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// weights to host fp32 (fp16 weight files are widened once; activations stay fp32 in this build)
std::vector<float> to_f32(const void* w, size_t n, int dtype) {
    std::vector<float> out(n);
    if (dtype == RT_F16) {
        // the blob of a weight file is not 2-byte aligned in general (name\0, count, data): no typed loads from it
        const char* h = static_cast<const char*>(w);
        for (size_t i = 0; i < n; i++) {
            _Float16 v;
            std::memcpy(&v, h + 2 * i, 2);
            out[i] = (float)v;
        }
    } else {
        std::memcpy(out.data(), w, n * sizeof(float));
    }
    return out;
}

template <int KH, int KW, int S, int NBW, int CC>
int launch_window(const SubConv& sc, const rt::ConvArgs& a, dim3 grid, hipStream_t st) {
#define RT_TILE(ty, txw, nw, wlds)                                                                        \
    if (sc.TY == ty && sc.TXW == txw && sc.NW == nw && sc.WLDS == (wlds ? 1 : 0)) {                       \
        hipLaunchKernelGGL((rt::conv_mfma_f32_kernel<KH, KW, S, ty, txw, NBW, CC, nw, wlds>), grid,       \
                           dim3(64 * nw), 0, st, a);                                                      \
        return 0;                                                                                         \
    }
    if (sc.y_il8) return 1;          // no interleaved-output form of the direct kernel (removed in round 2)
    // fp16 storage (half2 mode): the default tile only; fp32 in -> fp16 out (first layer) or fp16 -> fp16
    if (sc.x_f16 || sc.y_f16) {
        if (!(sc.TY == 4 && sc.TXW == 1 && sc.NW == 4 && sc.WLDS) || !sc.y_f16) return 1;
        if (sc.x_f16)
            hipLaunchKernelGGL((rt::conv_mfma_f32_kernel<KH, KW, S, 4, 1, NBW, CC, 4, true, _Float16, _Float16>), grid, dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((rt::conv_mfma_f32_kernel<KH, KW, S, 4, 1, NBW, CC, 4, true, float, _Float16>), grid, dim3(256), 0, st, a);
        return 0;
    }
    RT_TILE(4, 2, 4, true) RT_TILE(4, 1, 4, true) RT_TILE(2, 2, 4, true) RT_TILE(2, 1, 2, true) RT_TILE(8, 1, 8, true) RT_TILE(8, 1, 4, true)
#undef RT_TILE
    return 1;
}

int launch_sub(const SubConv& sc, const rt::ConvArgs& a, dim3 grid, hipStream_t st) {
#define RT_CASE(kh, kw, s, nbw, cc)                                                             \
    if (sc.KH == kh && sc.KW == kw && sc.S == s && sc.NBW == nbw && sc.CC == cc) {              \
        if (launch_window<kh, kw, s, nbw, cc>(sc, a, grid, st))                                 \
            return fail(RT_E_UNSUPPORTED, "conv: tile %dx%d/%d waves not instantiated", sc.TY, 32 * sc.TXW, sc.NW); \
        RT_LAUNCH_CHECK("conv_mfma_f32_kernel<" #kh "," #kw "," #s "," #nbw "," #cc ">");       \
        return 0;                                                                               \
    }
    RT_CASE(3, 3, 1, 1, 8) RT_CASE(3, 3, 1, 2, 8)
    RT_CASE(3, 3, 2, 1, 8) RT_CASE(3, 3, 2, 2, 8)
    RT_CASE(5, 5, 2, 1, 4)
    RT_CASE(1, 1, 1, 1, 8) RT_CASE(1, 1, 1, 2, 8)
    RT_CASE(1, 2, 1, 1, 8) RT_CASE(1, 2, 1, 2, 8)
    RT_CASE(2, 1, 1, 1, 8) RT_CASE(2, 1, 1, 2, 8)
    RT_CASE(2, 2, 1, 1, 8) RT_CASE(2, 2, 1, 2, 8)
#undef RT_CASE
    return fail(RT_E_UNSUPPORTED, "conv: no kernel for window %dx%d stride %d", sc.KH, sc.KW, sc.S);
}

void free_subs(rtConvPlan* p) {
    for (auto& s : p->subs) {
        if (s.w_dev) (void)hipFree(s.w_dev);
        if (s.choff_dev) (void)hipFree(s.choff_dev);
        if (s.shift_dev) (void)hipFree(s.shift_dev);
        if (s.zs_dev) (void)hipFree(s.zs_dev);
        if (s.small_il_dev) (void)hipFree(s.small_il_dev);
    }
    p->subs.clear();
}

void free_plan(rtConvPlan* p) {
    if (!p) return;
    free_subs(p);
    if (p->ff) {
        if (p->ff->pl) free_plan(p->ff->pl);
        if (p->ff->pr) free_plan(p->ff->pr);
        if (p->ff->wedge_dev) (void)hipFree(p->ff->wedge_dev);
        for (auto& kv : p->ff->own) if (kv.second.first) (void)hipFree(kv.second.first);
        delete p->ff;
    }
    if (p->bias_dev) (void)hipFree(p->bias_dev);
    if (p->rb_w1_dev) (void)hipFree(p->rb_w1_dev);
    if (p->rb_bias1_dev) (void)hipFree(p->rb_bias1_dev);
    if (p->rbd_w1_dev) (void)hipFree(p->rbd_w1_dev);
    if (p->rbd_w2_dev) (void)hipFree(p->rbd_w2_dev);
    if (p->rbh_w1_dev) (void)hipFree(p->rbh_w1_dev);
    if (p->rbh_w2_dev) (void)hipFree(p->rbh_w2_dev);
    delete p;
}

}  // namespace

extern "C" int rt_conv2d_plan_create(rtConvPlan** out, const rtConv2dDesc* d, const void* weights, const void* bias) {
    ExactScope exact_scope(d ? d->flags : 0);
    RT_REQUIRE(out && d && weights, "rt_conv2d_plan_create: null pointer");
    RT_REQUIRE(d->Cin > 0 && d->Cout > 0 && d->Hin > 0 && d->Win > 0, "rt_conv2d_plan_create: bad dims");
    RT_REQUIRE(d->dtype == RT_F32 || d->dtype == RT_F16, "rt_conv2d_plan_create: bad dtype");
    if (!window_supported(d->KH, d->KW, d->stride))
        return fail(RT_E_UNSUPPORTED, "conv2d: window %dx%d stride %d not supported", d->KH, d->KW, d->stride);
    const int Ho = (d->Hin + 2 * d->pad_h - d->KH) / d->stride + 1;
    const int Wo = (d->Win + 2 * d->pad_w - d->KW) / d->stride + 1;
    RT_REQUIRE(Ho > 0 && Wo > 0, "conv2d: empty output");
    RT_REQUIRE((int64_t)d->Cin * d->Hin * d->Win < (1ll << 29), "conv2d: input sample exceeds 2 GB (32-bit buffer offsets)");

    auto plan = new rtConvPlan();
    plan->act = d->act; plan->has_resid = d->has_residual; plan->dtype = RT_F32; plan->w_f16 = d->dtype == RT_F16;
    plan->out_dims[0] = d->Cout; plan->out_dims[1] = Ho; plan->out_dims[2] = Wo; plan->out_dims[3] = 1;
    plan->x_bstride = (int64_t)d->Cin * d->Hin * d->Win;
    plan->y_bstride = (int64_t)d->Cout * Ho * Wo;
    plan->is2d = 1; plan->cin = d->Cin; plan->hin = d->Hin; plan->win = d->Win;
    plan->desc2d = *d; plan->is_deconv = 0; plan->flags = d->flags;

    SubConv sc;
    sc.KH = d->KH; sc.KW = d->KW; sc.S = d->stride;
    sc.Cout = d->Cout; sc.Hi = d->Hin; sc.Wi = d->Win; sc.Ho = Ho; sc.Wo = Wo;
    sc.pad_y = d->pad_h; sc.pad_x = d->pad_w; sc.nz = 1;
    sc.y_cstride = (int64_t)Ho * Wo; sc.y_ystride = Wo; sc.y_xstride = 1;
    choose_tiling(sc);
    sc.cin_real = d->Cin;
    check_direct(sc, d->Cin);
    sc.CinPad = sc.direct ? d->Cin : rt::round_up(d->Cin, sc.CC);
    if (sc.CinPad > 512) { free_plan(plan); return fail(RT_E_UNSUPPORTED, "conv2d: Cin > 512"); }
    const std::vector<float> w = to_f32(weights, (size_t)d->Cout * d->Cin * d->KH * d->KW, d->dtype);
    plan->w_canon = w;
    const int Cin = d->Cin, KH = d->KH, KW = d->KW;
    auto wfun = [&](int co, int ci, int u, int v) {
        return w[(((size_t)co * Cin + ci) * KH + u) * KW + v];          // KCRS
    };
    int rc = s3p_eligible(sc, Cin) ? upload_s3p(sc, Cin, wfun)
             : s3first_eligible(sc, Cin, d->has_residual) ? upload_s3first(sc, Cin, wfun) : upload_packed(sc, Cin, wfun);
    std::vector<int> table(sc.CinPad, -1);
    for (int c = 0; c < Cin; c++) table[c] = c * d->Hin * d->Win;
    if (!rc) rc = upload_table(sc, table);
    plan->subs.push_back(sc);
    if (!rc) {
        const std::vector<float> b = bias ? to_f32(bias, d->Cout, d->dtype) : std::vector<float>();
        rc = upload_bias(plan, bias ? b.data() : nullptr, d->Cout);
    }
    if (rc) { free_plan(plan); return rc; }
    *out = plan;
    return 0;
}

// Fused residual block  y = act2(conv3x3(act1(conv3x3(x) + b1)) + b2 + x)  (conv_s3rb_kernel).  d1 / d2 describe the two
// convolutions as they would be planned on their own (d2->has_residual = 1, the residual being the block's input);
// RT_E_UNSUPPORTED when the pair is not of that form -- the caller then plans the layers separately.
extern "C" int rt_resblock_plan_create(rtConvPlan** out, const rtConv2dDesc* d1, const void* w1, const void* b1,
                                       const rtConv2dDesc* d2, const void* w2, const void* b2) {
    ExactScope exact_scope((d1 ? d1->flags : 0) | (d2 ? d2->flags : 0));
    RT_REQUIRE(out && d1 && d2 && w1 && w2, "rt_resblock_plan_create: null pointer");
    const bool form = d1->KH == 3 && d1->KW == 3 && d2->KH == 3 && d2->KW == 3 && d1->stride == 1 && d2->stride == 1 &&
                      d1->pad_h == 1 && d1->pad_w == 1 && d2->pad_h == 1 && d2->pad_w == 1 && !d1->has_residual && d2->has_residual &&
                      d1->Cout == d2->Cin && d2->Cout == d1->Cin && d1->Hin == d2->Hin && d1->Win == d2->Win &&
                      d1->Cin <= 32 && d1->Cout <= 32 && d1->Cin > 2 && d1->Hin > 0 && d1->Win > 0;
    if (!form || exact_fp32() || env_int("RT_NO_RB", 0) != 0)
        return fail(RT_E_UNSUPPORTED, "rt_resblock_plan_create: not a 3x3 / 3x3 stride-1 residual block with <= 32 channels");
    RT_REQUIRE((d1->dtype == RT_F32 || d1->dtype == RT_F16) && d1->dtype == d2->dtype, "rt_resblock_plan_create: bad dtype");
    RT_REQUIRE((int64_t)d1->Cin * d1->Hin * d1->Win < (1ll << 29), "rt_resblock_plan_create: input sample exceeds 2 GB");
    auto plan = new rtConvPlan();
    plan->act = d2->act; plan->has_resid = 1; plan->dtype = RT_F32; plan->w_f16 = d1->dtype == RT_F16;
    plan->out_dims[0] = d2->Cout; plan->out_dims[1] = d1->Hin; plan->out_dims[2] = d1->Win; plan->out_dims[3] = 1;
    plan->x_bstride = (int64_t)d1->Cin * d1->Hin * d1->Win;
    plan->y_bstride = (int64_t)d2->Cout * d1->Hin * d1->Win;
    plan->is2d = 1; plan->cin = d1->Cin; plan->hin = d1->Hin; plan->win = d1->Win;
    plan->desc2d = *d2; plan->is_deconv = 0; plan->flags = d1->flags | d2->flags;
    plan->rb_act1 = d1->act; plan->rb_cmid = d1->Cout;
    SubConv sc;
    sc.KH = 3; sc.KW = 3; sc.S = 1; sc.Cout = d2->Cout; sc.Hi = d1->Hin; sc.Wi = d1->Win; sc.Ho = d1->Hin; sc.Wo = d1->Win;
    sc.pad_y = 1; sc.pad_x = 1; sc.nz = 1;
    sc.y_cstride = (int64_t)sc.Ho * sc.Wo; sc.y_ystride = sc.Wo; sc.y_xstride = 1;
    sc.cin_real = d1->Cin; sc.CinPad = 32; sc.CC = 16;
    const std::vector<float> wa = to_f32(w1, (size_t)d1->Cout * d1->Cin * 9, d1->dtype), wb = to_f32(w2, (size_t)d2->Cout * d2->Cin * 9, d2->dtype);
    const int c1 = d1->Cin, c2 = d2->Cin;
    // both convolutions: split weights in conv_s3_kernel's slab order (two chunks of 16 input channels each)
    sc.split3 = 1;
    SubConv tmp = sc;
    tmp.Cout = d1->Cout;
    int rc = upload_packed(tmp, c1, [&](int co, int ci, int u, int v) { return wa[(((size_t)co * c1 + ci) * 3 + u) * 3 + v]; });
    plan->rb_w1_dev = tmp.w_dev;
    if (!rc) rc = upload_packed(sc, c2, [&](int co, int ci, int u, int v) { return wb[(((size_t)co * c2 + ci) * 3 + u) * 3 + v]; });
    sc.split3 = 0; sc.rb = 1; sc.TY = 4; sc.NW = 4; sc.TXW = 1; sc.NBW = 1;
    std::vector<int> table(sc.CinPad, -1);
    for (int c = 0; c < c1; c++) table[c] = c * d1->Hin * d1->Win;
    if (!rc) rc = upload_table(sc, table);
    plan->subs.push_back(sc);
    if (!rc) {
        const std::vector<float> bb = b2 ? to_f32(b2, d2->Cout, d2->dtype) : std::vector<float>();
        rc = upload_bias(plan, b2 ? bb.data() : nullptr, d2->Cout);
    }
    if (!rc) {
        std::vector<float> ba(64, 0.f);
        if (b1) { const std::vector<float> t = to_f32(b1, d1->Cout, d1->dtype); std::copy(t.begin(), t.end(), ba.begin()); }
        if (hipMalloc((void**)&plan->rb_bias1_dev, ba.size() * 4) != hipSuccess ||
            hipMemcpy(plan->rb_bias1_dev, ba.data(), ba.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(RT_E_NOMEM, "rt_resblock_plan_create: device allocation failed");
    }
    // the tower form: the same slabs with output channel co in row rbd_row(co) of the A operands (conv_rbd.hip.h: lane (pixel, kg) of
    // an accumulator owns channels 8 kg .. + 7 and 16 + 8 kg .. + 7)
    if (!rc && d1->Cin == 32 && d1->Cout == 32 && d2->Cout == 32) {
        auto rbd_row = [](int co) { return 8 * (2 * (co >> 4) + ((co >> 2) & 1)) + 4 * ((co >> 3) & 1) + (co & 3); };
        auto pack = [&](const std::vector<float>& w, float** dev) {
            std::vector<uint16_t> slab((size_t)18 * 2 * 64 * 8, 0);
            for (int co = 0; co < 32; co++)
                for (int ci = 0; ci < 32; ci++)
                    for (int t = 0; t < 9; t++) {
                        uint16_t hi, lo;
                        split_f16(w[((size_t)co * 32 + ci) * 9 + t], hi, lo);
                        const size_t sl = (size_t)(ci / 16) * 9 + t;
                        const int kg = (ci % 16) / 8, e = ci % 8, row = rbd_row(co);
                        slab[(((sl * 2 + 0) * 2 + kg) * 32 + row) * 8 + e] = hi;
                        slab[(((sl * 2 + 1) * 2 + kg) * 32 + row) * 8 + e] = lo;
                    }
            if (hipMalloc((void**)dev, slab.size() * 2) != hipSuccess || hipMemcpy(*dev, slab.data(), slab.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
                return fail(RT_E_NOMEM, "rt_resblock_plan_create: device allocation failed");
            return 0;
        };
        rc = pack(wa, &plan->rbd_w1_dev);
        if (!rc) rc = pack(wb, &plan->rbd_w2_dev);
        // half2 form: the weights rounded to fp16 (what repack_f16mma gives the layer-by-layer kernel), [chunk * 9 + tap][k-group][row][8]
        auto pack16 = [&](const std::vector<float>& w, float** dev) {
            std::vector<uint16_t> slab((size_t)18 * 64 * 8, 0);
            for (int co = 0; co < 32; co++)
                for (int ci = 0; ci < 32; ci++)
                    for (int t = 0; t < 9; t++) {
                        const _Float16 hv = (_Float16)w[((size_t)co * 32 + ci) * 9 + t];
                        const size_t sl = (size_t)(ci / 16) * 9 + t;
                        std::memcpy(&slab[((sl * 2 + (ci % 16) / 8) * 32 + rbd_row(co)) * 8 + ci % 8], &hv, 2);
                    }
            if (hipMalloc((void**)dev, slab.size() * 2) != hipSuccess || hipMemcpy(*dev, slab.data(), slab.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
                return fail(RT_E_NOMEM, "rt_resblock_plan_create: device allocation failed");
            return 0;
        };
        if (!rc) rc = pack16(wa, &plan->rbh_w1_dev);
        if (!rc) rc = pack16(wb, &plan->rbh_w2_dev);
    }
    if (rc) { free_plan(plan); return rc; }
    *out = plan;
    return 0;
}

// Pre-split tensors between the tower blocks (conv_rbd.hip.h): (C/8, H, pitch, [8 x fp16 hi | 8 x scaled fp16 lo]) per sample, the operands
// of the split-fp16 MFMAs as stored values, x = hi + lo * 2^-11 -- the same 4 bytes per element and the same strides as the fp32 tensor.
// x_split: the block reads one (and takes its skip connection from it), y_split: it writes one.
extern "C" int rt_resblock_plan_supports_split(const rtConvPlan* plan);
extern "C" int rt_resblock_plan_set_split(rtConvPlan* plan, int x_split, int y_split) {
    RT_REQUIRE(plan, "rt_resblock_plan_set_split: null plan");
    if (!x_split && !y_split) { plan->x_split = plan->y_split = 0; return 0; }
    if (!rt_resblock_plan_supports_split(plan))
        return fail(RT_E_UNSUPPORTED, "rt_resblock_plan_set_split: only the tower block takes pre-split tensors (32 -> 32 -> 32 channels, ELU after both "
                                       "convolutions, fp32 channel-interleaved tensors: call rt_conv_plan_set_layouts(1, 1, 1) first)");
    plan->x_split = x_split != 0; plan->y_split = y_split != 0;
    return 0;
}

extern "C" int rt_resblock_plan_supports_split(const rtConvPlan* plan) {
    if (!plan || plan->subs.size() != 1 || !plan->rbd_w1_dev || env_int("RT_NO_RBD", 0) != 0) return 0;
    const SubConv& sc = plan->subs[0];
    return sc.rb && sc.x_il8 && sc.y_il8 && !sc.x_f16 && !sc.y_f16 && plan->cin == 32 && plan->rb_cmid == 32 && sc.Cout == 32 && plan->rb_act1 == 1 &&
           plan->act == 1 && !(plan->flags & RT_CONV_EXACT_FP32);
}

extern "C" int rt_deconv2d_plan_create(rtConvPlan** out, const rtConv2dDesc* d, const void* weights, const void* bias) {
    ExactScope exact_scope(d ? d->flags : 0);
    RT_REQUIRE(out && d && weights, "rt_deconv2d_plan_create: null pointer");
    RT_REQUIRE(d->Cin > 0 && d->Cout > 0 && d->Hin > 0 && d->Win > 0, "rt_deconv2d_plan_create: bad dims");
    RT_REQUIRE(d->dtype == RT_F32 || d->dtype == RT_F16, "rt_deconv2d_plan_create: bad dtype");
    if (!((d->KH == 3 && d->KW == 3) || (d->KH == 1 && d->KW == 1)) || d->stride < 1 || d->stride > 2)
        return fail(RT_E_UNSUPPORTED, "deconv2d: only 3x3 / 1x1 kernels with stride 1 or 2");
    const int s = d->stride;
    const int Ho = (d->Hin - 1) * s - 2 * d->pad_h + d->KH;
    const int Wo = (d->Win - 1) * s - 2 * d->pad_w + d->KW;
    RT_REQUIRE(Ho > 0 && Wo > 0, "deconv2d: empty output");
    RT_REQUIRE((int64_t)d->Cin * d->Hin * d->Win < (1ll << 29), "deconv2d: input sample exceeds 2 GB (32-bit buffer offsets)");

    auto plan = new rtConvPlan();
    plan->act = d->act; plan->has_resid = d->has_residual; plan->dtype = RT_F32; plan->w_f16 = d->dtype == RT_F16;
    plan->out_dims[0] = d->Cout; plan->out_dims[1] = Ho; plan->out_dims[2] = Wo; plan->out_dims[3] = 1;
    plan->x_bstride = (int64_t)d->Cin * d->Hin * d->Win;
    plan->y_bstride = (int64_t)d->Cout * Ho * Wo;
    plan->is2d = 1; plan->cin = d->Cin; plan->hin = d->Hin; plan->win = d->Win;
    plan->desc2d = *d; plan->is_deconv = 1; plan->flags = d->flags;
    const std::vector<float> w = to_f32(weights, (size_t)d->Cin * d->Cout * d->KH * d->KW, d->dtype);
    plan->w_canon = w;
    const int Cin = d->Cin, Cout = d->Cout, KH = d->KH, KW = d->KW;

    // Last layer of ResNet-18 2D (32 -> 1, 3x3, stride 2): 2x2-output-block kernel, see deconv3d_small.hip.h
    if (Cout <= 2 && KH == 3 && KW == 3 && s == 2 && env_int("RT_NO_DECONV3D_SMALL", 0) == 0) {
        const Phase1D qy[2] = {phase1d(2, d->pad_h, 3, 0), phase1d(2, d->pad_h, 3, 1)};
        const Phase1D qx[2] = {phase1d(2, d->pad_w, 3, 0), phase1d(2, d->pad_w, 3, 1)};
        auto base_of = [](const Phase1D* ph, bool& ok) {
            int lo = 1 << 20, hi = -(1 << 20);
            for (int f = 0; f < 2; f++)
                for (int u = 0; u < ph[f].K; u++) { lo = std::min(lo, u - ph[f].pad); hi = std::max(hi, u - ph[f].pad); }
            ok = ok && hi - lo <= 1;
            return lo;
        };
        bool ok = true;
        const int by = base_of(qy, ok), bx = base_of(qx, ok);
        if (ok) {
            std::vector<float> packed((size_t)Cin * Cout * 16, 0.f);
            for (int k = 0; k < Cin; k++)
                for (int co = 0; co < Cout; co++)
                    for (int fy = 0; fy < 2; fy++)
                        for (int fx = 0; fx < 2; fx++)
                            for (int uy = 0; uy < qy[fy].K; uy++)
                                for (int ux = 0; ux < qx[fx].K; ux++) {
                                    const int jy = uy - qy[fy].pad - by, jx = ux - qx[fx].pad - bx;
                                    packed[(((size_t)k * Cout + co) * 4 + 2 * fy + fx) * 4 + 2 * jy + jx] +=
                                        w[(((size_t)k * Cout + co) * KH + qy[fy].tap[uy]) * KW + qx[fx].tap[ux]];   // (Cin,Cout,R,S)
                                }
            SubConv sc;
            sc.small3d = 2; sc.Cout = Cout; sc.nz = 1;
            sc.s3.K = Cin; sc.s3.Dy = 1; sc.s3.Hy = d->Hin; sc.s3.Wy = d->Win;
            sc.s3.Dx = 1; sc.s3.Hx = Ho; sc.s3.Wx = Wo; sc.s3.C = Cout;
            sc.s3.bz = 0; sc.s3.by = by; sc.s3.bx = bx; sc.s3.Mz = 1;
            sc.s3.xp = d->Win; sc.s3.yp = Wo;
            sc.s3.sparse = (by == 0 && bx == 0 && drop_structural_zeros<false, 0>(packed)) ? 1 : 0;
            int rc = upload_weights(sc, packed);
            plan->subs.push_back(sc);
            if (!rc) {
                const std::vector<float> b = bias ? to_f32(bias, Cout, d->dtype) : std::vector<float>();
                rc = upload_bias(plan, bias ? b.data() : nullptr, Cout);
            }
            if (rc) { free_plan(plan); return rc; }
            *out = plan;
            return 0;
        }
    }
    // All s*s output phases go into ONE launch: the window is the largest phase window (2x2 for a 3x3
    // stride-2 kernel), phases with fewer taps get zero weights for the missing ones, and a ZSlice per
    // phase carries its padding, output origin and weight slab.
    std::vector<Phase1D> py_ph, px_ph;
    int wy = 1, wx = 1;
    for (int ph = 0; ph < s; ph++) {
        py_ph.push_back(phase1d(s, d->pad_h, KH, ph));
        px_ph.push_back(phase1d(s, d->pad_w, KW, ph));
        wy = std::max(wy, py_ph.back().K);
        wx = std::max(wx, px_ph.back().K);
    }
    SubConv sc;
    sc.KH = wy; sc.KW = wx; sc.S = 1; sc.Cout = Cout; sc.Hi = d->Hin; sc.Wi = d->Win;
    sc.Ho = (Ho + s - 1) / s; sc.Wo = (Wo + s - 1) / s;
    sc.y_cstride = (int64_t)Ho * Wo; sc.y_ystride = s * Wo; sc.y_xstride = s;
    sc.nz = 0;
    for (int py = 0; py < s; py++)
        for (int px = 0; px < s; px++)
            if (py < Ho && px < Wo) sc.nz++;
    choose_tiling(sc, false);      // phase launches carry ZSlice tables
    sc.cin_real = Cin;
    check_direct(sc, Cin);
    sc.CinPad = sc.direct ? Cin : rt::round_up(Cin, sc.CC);
    int rc = 0;
    if (sc.CinPad > 512 || !window_supported(sc.KH, sc.KW, 1)) rc = fail(RT_E_UNSUPPORTED, "deconv2d: unsupported shape");
    std::vector<float> packed;
    std::vector<rt::ZSlice> zs;
    for (int py = 0; py < s && !rc; py++)
        for (int px = 0; px < s && !rc; px++) {
            if (py >= Ho || px >= Wo) continue;
            const Phase1D &ay = py_ph[py], &ax = px_ph[px];
            rt::ZSlice z{};
            z.pad_y = ay.K ? ay.pad : 0; z.pad_x = ax.K ? ax.pad : 0;
            z.Ho = (Ho - py + s - 1) / s; z.Wo = (Wo - px + s - 1) / s;
            z.ch_row = 0;
            z.y_off = (int64_t)py * Wo + px;
            z.r_off = z.y_off;
            z.r_off_il8 = 8 * z.y_off;                 // 2-D phases: the offset is a pixel offset
            z.r_off_il4 = 4 * z.y_off;
            z.tap_mask = 0;
            for (int u = 0; u < ay.K; u++)
                for (int v = 0; v < ax.K; v++) z.tap_mask |= 1u << (u * sc.KW + v);
            z.w_off = pack_into(packed, sc, Cin, [&](int co, int ci, int u, int v) {
                if (u >= ay.K || v >= ax.K) return 0.f;                               // padded tap / empty phase
                return w[(((size_t)ci * Cout + co) * KH + ay.tap[u]) * KW + ax.tap[v]];   // (Cin,Cout,R,S)
            });
            zs.push_back(z);
        }
    if (!rc) rc = upload_weights(sc, packed);
    if (!rc && (s > 1)) { rc = upload_zslices(sc, zs); sc.zs_host = zs; }
    if (!rc && s == 1) { sc.pad_y = zs[0].pad_y; sc.pad_x = zs[0].pad_x; sc.Ho = zs[0].Ho; sc.Wo = zs[0].Wo; sc.y_off = zs[0].y_off; }
    std::vector<int> table(sc.CinPad, -1);
    for (int c = 0; c < Cin; c++) table[c] = c * d->Hin * d->Win;
    if (!rc) rc = upload_table(sc, table);
    plan->subs.push_back(sc);
    if (!rc) {
        const std::vector<float> b = bias ? to_f32(bias, Cout, d->dtype) : std::vector<float>();
        rc = upload_bias(plan, bias ? b.data() : nullptr, Cout);
    }
    if (rc) { free_plan(plan); return rc; }
    *out = plan;
    return 0;
}

namespace {
int check_conv3d_desc(const rtConv3dDesc* d, const char* who) {
    RT_REQUIRE(d->C > 0 && d->K > 0 && d->D > 0 && d->H > 0 && d->W > 0, "%s: bad dims", who);
    RT_REQUIRE(d->dtype == RT_F32 || d->dtype == RT_F16, "%s: bad dtype", who);
    RT_REQUIRE(d->kernel[0] >= 1 && d->kernel[0] <= 8, "%s: bad kernel depth", who);
    RT_REQUIRE(d->kernel[1] == d->kernel[2] && (d->kernel[1] == 1 || d->kernel[1] == 3), "%s: R,S must be 1x1 or 3x3", who);
    RT_REQUIRE(d->stride[1] == d->stride[2] && d->stride[1] >= 1 && d->stride[1] <= 2, "%s: H/W stride must be equal, 1 or 2", who);
    RT_REQUIRE(d->stride[0] >= 1 && d->stride[0] <= 2, "%s: D stride must be 1 or 2", who);
    // same validation as the reference plugins (lib/conv3d_plugin.cpp:43-49)
    RT_REQUIRE(d->pad_start[1] == d->pad_end[1] && d->pad_start[2] == d->pad_end[2], "%s: H/W padding must be symmetric", who);
    RT_REQUIRE(d->pad_start[0] == d->pad_end[0] || d->pad_start[0] == d->pad_end[0] - 1, "%s: unsupported D padding", who);
    return 0;
}
}  // namespace

extern "C" int rt_conv3d_plan_create(rtConvPlan** out, const rtConv3dDesc* d, const void* weights, const void* bias) {
    ExactScope exact_scope(d ? d->flags : 0);
    RT_REQUIRE(out && d && weights, "rt_conv3d_plan_create: null pointer");
    if (int rc = check_conv3d_desc(d, "rt_conv3d_plan_create")) return rc;
    const int V = d->kernel[0], R = d->kernel[1], Sk = d->kernel[2];
    const int sd = d->stride[0], sh = d->stride[1];
    const int pd = d->pad_start[0], ph = d->pad_start[1], pw = d->pad_start[2];
    // output dims exactly as cuDNN derives them from pad_start (lib/conv3d_plugin.cpp:74-100)
    const int Do = (d->D + 2 * pd - V) / sd + 1, Ho = (d->H + 2 * ph - R) / sh + 1, Wo = (d->W + 2 * pw - Sk) / sh + 1;
    RT_REQUIRE(Do > 0 && Ho > 0 && Wo > 0, "conv3d: empty output");
    RT_REQUIRE((int64_t)d->D * d->C * d->H * d->W < (1ll << 29), "conv3d: input sample exceeds 2 GB (32-bit buffer offsets)");
    if (!window_supported(R, Sk, sh)) return fail(RT_E_UNSUPPORTED, "conv3d: window %dx%d stride %d", R, Sk, sh);

    auto plan = new rtConvPlan();
    plan->flags = d->flags;
    plan->act = d->act; plan->has_resid = d->has_residual; plan->dtype = RT_F32; plan->w_f16 = d->dtype == RT_F16;
    const int K = d->K, C = d->C;
    if (d->out_dchw) { plan->out_dims[0] = Do; plan->out_dims[1] = K; }
    else { plan->out_dims[0] = K; plan->out_dims[1] = Do; }
    plan->out_dims[2] = Ho; plan->out_dims[3] = Wo;
    const int Dreal = d->D - d->in_pad_end;               // slices that exist in memory (folded Pad plugin)
    RT_REQUIRE(d->in_pad_end >= 0 && Dreal > 0, "conv3d: in_pad_end %d out of range", d->in_pad_end);
    const int F = d->cv_fold;                             // folded default cost volume: x is the (2F, H, W) tensor [left | right]
    RT_REQUIRE(F == 0 || (F > 0 && C == 2 * F && F % 4 == 0 && d->in_pad_end == 0), "conv3d: cv_fold needs C == 2 * cv_fold, a multiple of 4, no folded pad");
    plan->x_bstride = F ? (int64_t)C * d->H * d->W : (int64_t)Dreal * C * d->H * d->W;
    plan->y_bstride = (int64_t)K * Do * Ho * Wo;

    SubConv sc;
    sc.KH = R; sc.KW = Sk; sc.S = sh; sc.Cout = K;
    sc.Hi = d->H; sc.Wi = d->W; sc.Ho = Ho; sc.Wo = Wo; sc.pad_y = ph; sc.pad_x = pw; sc.nz = Do;
    sc.y_ystride = Wo; sc.y_xstride = 1;
    if (d->out_dchw) { sc.y_cstride = (int64_t)Ho * Wo; sc.y_zstride = (int64_t)K * Ho * Wo; }
    else { sc.y_cstride = (int64_t)Do * Ho * Wo; sc.y_zstride = (int64_t)Ho * Wo; }
    choose_tiling(sc);
    const int cin_real = V * C;
    sc.cin_real = cin_real;
    check_direct(sc, cin_real);
    sc.CinPad = sc.direct ? cin_real : rt::round_up(cin_real, sc.CC);
    if (sc.CinPad > 512) { free_plan(plan); return fail(RT_E_UNSUPPORTED, "conv3d: V*C = %d > 512", cin_real); }
    const std::vector<float> w = to_f32(weights, (size_t)K * V * C * R * Sk, d->dtype);
    int rc = upload_packed(sc, cin_real, [&](int co, int ci, int u, int v) {
        return w[((size_t)co * cin_real + ci) * R * Sk + u * Sk + v];      // (K, V*C, R, S): ci = v*C + c
    });
    // gather table: z-slice `dz` reads input depth dz*sd + v - pd for tap v (the (D*C)-merged axis of
    // lib/conv_utils.cpp:27-32 with its D stride/pad multiplied by C, :58-72); out-of-range depth = zeros.
    std::vector<int> table((size_t)Do * sc.CinPad, -1), shift((size_t)Do * sc.CinPad, 0);
    const int64_t plane = (int64_t)d->H * d->W;
    for (int dz = 0; dz < Do; dz++)
        for (int v = 0; v < V; v++) {
            const int din = dz * sd + v - pd;
            if (din < 0 || din >= Dreal) continue;
            for (int c = 0; c < C; c++) {
                // folded cost volume (lib/kernels.cu:72-97): every depth slice reads the same 2F planes, the right-image half
                // shifted by the slice's disparity
                table[(size_t)dz * sc.CinPad + v * C + c] = F ? (int)(c * plane) : (int)(((int64_t)din * C + c) * plane);
                if (F && c >= F) shift[(size_t)dz * sc.CinPad + v * C + c] = din;
            }
        }
    if (!rc) rc = upload_table(sc, table);
    if (!rc && F) {
        if (!sc.split3 || sc.KH != 3 || sc.S > 2) rc = fail(RT_E_UNSUPPORTED, "conv3d: cv_fold is built for the split-fp16 3x3 kernels only");
        else if (hipMalloc((void**)&sc.shift_dev, shift.size() * sizeof(int)) != hipSuccess ||
                 hipMemcpy(sc.shift_dev, shift.data(), shift.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(RT_E_NOMEM, "conv3d: device allocation failed");
    }
    plan->subs.push_back(sc);
    plan->is_conv3d = 1; plan->c3d_C = C; plan->c3d_cin = cin_real; plan->c3d_dchw = d->out_dchw != 0; plan->c3d_fold = F;
    plan->c3d_dw = V == 3 && R == 3 && Sk == 3 && sd == 1 && sh == 1 && pd == 1 && ph == 1 && pw == 1 && d->in_pad_end == 0 && F == 0 && C % 16 == 0 && Do == d->D;
    plan->w_canon = w;                                         // (K, V*C, R, S): re-packed when the plan switches to fp16 operands
    if (!rc) {
        const std::vector<float> b = bias ? to_f32(bias, K, d->dtype) : std::vector<float>();
        rc = upload_bias(plan, bias ? b.data() : nullptr, K);
    }
    // The folded cost volume, factored (fold_factor.hip.h): 3x3x3, stride 1, pad 1, depth-major output, no skip tensor -- the first Conv3D
    // of every 3-D model of the reference.  The gather form above stays as the fallback for tensors this form does not take.
    if (!rc && F && V == 3 && R == 3 && Sk == 3 && sd == 1 && sh == 1 && pd == 1 && ph == 1 && pw == 1 && Do == d->D && Ho == d->H && Wo == d->W &&
        d->D >= 3 && d->out_dchw && !d->has_residual && K % 8 == 0 && !(d->flags & RT_CONV_EXACT_FP32) && sc.split3 && env_int("RT_NO_FOLD_FACTOR", 0) == 0) {
        auto ff = new FoldFactor();
        plan->ff = ff;
        ff->F = F; ff->K = K; ff->D = d->D; ff->H = d->H; ff->W = d->W;
        ff->a_elems = (int64_t)3 * K * d->H * d->W; ff->c_elems = (int64_t)3 * K * d->H * (d->W + 2);
        ff->t_elems = (int64_t)3 * K * d->H * (d->W + 2);
        ff->e_elems = (int64_t)d->D * K * d->H;
        std::vector<float> wl((size_t)3 * K * F * 9, 0.f), wr((size_t)3 * K * F * 9, 0.f), wedge((size_t)9 * F * K, 0.f);
        for (int k = 0; k < K; k++)
            for (int c = 0; c < F; c++)
                for (int t = 0; t < 9; t++) {
                    float wj[3];
                    for (int j = 0; j < 3; j++) {
                        wj[j] = w[((size_t)k * cin_real + j * C + c) * 9 + t];                               // left half, depth tap j
                        const float wrj = w[((size_t)k * cin_real + j * C + F + c) * 9 + t];                 // right half
                        wr[(((size_t)j * K + k) * F + c) * 9 + t] = wrj;
                        if (t % 3 == 2) wedge[(((size_t)j * 3 + t / 3) * F + c) * K + k] = wrj;
                    }
                    wl[(((size_t)0 * K + k) * F + c) * 9 + t] = wj[1] + wj[2];                                   // first slice: depth taps 1, 2
                    wl[(((size_t)1 * K + k) * F + c) * 9 + t] = (wj[0] + wj[1]) + wj[2];
                    wl[(((size_t)2 * K + k) * F + c) * 9 + t] = wj[0] + wj[1];                                   // last slice
                }
        rtConv2dDesc d2{};
        d2.Cin = F; d2.Cout = 3 * K; d2.Hin = d->H; d2.Win = d->W; d2.KH = d2.KW = 3; d2.stride = 1; d2.pad_h = d2.pad_w = 1;
        d2.act = RT_ACT_NONE; d2.has_residual = 0; d2.dtype = RT_F32; d2.flags = d->flags;
        rc = rt_conv2d_plan_create(&ff->pl, &d2, wl.data(), nullptr);
        if (!rc) rc = rt_conv_plan_set_batch_strides(ff->pl, (int64_t)C * d->H * d->W, 0, 0);                    // L: the first F planes of a (2F, H, W) sample
        // C' = conv3x3([0 | R]) = conv3x3(R) with a left pad of 2: the plan reads the right feature map where it lies (round 4 copied it
        // behind a zero column first); its output is W + 2 wide, column W + 1 is never read
        d2.pad_w = 2;
        if (!rc) rc = rt_conv2d_plan_create(&ff->pr, &d2, wr.data(), nullptr);
        if (!rc) rc = rt_conv_plan_set_batch_strides(ff->pr, (int64_t)C * d->H * d->W, 0, 0);                    // R: planes F .. 2F-1 of a (2F, H, W) sample
        if (!rc && (hipMalloc((void**)&ff->wedge_dev, wedge.size() * 4) != hipSuccess ||
                    hipMemcpy(ff->wedge_dev, wedge.data(), wedge.size() * 4, hipMemcpyHostToDevice) != hipSuccess))
            rc = fail(RT_E_NOMEM, "conv3d: device allocation failed");
    }
    if (rc) { free_plan(plan); return rc; }
    *out = plan;
    return 0;
}

namespace {
// Is the factored form the one this plan launches?  (planar fp32 feature maps; any of the four output forms)
bool fold_factor_active(const rtConvPlan* plan) {
    if (!plan->ff || plan->subs.size() != 1) return false;
    const SubConv& sc = plan->subs[0];
    return !sc.x_f16 && !sc.x_il8 && !sc.f16mma && !plan->has_resid;
}

size_t fold_factor_bytes(const rtConvPlan* plan, int batch) {
    const FoldFactor* ff = plan->ff;
    return (size_t)(ff->a_elems + ff->c_elems + ff->t_elems + ff->e_elems) * 4 * (size_t)batch;
}

int enqueue_fold_factor(const rtConvPlan* plan, const void* x, void* y, int batch, void* ws, size_t ws_bytes, rtStream s, int hints) {
    FoldFactor* ff = plan->ff;
    const SubConv& sc = plan->subs[0];
    const size_t need = fold_factor_bytes(plan, batch);
    float* buf = static_cast<float*>(ws);
    if (!buf) {
        // no caller workspace: a block owned by the plan, ONE PER STREAM (launches of one stream are ordered, so they may share scratch;
        // two streams never do).  A block only grows, and only after the work its stream still has in flight on the old one is done;
        // that wait and the allocation cannot be captured into a graph: a capturing caller brings its own workspace (rt_conv_enqueue_ws).
        std::lock_guard<std::mutex> lock(ff->mu);
        std::pair<void*, size_t>& blk = ff->own[s];
        if (blk.second < need) {
#ifndef HIPEMU
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(S(s), &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
                return fail(RT_E_UNSUPPORTED, "conv3d (factored cost volume): the plan's own scratch cannot grow inside a stream capture -- pass a workspace (rt_conv_enqueue_ws)");
#endif
            if (blk.first) { (void)hipStreamSynchronize(S(s)); (void)hipFree(blk.first); blk = {nullptr, 0}; }
            if (hipMalloc(&blk.first, need) != hipSuccess) { blk = {nullptr, 0}; return fail(RT_E_NOMEM, "conv3d (factored cost volume): %zu bytes of scratch", need); }
            blk.second = need;
        }
        buf = static_cast<float*>(blk.first);
    } else {
        RT_REQUIRE(ws_bytes >= need, "rt_conv_enqueue_ws: workspace of %zu bytes, the plan needs %zu for batch %d", ws_bytes, need, batch);
        RT_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "rt_conv_enqueue_ws: workspace must be 16-byte aligned");
    }
    rt::FoldFactorArgs a;
    a.x = static_cast<const float*>(x);
    a.a = buf; a.c = buf + (int64_t)batch * ff->a_elems;
    a.t = buf + (int64_t)batch * (ff->a_elems + ff->c_elems);
    a.e = buf + (int64_t)batch * (ff->a_elems + ff->c_elems + ff->t_elems);
    a.wedge = ff->wedge_dev; a.bias = plan->bias_dev; a.y = y;
    a.F = ff->F; a.K = ff->K; a.D = ff->D; a.H = ff->H; a.W = ff->W; a.act = plan->act; a.batch = batch;
    a.x_bstride = plan->x_bstride; a.a_bstride = ff->a_elems; a.c_bstride = ff->c_elems; a.t_bstride = ff->t_elems; a.e_bstride = ff->e_elems;
    a.y_bstride = plan->y_bstride;
    const int G = sc.y_f16 ? 8 : 4;
    RT_REQUIRE((int64_t)batch * ff->K <= 65535 && ff->H <= 65535, "rt_conv_enqueue: grid limit exceeded");
    if (int rc = rt_conv_enqueue_hint(ff->pl, x, const_cast<float*>(a.a), nullptr, batch, s, hints)) return rc;
    if (int rc = rt_conv_enqueue_hint(ff->pr, a.x + (int64_t)ff->F * ff->H * ff->W, const_cast<float*>(a.c), nullptr, batch, s, hints)) return rc;
    const dim3 tgrid((unsigned)rt::cdiv(ff->W + 2, 256), (unsigned)ff->H, (unsigned)(batch * (ff->K / G)));
    if (G == 8) hipLaunchKernelGGL(rt::fold_t_kernel<8>, tgrid, dim3(256), 0, S(s), a);
    else hipLaunchKernelGGL(rt::fold_t_kernel<4>, tgrid, dim3(256), 0, S(s), a);
    RT_LAUNCH_CHECK("fold_t_kernel");
    hipLaunchKernelGGL(rt::fold_edge_kernel, dim3((unsigned)rt::cdiv(ff->H * (ff->K / 4), 64), (unsigned)ff->D, (unsigned)batch), dim3(256), 0, S(s), a);
    RT_LAUNCH_CHECK("fold_edge_kernel");
    // the combining pass over x < W - 1, then the last column (edge term) -- see fold_combine_kernel
    const dim3 grid((unsigned)rt::cdiv(ff->W - 1, 256), (unsigned)ff->H, (unsigned)(batch * (ff->K / G)));
    const dim3 lgrid((unsigned)rt::cdiv((int64_t)ff->H * ff->D * (ff->K / G) * batch, 256), 1u, 1u);
    const int U = plan->opt_fold_u;
#define RT_FOLD_COMBINE(T, il, u)                                                                          \
    do {                                                                                                   \
        if (ff->W > 1) hipLaunchKernelGGL((rt::fold_combine_kernel<T, il, u, false>), grid, dim3(256), 0, S(s), a); \
        hipLaunchKernelGGL((rt::fold_combine_kernel<T, il, 1, true>), lgrid, dim3(256), 0, S(s), a);       \
    } while (0)
#define RT_FOLD_COMBINE_U(T, il)                                                                           \
    do {                                                                                                   \
        if (U == 4) RT_FOLD_COMBINE(T, il, 4); else if (U == 2) RT_FOLD_COMBINE(T, il, 2); else RT_FOLD_COMBINE(T, il, 1); \
    } while (0)
    if (sc.y_f16) { if (sc.y_il8) RT_FOLD_COMBINE_U(_Float16, true); else RT_FOLD_COMBINE_U(_Float16, false); }
    else { if (sc.y_il8) RT_FOLD_COMBINE_U(float, true); else RT_FOLD_COMBINE_U(float, false); }
#undef RT_FOLD_COMBINE_U
#undef RT_FOLD_COMBINE
    RT_LAUNCH_CHECK("fold_combine_kernel");
    return 0;
}
}  // namespace

namespace {
// The launches of a transposed 3-D convolution (not the small-output last layer): one launch per output-depth class (depths cls, cls+sd,
// ... share the same set of depth taps); inside it every (depth position, y/x output phase) is a ZSlice -- see rt_deconv2d_plan_create.
// f16mma = false: the split-fp16 kernel (fp32 tensors, or planar fp16 ones in half2 mode).  f16mma = true: fp16 operands on
// conv_f16mma_kernel with a CHANNEL-INTERLEAVED input (K/8, Dy, Hy, Wy, 8) fp16 -- one 16-byte load per pixel and channel group instead of
// eight 2-byte ones, one MFMA per tap instead of two -- which then also writes interleaved outputs, (D, C/8, H, W, 8) or, with the fused
// Transform, (C/8, D, H, W, 8) (ZSlice::y_off_il8).  Rebuilds plan->subs from plan->desc3d / in_dims3 / w_canon; storage types and layouts
// of the old launches are carried over by the caller.
// dp4 (with f16mma): the four-phases-per-workgroup kernel (deconv_f16p.hip.h) -- 3x3 window, stride 2, pad 1 in H and W, interleaved output.
bool deconv3d_dp4_ok(const rtConvPlan* plan) {
    const rtConv3dDesc& d = plan->desc3d;
    return d.kernel[1] == 3 && d.kernel[2] == 3 && d.stride[1] == 2 && d.pad_start[1] == 1 && d.pad_start[2] == 1 && d.H == 2 * plan->in_dims3[1] - 1 &&
           d.W == 2 * plan->in_dims3[2] - 1 && d.H >= 2 && d.W >= 2 && d.C % 8 == 0 && d.K % 8 == 0 && env_int("RT_NO_DECONV_P4", 0) == 0;
}

// fp32 tensors: the four-phase kernel in split-fp16 form (deconv_s3p.hip.h) -- planar input and output, the same geometry
bool deconv3d_p4f32_ok(const rtConvPlan* plan) {
    const rtConv3dDesc& d = plan->desc3d;
    return d.kernel[1] == 3 && d.kernel[2] == 3 && d.stride[1] == 2 && d.stride[2] == 2 && d.pad_start[1] == 1 && d.pad_start[2] == 1 &&
           d.H == 2 * plan->in_dims3[1] - 1 && d.W == 2 * plan->in_dims3[2] - 1 && d.H >= 2 && d.W >= 2 && d.C > 2 && (plan->flags & RT_CONV_EXACT_FP32) == 0 &&
           !plan->w_f16 && env_int("RT_NO_DECONV_P4F32", 0) == 0;
}

int build_deconv3d_subs(rtConvPlan* plan, bool f16mma, bool dp4 = false) {
    const rtConv3dDesc* d = &plan->desc3d;
    const std::vector<float>& w = plan->w_canon;
    const int Dlim = d->out_depth > 0 ? d->out_depth : d->D;
    const bool cdhw = d->out_dchw != 0;
    const int V = d->kernel[0], R = d->kernel[1], Sk = d->kernel[2];
    const int sd = d->stride[0], sh = d->stride[1];
    const int pd = d->pad_start[0], ph_ = d->pad_start[1], pw = d->pad_start[2];
    const int Dy = plan->in_dims3[0], Hy = plan->in_dims3[1], Wy = plan->in_dims3[2];
    const int Hx = d->H, Wx = d->W, K = d->K, C = d->C;
    const int64_t in_plane = (int64_t)Hy * Wy, out_plane = (int64_t)Hx * Wx;
    free_subs(plan);
    int rc = 0;
    std::vector<Phase1D> py_ph, px_ph;
    int wy = 1, wx = 1;
    for (int ph = 0; ph < sh; ph++) {
        py_ph.push_back(phase1d(sh, ph_, R, ph));
        px_ph.push_back(phase1d(sh, pw, Sk, ph));
        wy = std::max(wy, py_ph.back().K);
        wx = std::max(wx, px_ph.back().K);
    }
    for (int cls = 0; cls < sd && !rc; cls++) {          // output depths dx = cls, cls + sd, ...
        if (cls >= Dlim) continue;
        const Phase1D az = phase1d(sd, pd, V, cls);
        const int nzd = (Dlim - cls + sd - 1) / sd;
        const int nv = std::max(az.K, 1);
        SubConv sc;
        sc.KH = wy; sc.KW = wx; sc.S = 1; sc.Cout = C; sc.Hi = Hy; sc.Wi = Wy;
        sc.Ho = (Hx + sh - 1) / sh; sc.Wo = (Wx + sh - 1) / sh;
        sc.y_cstride = cdhw ? (int64_t)Dlim * out_plane : out_plane; sc.y_ystride = sh * Wx; sc.y_xstride = sh;
        sc.r_cstride = out_plane;
        int nph = 0;
        for (int py = 0; py < sh; py++)
            for (int px = 0; px < sh; px++)
                if (py < Hx && px < Wx) nph++;
        sc.nz = nzd * nph;
        choose_tiling(sc, false);      // phase launches carry ZSlice tables
        const int cin_real = nv * K;
        sc.cin_real = cin_real;
        check_direct(sc, cin_real);
        if (f16mma) {
            if (!sc.split3 || sc.direct) { rc = fail(RT_E_UNSUPPORTED, "conv3d_transpose: no fp16-operand form for this window"); break; }
            sc.split3 = 0; sc.f16mma = 1; sc.CC = 16; sc.NBW = 1; sc.TXW = 1; sc.TY = sc.NW = 4;
        }
        if (!f16mma && dp4 && sc.split3 && !sc.direct) {      // (not the split kernel -- exact fp32 arithmetic: the phase form below)
            // fp32 tensors, four phases per workgroup (deconv_s3p.hip.h): split slabs in kernel order, planar gather table, one ZSlice per output depth
            sc.dp4 = 1; sc.KH = sc.KW = 3; sc.nz = nzd; sc.CC = 16; sc.NBW = 1; sc.TXW = 1; sc.TY = sc.NW = 4;
            sc.Ho = Hx; sc.Wo = Wx;
            sc.CinPad = rt::round_up(cin_real, sc.CC);
            if (sc.CinPad > 512) { rc = fail(RT_E_UNSUPPORTED, "conv3d_transpose: unsupported shape"); break; }
            std::vector<float> packed;
            pack_into(packed, sc, cin_real, [&](int co, int ci, int u, int v) {
                if (az.K == 0) return 0.f;
                const int j = ci / K, k = ci % K;
                return w[((((size_t)k * V + az.tap[j]) * C + co) * R + u) * Sk + v];            // KVCRS, kernel taps as they are
            });
            std::vector<rt::ZSlice> zs;
            for (int m = 0; m < nzd; m++) {
                rt::ZSlice z{};
                const int64_t dx = cls + m * sd;
                z.ch_row = m;
                z.r_off = dx * C * out_plane; z.r_off_il8 = z.r_off; z.r_off_il4 = z.r_off;
                z.y_off = cdhw ? dx * out_plane : z.r_off;
                z.y_off_il8 = cdhw ? 4 * dx * out_plane : z.r_off;      // (C/4, D, H, W, 4) / (D, C/4, H, W, 4): deconv_s3p_kernel<true>
                zs.push_back(z);
            }
            std::vector<int> table((size_t)nzd * sc.CinPad, -1);
            for (int m = 0; m < nzd; m++)
                for (int j = 0; j < az.K; j++) {
                    const int dy = m + j - az.pad;
                    if (dy < 0 || dy >= Dy) continue;
                    for (int k = 0; k < K; k++) table[(size_t)m * sc.CinPad + j * K + k] = (int)(((int64_t)k * Dy + dy) * in_plane);
                }
            if (hipMalloc((void**)&sc.w_dev, packed.size() * 4) != hipSuccess || hipMemcpy(sc.w_dev, packed.data(), packed.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(RT_E_NOMEM, "conv3d_transpose: device allocation failed");
            if (!rc) rc = upload_zslices(sc, zs);
            if (!rc) rc = upload_table(sc, table);
            plan->subs.push_back(sc);
            continue;
        }
        if (f16mma && dp4) {
            // one workgroup = a 4 x 32 tile of the input grid, all four phases: weights in kernel order (3 x 3 taps), one ZSlice per output depth
            sc.dp4 = 1; sc.KH = sc.KW = 3; sc.nz = nzd;
            sc.Ho = Hx; sc.Wo = Wx;                               // the FULL output plane
            sc.CinPad = rt::round_up(cin_real, sc.CC);
            if (sc.CinPad > 512) { rc = fail(RT_E_UNSUPPORTED, "conv3d_transpose: unsupported shape"); break; }
            std::vector<uint16_t> p16;
            pack_f16_into(p16, sc, cin_real, [&](int co, int ci, int u, int v) {
                if (az.K == 0) return 0.f;
                const int j = ci / K, k = ci % K;
                return w[((((size_t)k * V + az.tap[j]) * C + co) * R + u) * Sk + v];            // KVCRS, kernel taps as they are
            });
            std::vector<rt::ZSlice> zs;
            for (int m = 0; m < nzd; m++) {
                rt::ZSlice z{};
                const int64_t dx = cls + m * sd;
                z.ch_row = m;
                z.r_off = dx * C * out_plane; z.r_off_il8 = z.r_off; z.r_off_il4 = z.r_off;
                z.y_off = cdhw ? dx * out_plane : z.r_off;
                z.y_off_il8 = cdhw ? 8 * dx * out_plane : z.r_off;
                zs.push_back(z);
            }
            std::vector<int> table((size_t)nzd * sc.CinPad, -1);
            for (int m = 0; m < nzd; m++)
                for (int j = 0; j < az.K; j++) {
                    const int dy = m + j - az.pad;
                    if (dy < 0 || dy >= Dy) continue;
                    for (int k = 0; k < K; k++) table[(size_t)m * sc.CinPad + j * K + k] = (int)((((int64_t)(k / 8) * Dy + dy) * in_plane) * 8 + k % 8);
                }
            if (hipMalloc((void**)&sc.w_dev, p16.size() * 2) != hipSuccess || hipMemcpy(sc.w_dev, p16.data(), p16.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(RT_E_NOMEM, "conv3d_transpose: device allocation failed");
            if (!rc) rc = upload_zslices(sc, zs);
            if (!rc) rc = upload_table(sc, table);
            plan->subs.push_back(sc);
            continue;
        }
        sc.CinPad = sc.direct ? cin_real : rt::round_up(cin_real, sc.CC);
        if (sc.CinPad > 512 || !window_supported(sc.KH, sc.KW, 1)) { rc = fail(RT_E_UNSUPPORTED, "conv3d_transpose: unsupported shape"); break; }
        std::vector<float> packed;
        std::vector<uint16_t> packed16;
        std::vector<int64_t> w_offs;
        for (int py = 0; py < sh; py++)
            for (int px = 0; px < sh; px++) {
                if (py >= Hx || px >= Wx) continue;
                const Phase1D &ay = py_ph[py], &ax = px_ph[px];
                auto wfun = [&](int co, int ci, int u, int v) {
                    if (az.K == 0 || u >= ay.K || v >= ax.K) return 0.f;
                    const int j = ci / K, k = ci % K;                        // gathered channel = (depth tap j, input channel k)
                    return w[((((size_t)k * V + az.tap[j]) * C + co) * R + ay.tap[u]) * Sk + ax.tap[v]];   // KVCRS
                };
                w_offs.push_back(f16mma ? pack_f16_into(packed16, sc, cin_real, wfun) : pack_into(packed, sc, cin_real, wfun));
            }
        std::vector<rt::ZSlice> zs;
        for (int m = 0; m < nzd; m++) {
            int iph = 0;
            for (int py = 0; py < sh; py++)
                for (int px = 0; px < sh; px++) {
                    if (py >= Hx || px >= Wx) continue;
                    const Phase1D &ay = py_ph[py], &ax = px_ph[px];
                    rt::ZSlice z{};
                    z.pad_y = ay.K ? ay.pad : 0; z.pad_x = ax.K ? ax.pad : 0;
                    z.Ho = (Hx - py + sh - 1) / sh; z.Wo = (Wx - px + sh - 1) / sh;
                    z.ch_row = m;
                    z.tap_mask = 0;
                    for (int u = 0; u < ay.K; u++)
                        for (int v = 0; v < ax.K; v++) z.tap_mask |= 1u << (u * sc.KW + v);
                    const int64_t pix = (int64_t)py * Wx + px, dx = cls + m * sd;
                    z.r_off = dx * C * out_plane + pix;
                    z.r_off_il8 = dx * C * out_plane + 8 * pix;                                 // (D, C/8, H, W, 8) fp16 skip tensor
                    z.r_off_il4 = dx * C * out_plane + 4 * pix;                                 // (D, C/4, H, W, 4) fp32 skip tensor
                    z.y_off = cdhw ? dx * out_plane + pix : z.r_off;
                    z.y_off_il8 = cdhw ? 8 * (dx * out_plane + pix) : z.r_off_il8;              // (C/8, D, H, W, 8) / (D, C/8, H, W, 8)
                    z.w_off = w_offs[iph++];
                    zs.push_back(z);
                }
        }
        // table row m <-> output depth dx = cls + m*sd, which reads input depth m + j - az.pad for depth tap j.  Planar input (K, Dy, Hy, Wy):
        // plane offset of channel k; interleaved input (K/8, Dy, Hy, Wy, 8): element offset of the slot plane of channel group k / 8
        std::vector<int> table((size_t)nzd * sc.CinPad, -1);
        for (int m = 0; m < nzd; m++)
            for (int j = 0; j < az.K; j++) {
                const int dy = m + j - az.pad;
                if (dy < 0 || dy >= Dy) continue;
                for (int k = 0; k < K; k++)
                    table[(size_t)m * sc.CinPad + j * K + k] = f16mma ? (int)((((int64_t)(k / 8) * Dy + dy) * in_plane) * 8 + k % 8)
                                                                      : (int)(((int64_t)k * Dy + dy) * in_plane);
            }
        if (f16mma) {
            if (hipMalloc((void**)&sc.w_dev, packed16.size() * 2) != hipSuccess ||
                hipMemcpy(sc.w_dev, packed16.data(), packed16.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(RT_E_NOMEM, "conv3d_transpose: device allocation failed");
        } else {
            rc = upload_weights(sc, packed);
        }
        if (!rc) rc = upload_zslices(sc, zs);
        if (!rc) rc = upload_table(sc, table);
        plan->subs.push_back(sc);
    }
    return rc;
}
}  // namespace

extern "C" int rt_conv3d_transpose_plan_create(rtConvPlan** out, const rtConv3dDesc* d, const int in_dims[3],
                                               const void* weights, const void* bias) {
    ExactScope exact_scope(d ? d->flags : 0);
    RT_REQUIRE(out && d && in_dims && weights, "rt_conv3d_transpose_plan_create: null pointer");
    if (int rc = check_conv3d_desc(d, "rt_conv3d_transpose_plan_create")) return rc;
    // fused epilogue layouts (executor only, see rt_stereo.h): keep the first out_depth output slices (Slice plugin)
    // and/or write (C,D,H,W) instead of (D,C,H,W) (Transform plugin); the residual always is (D,C,H,W)
    const int Dlim = d->out_depth > 0 ? d->out_depth : d->D;
    RT_REQUIRE(Dlim <= d->D, "conv3d_transpose: out_depth %d exceeds the output depth %d", Dlim, d->D);
    const bool cdhw = d->out_dchw != 0;
    const int V = d->kernel[0], R = d->kernel[1], Sk = d->kernel[2];
    const int sd = d->stride[0], sh = d->stride[1];
    const int pd = d->pad_start[0], ph_ = d->pad_start[1], pw = d->pad_start[2];
    const int Dy = in_dims[0], Hy = in_dims[1], Wy = in_dims[2];
    const int Dx = d->D, Hx = d->H, Wx = d->W, K = d->K, C = d->C;
    // the reference verifies that the forward conv of the output gives the input dims
    // (lib/conv3d_transpose_plugin.cpp:108-111)
    RT_REQUIRE((Dx + 2 * pd - V) / sd + 1 == Dy && (Hx + 2 * ph_ - R) / sh + 1 == Hy && (Wx + 2 * pw - Sk) / sh + 1 == Wy,
               "conv3d_transpose: out dims (%d,%d,%d) inconsistent with in dims (%d,%d,%d)", Dx, Hx, Wx, Dy, Hy, Wy);
    RT_REQUIRE((int64_t)K * Dy * Hy * Wy < (1ll << 29), "conv3d_transpose: input sample exceeds 2 GB (32-bit buffer offsets)");

    auto plan = new rtConvPlan();
    plan->flags = d->flags;
    plan->act = d->act; plan->has_resid = d->has_residual; plan->dtype = RT_F32; plan->w_f16 = d->dtype == RT_F16;
    plan->out_dims[0] = cdhw ? C : Dlim; plan->out_dims[1] = cdhw ? Dlim : C; plan->out_dims[2] = Hx; plan->out_dims[3] = Wx;
    plan->x_bstride = (int64_t)K * Dy * Hy * Wy;
    plan->y_bstride = (int64_t)Dlim * C * Hx * Wx;
    const std::vector<float> w = to_f32(weights, (size_t)K * V * C * R * Sk, d->dtype);
    int rc = 0;
    // Last layer of the 3-D models: one or two output channels, 3x3x3, stride 2 -> 2x2x2-block kernel
    if (C <= 2 && !cdhw && V == 3 && R == 3 && Sk == 3 && sd == 2 && sh == 2 && env_int("RT_NO_DECONV3D_SMALL", 0) == 0) {
        const Phase1D pz[2] = {phase1d(2, pd, 3, 0), phase1d(2, pd, 3, 1)};
        const Phase1D py[2] = {phase1d(2, ph_, 3, 0), phase1d(2, ph_, 3, 1)};
        const Phase1D px[2] = {phase1d(2, pw, 3, 0), phase1d(2, pw, 3, 1)};
        auto base_of = [](const Phase1D* ph, bool& ok) {
            int lo = 1 << 20, hi = -(1 << 20);
            for (int f = 0; f < 2; f++)
                for (int u = 0; u < ph[f].K; u++) { lo = std::min(lo, u - ph[f].pad); hi = std::max(hi, u - ph[f].pad); }
            ok = ok && hi - lo <= 1;
            return lo;
        };
        bool ok = true;
        const int bz = base_of(pz, ok), by = base_of(py, ok), bx = base_of(px, ok);
        if (ok) {
            const int CO = C;      // kernel template = C
            std::vector<float> packed((size_t)K * CO * 64, 0.f);
            for (int k = 0; k < K; k++)
                for (int co = 0; co < CO; co++)
                    for (int fz = 0; fz < 2; fz++)
                        for (int fy = 0; fy < 2; fy++)
                            for (int fx = 0; fx < 2; fx++)
                                for (int uz = 0; uz < pz[fz].K; uz++)
                                    for (int uy = 0; uy < py[fy].K; uy++)
                                        for (int ux = 0; ux < px[fx].K; ux++) {
                                            const int jz = uz - pz[fz].pad - bz, jy = uy - py[fy].pad - by, jx = ux - px[fx].pad - bx;
                                            const int f = 4 * fz + 2 * fy + fx, j = 4 * jz + 2 * jy + jx;
                                            packed[(((size_t)k * CO + co) * 8 + f) * 8 + j] +=
                                                w[((((size_t)k * V + pz[fz].tap[uz]) * C + co) * R + py[fy].tap[uy]) * Sk + px[fx].tap[ux]];
                                        }
            SubConv sc;
            sc.small3d = 1; sc.Cout = C; sc.nz = 1;
            sc.s3.K = K; sc.s3.Dy = Dy; sc.s3.Hy = Hy; sc.s3.Wy = Wy;
            sc.s3.Dx = Dlim; sc.s3.Hx = Hx; sc.s3.Wx = Wx; sc.s3.C = C;
            sc.s3.bz = bz; sc.s3.by = by; sc.s3.bx = bx; sc.s3.Mz = (Dlim + 1) / 2;
            sc.s3.xp = Wy; sc.s3.yp = Wx;
            sc.small_w = packed;
            plan->is_deconv3d = 1; plan->c3d_C = C;
            sc.s3.sparse = (by != 0 || bx != 0) ? 0 : (bz == 0 && drop_structural_zeros<true, 0>(packed)) ? 1 :
                           (bz == -1 && drop_structural_zeros<true, 4>(packed)) ? 1 + 4 : 0;
            RT_REQUIRE((int64_t)Dx * C * Hx * Wx < (1ll << 29), "conv3d_transpose: output sample exceeds 2 GB (32-bit buffer offsets)");
            rc = upload_weights(sc, packed);
            plan->subs.push_back(sc);
            if (!rc) {
                const std::vector<float> b = bias ? to_f32(bias, C, d->dtype) : std::vector<float>();
                rc = upload_bias(plan, bias ? b.data() : nullptr, C);
            }
            if (rc) { free_plan(plan); return rc; }
            *out = plan;
            return 0;
        }
    }
    plan->desc3d = *d;
    for (int i = 0; i < 3; i++) plan->in_dims3[i] = in_dims[i];
    plan->w_canon = w;                                         // KVCRS: the launches are rebuilt from it when the plan changes kernels
    plan->is_deconv3d = 1; plan->c3d_C = C;
    rc = build_deconv3d_subs(plan, false, deconv3d_p4f32_ok(plan));
    if (!rc) {
        const std::vector<float> b = bias ? to_f32(bias, C, d->dtype) : std::vector<float>();
        rc = upload_bias(plan, bias ? b.data() : nullptr, C);
    }
    if (rc) { free_plan(plan); return rc; }
    *out = plan;
    return 0;
}

// Row pitch (elements) of the input and output planes of a 2-D plan; 0 = dense.  Lets an executor keep its
// internal activations 128-byte aligned per row (W is odd in every Stereo DNN network): +6 % on the 3x3 layers.
extern "C" int rt_conv_plan_set_pitch(rtConvPlan* plan, int in_pitch, int out_pitch) {
    RT_REQUIRE(plan, "rt_conv_plan_set_pitch: null plan");
    RT_REQUIRE(plan->is2d, "rt_conv_plan_set_pitch: only 2-D convolution plans can be re-pitched");
    const int Hout = plan->out_dims[1], Wout = plan->out_dims[2];
    const int ip = in_pitch ? in_pitch : plan->win, op = out_pitch ? out_pitch : Wout;
    RT_REQUIRE(ip >= plan->win && op >= Wout, "rt_conv_plan_set_pitch: pitch smaller than the row");
    RT_REQUIRE((int64_t)plan->cin * plan->hin * ip < (1ll << 29), "rt_conv_plan_set_pitch: input sample exceeds 2 GB");
    for (const SubConv& sc : plan->subs)
        RT_REQUIRE(!sc.f16mma || ip % 2 == 0, "rt_conv_plan_set_pitch: the fp16-arithmetic kernel needs an even input pitch");
    const int old_op = plan->out_pitch ? plan->out_pitch : Wout;
    auto repitch = [&](int64_t off) { return (off / old_op) * op + off % old_op; };
    for (SubConv& sc : plan->subs) {
        if (sc.small3d) { sc.s3.xp = ip; sc.s3.yp = op; continue; }
        sc.x_pitch = ip;
        std::vector<int> table(sc.CinPad, -1);
        for (int c = 0; c < plan->cin; c++) table[c] = c * plan->hin * ip;
        RT_HIP(hipMemcpy(sc.choff_dev, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice));
        sc.y_ystride = (sc.y_ystride / old_op) * op;
        sc.y_cstride = (int64_t)Hout * op;
        sc.y_off = repitch(sc.y_off);
        if (!sc.zs_host.empty()) {
            for (auto& z : sc.zs_host) { z.y_off = repitch(z.y_off); z.r_off = z.y_off; z.r_off_il8 = 8 * z.y_off; z.r_off_il4 = 4 * z.y_off; }
            RT_HIP(hipMemcpy(sc.zs_dev, sc.zs_host.data(), sc.zs_host.size() * sizeof(rt::ZSlice), hipMemcpyHostToDevice));
        }
    }
    plan->x_bstride = (int64_t)plan->cin * plan->hin * ip;
    plan->y_bstride = (int64_t)plan->out_dims[0] * Hout * op;
    plan->in_pitch = in_pitch ? ip : 0;
    plan->out_pitch = op == Wout ? 0 : op;
    return 0;
}

// Per-sample strides (elements; 0 = leave as it is) of the input, output and residual tensor of a plan: lets an executor
// place a tensor inside a larger buffer (a channel range of a concatenated tensor).  Call after rt_conv_plan_set_pitch,
// which recomputes the dense strides.
extern "C" int rt_conv_plan_set_batch_strides(rtConvPlan* plan, int64_t x_bstride, int64_t y_bstride, int64_t r_bstride) {
    RT_REQUIRE(plan, "rt_conv_plan_set_batch_strides: null plan");
    RT_REQUIRE(x_bstride >= 0 && y_bstride >= 0 && r_bstride >= 0, "rt_conv_plan_set_batch_strides: negative stride");
    RT_REQUIRE(x_bstride < (1ll << 29) && y_bstride < (1ll << 29) && r_bstride < (1ll << 29), "rt_conv_plan_set_batch_strides: sample exceeds 2 GB");
    if (x_bstride) plan->x_bstride = x_bstride;
    if (y_bstride) {
        if (!plan->r_bstride) plan->r_bstride = plan->y_bstride;      // the residual keeps its own (so far implied) stride
        plan->y_bstride = y_bstride;
    }
    if (r_bstride) plan->r_bstride = r_bstride;
    return 0;
}

namespace {
// Switches a 2-D plan to the fp16-arithmetic kernel: re-packs the weights it was created with.
int repack_f16mma(rtConvPlan* plan) {
    SubConv& sc = plan->subs[0];
    const rtConv2dDesc& d = plan->desc2d;
    const std::vector<float>& w = plan->w_canon;
    const int Cin = d.Cin, Cout = d.Cout;
    sc.CC = 16;
    sc.CinPad = rt::round_up(Cin, 16);
    sc.NBW = 1; sc.TXW = 1;
    sc.TY = sc.NW = 4;       // 8-row tiles (weights staged once per 8 rows) measured the same: 41.7 us either way
    std::vector<uint16_t> packed;
    if (!plan->is_deconv) {
        pack_f16_into(packed, sc, Cin, [&](int co, int ci, int u, int v) { return w[(((size_t)co * Cin + ci) * d.KH + u) * d.KW + v]; });
    } else {
        const int s = d.stride;
        size_t iz = 0;
        for (int py = 0; py < s; py++)
            for (int px = 0; px < s; px++) {
                if (py >= plan->out_dims[1] || px >= plan->out_dims[2]) continue;
                const Phase1D ay = phase1d(s, d.pad_h, d.KH, py), ax = phase1d(s, d.pad_w, d.KW, px);
                const int64_t off = pack_f16_into(packed, sc, Cin, [&](int co, int ci, int u, int v) {
                    if (u >= ay.K || v >= ax.K) return 0.f;
                    return w[(((size_t)ci * Cout + co) * d.KH + ay.tap[u]) * d.KW + ax.tap[v]];
                });
                if (!sc.zs_host.empty()) sc.zs_host[iz++].w_off = off;
            }
        if (!sc.zs_host.empty())
            RT_HIP(hipMemcpy(sc.zs_dev, sc.zs_host.data(), sc.zs_host.size() * sizeof(rt::ZSlice), hipMemcpyHostToDevice));
    }
    if (sc.w_dev) (void)hipFree(sc.w_dev);
    sc.w_dev = nullptr;
    RT_HIP(hipMalloc((void**)&sc.w_dev, packed.size() * 2));
    RT_HIP(hipMemcpy(sc.w_dev, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    if (sc.choff_dev) (void)hipFree(sc.choff_dev);
    sc.choff_dev = nullptr;
    const int ip = sc.x_pitch ? sc.x_pitch : plan->win;
    std::vector<int> table(sc.CinPad, -1);
    for (int c = 0; c < Cin; c++) table[c] = c * plan->hin * ip;
    if (int rc = upload_table(sc, table)) return rc;
    sc.wino = 0; sc.s3p = sc.split3 = sc.s3first = 0; sc.f16mma = 1;
    return 0;
}

// First layer in half2 mode (conv_f16_first.hip.h): weights as the 5 A operands [block][r][h][co][8], k = 3*s + c.
int repack_f16first(rtConvPlan* plan) {
    SubConv& sc = plan->subs[0];
    const rtConv2dDesc& d = plan->desc2d;
    const std::vector<float>& w = plan->w_canon;
    const int nblk = (int)rt::cdiv(d.Cout, 32);
    std::vector<uint16_t> packed((size_t)nblk * rt::ConvF16FirstCfg::W_SLOTS * 8, 0);
    for (int co = 0; co < d.Cout; co++)
        for (int c = 0; c < d.Cin; c++)
            for (int r = 0; r < 5; r++)
                for (int sx = 0; sx < 5; sx++) {
                    const int k = 3 * sx + c, h = k / 8, e = k % 8;
                    const _Float16 hv = (_Float16)w[(((size_t)co * d.Cin + c) * 5 + r) * 5 + sx];
                    uint16_t bits;
                    std::memcpy(&bits, &hv, 2);
                    packed[((((size_t)(co / 32) * 5 + r) * 2 + h) * 32 + co % 32) * 8 + e] = bits;
                }
    if (sc.w_dev) (void)hipFree(sc.w_dev);
    sc.w_dev = nullptr;
    RT_HIP(hipMalloc((void**)&sc.w_dev, packed.size() * 2));
    RT_HIP(hipMemcpy(sc.w_dev, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    sc.NBW = 1; sc.TXW = 1; sc.TY = sc.NW = 4;
    sc.wino = 0; sc.s3p = sc.split3 = sc.s3first = 0; sc.f16first = 1;
    return 0;
}

// Back to the fp32 kernels after repack_f16mma / repack_f16first: the weights the plan was created with are packed
// again in the fp32 slab order, tiling and gather table are rebuilt (an executor that tried half2 mode and has to fall
// back to fp32 activations calls rt_conv_plan_set_io_types(F32, F32) on plans it already switched).
int repack_f32(rtConvPlan* plan) {
    SubConv& sc = plan->subs[0];
    const rtConv2dDesc& d = plan->desc2d;
    const std::vector<float>& w = plan->w_canon;
    const int Cin = d.Cin, Cout = d.Cout;
    sc.f16mma = sc.f16first = sc.s3p = sc.split3 = sc.s3first = 0;
    sc.x_f16 = sc.y_f16 = 0;
    sc.x_il8 = sc.y_il8 = sc.r_il8 = 0;
    choose_tiling(sc, !plan->is_deconv);
    check_direct(sc, Cin);
    sc.CinPad = sc.direct ? Cin : rt::round_up(Cin, sc.CC);
    std::vector<float> packed;
    if (!plan->is_deconv && s3p_eligible(sc, Cin)) {
        if (int rc = upload_s3p(sc, Cin, [&](int co, int ci, int u, int v) { return w[(((size_t)co * Cin + ci) * d.KH + u) * d.KW + v]; })) return rc;
    } else if (!plan->is_deconv && s3first_eligible(sc, Cin, plan->has_resid)) {
        if (int rc = upload_s3first(sc, Cin, [&](int co, int ci, int u, int v) { return w[(((size_t)co * Cin + ci) * d.KH + u) * d.KW + v]; })) return rc;
    } else if (!plan->is_deconv) {
        pack_into(packed, sc, Cin, [&](int co, int ci, int u, int v) { return w[(((size_t)co * Cin + ci) * d.KH + u) * d.KW + v]; });
    } else {
        const int s = d.stride;
        size_t iz = 0;
        for (int py = 0; py < s; py++)
            for (int px = 0; px < s; px++) {
                if (py >= plan->out_dims[1] || px >= plan->out_dims[2]) continue;
                const Phase1D ay = phase1d(s, d.pad_h, d.KH, py), ax = phase1d(s, d.pad_w, d.KW, px);
                const int64_t off = pack_into(packed, sc, Cin, [&](int co, int ci, int u, int v) {
                    if (u >= ay.K || v >= ax.K) return 0.f;
                    return w[(((size_t)ci * Cout + co) * d.KH + ay.tap[u]) * d.KW + ax.tap[v]];
                });
                if (!sc.zs_host.empty()) sc.zs_host[iz++].w_off = off;
            }
        if (!sc.zs_host.empty())
            RT_HIP(hipMemcpy(sc.zs_dev, sc.zs_host.data(), sc.zs_host.size() * sizeof(rt::ZSlice), hipMemcpyHostToDevice));
    }
    if (!sc.s3p && !sc.s3first) {
        if (sc.w_dev) (void)hipFree(sc.w_dev);
        sc.w_dev = nullptr;
        if (int rc = upload_weights(sc, packed)) return rc;
    }
    if (sc.choff_dev) (void)hipFree(sc.choff_dev);
    sc.choff_dev = nullptr;
    const int ip = sc.x_pitch ? sc.x_pitch : plan->win;
    std::vector<int> table(sc.CinPad, -1);
    for (int c = 0; c < Cin; c++) table[c] = c * plan->hin * ip;
    return upload_table(sc, table);
}

// Conv3D in half2 mode: both 4-D tensors fp16 -> conv_f16mma_kernel (fp16 operands, one MFMA per tap, channel-interleaved tensors
// (D, C/8, H, W, 8) welcome) instead of the split kernel reading and writing planar 2-byte elements.  Same gather table (the offset of
// channel group c/8 of depth d equals the planar offset of channel c), same tiling; the weights are re-packed from the plan's copy.
int switch_conv3d_f16mma(rtConvPlan* plan, bool on) {
    SubConv& sc = plan->subs[0];
    if ((sc.f16mma != 0) == on) return 0;
    const int cin_real = plan->c3d_cin, taps = sc.KH * sc.KW;
    const std::vector<float>& w = plan->w_canon;
    auto wfun = [&](int co, int ci, int u, int v) { return w[((size_t)co * cin_real + ci) * taps + u * sc.KW + v]; };
    if (sc.w_dev) (void)hipFree(sc.w_dev);
    sc.w_dev = nullptr;
    if (on) {
        std::vector<uint16_t> packed;
        pack_f16_into(packed, sc, cin_real, wfun);
        RT_HIP(hipMalloc((void**)&sc.w_dev, packed.size() * 2));
        RT_HIP(hipMemcpy(sc.w_dev, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
        sc.split3 = 0; sc.f16mma = 1;
    } else {
        sc.f16mma = 0; sc.split3 = 1;
        sc.x_il8 = sc.y_il8 = sc.r_il8 = 0;
        std::vector<float> packed;
        pack_into(packed, sc, cin_real, wfun);
        if (int rc = upload_weights(sc, packed)) return rc;
    }
    return 0;
}

bool f16mma_window(const SubConv& sc) {
    return (sc.KH == 3 && sc.KW == 3 && (sc.S == 1 || sc.S == 2)) || (sc.S == 1 && sc.KH <= 2 && sc.KW <= 2);
}
}  // namespace

// Storage type (RT_F32 / RT_F16) of the input and of the output + residual of a 2-D plan: TensorRT's half2 mode keeps
// activations in fp16 between layers; the arithmetic stays fp32.  fp16 outputs need an even row pitch (pixel pairs
// are written as one 4-byte word).
extern "C" int rt_conv_plan_set_io_types(rtConvPlan* plan, int x_dtype, int y_dtype) {
    RT_REQUIRE(plan, "rt_conv_plan_set_io_types: null plan");
    ExactScope exact_scope(plan->flags);       // re-planning (repack_f32: choose_tiling, s3first / s3p eligibility) under the plan's own options
    RT_REQUIRE((x_dtype == RT_F32 || x_dtype == RT_F16) && (y_dtype == RT_F32 || y_dtype == RT_F16), "rt_conv_plan_set_io_types: bad dtype");
    plan->softarg = 0;
    if (!plan->is2d) {
        // 3-D plans (Conv3D / Conv3DTranspose): fp16 storage of the dense (D,C,H,W) / (K,D,H,W) tensors, the split-fp16 kernel
        // reads / writes them as they are (an fp16 input is its own high part); the small-output last layer reads fp16, writes fp32
        // (a Conv3D whose INPUT is then declared channel-interleaved -- rt_conv_plan_set_layouts -- moves to fp16 operands,
        //  conv_f16mma_kernel; a change of the storage types takes it back to the split kernel and planar tensors first)
        const bool both = x_dtype == RT_F16 && y_dtype == RT_F16;
        if (plan->is_conv3d && plan->subs.size() == 1 && plan->subs[0].f16mma)
            if (int rc = switch_conv3d_f16mma(plan, false)) return rc;
        if (plan->is_deconv3d && !plan->subs.empty() && !plan->subs[0].small3d) {   // back to the split kernel and planar tensors first
            const bool p4 = x_dtype == RT_F32 && y_dtype == RT_F32 && deconv3d_p4f32_ok(plan);      // (fp32 tensors: its four-phase form)
            if (plan->subs[0].f16mma || (plan->subs[0].dp4 != 0) != p4)
                if (int rc = build_deconv3d_subs(plan, false, p4)) return rc;
        }
        for (SubConv& sc : plan->subs) {
            // (an fp16 input with an fp32 output has no kernel: ADVICE r02 -- say so here, not at the first enqueue)
            const bool ok = sc.small3d ? (sc.small3d == 1 && y_dtype == RT_F32) : (sc.f16mma ? both : (sc.split3 && (y_dtype == RT_F16 || x_dtype == RT_F32)));
            if (!ok && (x_dtype == RT_F16 || y_dtype == RT_F16))
                return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_io_types: this 3-D plan has no kernel for %s input and %s output (window %dx%d stride %d)",
                            x_dtype == RT_F16 ? "fp16" : "fp32", y_dtype == RT_F16 ? "fp16" : "fp32", sc.KH, sc.KW, sc.S);
        }
        for (SubConv& sc : plan->subs) {
            sc.x_f16 = x_dtype == RT_F16; sc.y_f16 = y_dtype == RT_F16;
            if (!sc.y_f16) sc.y_il8 = sc.r_il8 = 0;            // interleaved 3-D tensors are fp16 ones
            if (!sc.x_f16) sc.x_il8 = 0;
        }
        return 0;
    }
    const int xf = x_dtype == RT_F16, yf = y_dtype == RT_F16;
    const int op = plan->out_pitch ? plan->out_pitch : plan->out_dims[2];
    const bool s3_f32_to_f16 = !xf && yf && plan->subs.size() == 1 && plan->subs[0].split3;      // single 2-byte / interleaved 8-byte stores: any pitch
    RT_REQUIRE(!yf || s3_f32_to_f16 || (op & 1) == 0, "rt_conv_plan_set_io_types: fp16 output rows need an even pitch (got %d)", op);
    for (SubConv& sc : plan->subs) {
        if (!xf && !yf) {
            // a plan that was switched to fp16 operands holds fp16 weight slabs and the fp16 tiling: restore the fp32 form
            if ((sc.f16mma || sc.f16first) && plan->subs.size() == 1) { if (int rc = repack_f32(plan)) return rc; }
            sc.x_f16 = sc.y_f16 = 0;
            continue;
        }
        if (sc.direct) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_io_types: the direct (Cout <= 2) kernel is fp32 only");
        // both tensors fp16: fp16 operands on the matrix cores (the stored values are the operands, fp32 accumulate)
        // (its gathers move 4-byte pixel pairs: row pitch, plane and sample strides must be even)
        const int xp = sc.x_pitch ? sc.x_pitch : sc.Wi;
        if (xf && yf && !sc.small3d && !sc.f16mma && !sc.rb && plan->subs.size() == 1 && f16mma_window(sc) && xp % 2 == 0 &&
            plan->x_bstride % 2 == 0 && env_int("RT_NO_F16MMA", 0) == 0) {
            if (int rc = repack_f16mma(plan)) return rc;
        }
        if (sc.f16mma) {
            if (!(xf && yf)) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_io_types: plan was switched to fp16 arithmetic, both tensors must stay fp16");
            sc.x_f16 = sc.y_f16 = 1;
            continue;
        }
        // the network's first layer (5x5 stride 2 on the 3-channel fp32 image): fp16 operands as well
        if (!xf && yf && !sc.f16first && !plan->is_deconv && plan->subs.size() == 1 && sc.KH == 5 && sc.KW == 5 && sc.S == 2 &&
            plan->cin <= 3 && !plan->has_resid && !sc.direct && !sc.zs_dev && env_int("RT_NO_F16MMA", 0) == 0) {
            if (int rc = repack_f16first(plan)) return rc;
        }
        if (sc.f16first) {
            if (xf || !yf) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_io_types: plan was switched to the fp16 first-layer kernel (fp32 in, fp16 out)");
            sc.x_f16 = 0; sc.y_f16 = 1;
            continue;
        }
        // fp32 in, fp16 out on the split kernel (3x3 stride 1): the last layer of a feature tower of a 3-D model in half2 mode writes the
        // fp16 feature map the folded-cost-volume Conv3D reads (conv_s3_kernel<.., float, _Float16>)
        if (sc.split3 && !xf && yf && sc.KH == 3 && sc.KW == 3 && sc.S == 1 && !plan->is_deconv && !plan->has_resid && sc.TY == 4) {
            sc.x_f16 = 0; sc.y_f16 = 1;
            continue;
        }
        // the tower block in half2 mode: both tensors fp16, fp16 operands, one launch (conv_f16rbd_kernel; interleaved tensors, checked at enqueue)
        if (sc.rb && xf && yf && plan->rbh_w1_dev && plan->cin == 32 && plan->rb_cmid == 32 && sc.Cout == 32 && plan->rb_act1 == 1 && plan->act == 1 &&
            env_int("RT_NO_F16MMA", 0) == 0 && env_int("RT_NO_RBH", 0) == 0) {
            sc.x_f16 = sc.y_f16 = 1;
            continue;
        }
        if (sc.s3p || sc.split3 || sc.s3first || sc.rb) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_io_types: the split-fp16 kernels take fp32 tensors (both fp16: fp16 operands instead)");
        if (sc.small3d) { if (!(sc.small3d == 2 && xf)) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_io_types: unsupported combination for the small-output kernel"); }
        else if (sc.wino) { if (!(xf && yf)) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_io_types: Winograd layers take fp16 on both sides"); }
        else if (!yf) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_io_types: fp16 -> fp32 is only built for the small-output kernel");
        else set_tile(sc, 6);                  // the fp16 instantiations exist for the default tile
        sc.x_f16 = xf; sc.y_f16 = yf;
    }
    return 0;
}

// Channel-interleaved tensors (see conv_wino.hip.h / conv_f16.hip.h): which of a plan's tensors can have the layout --
// bit 0 input, bit 1 output, bit 2 residual; 0 = none ...
extern "C" int rt_conv_plan_supports_il8(const rtConvPlan* plan) {
    if (!plan) return 0;
    ExactScope exact_scope(plan->flags);
    if (!plan->is2d) {
        // 3-D plans, half2 mode: fp16 tensors in depth-major form (D, C, H, W) may be stored (D, C/8, H, W, 8).  Conv3D on fp16
        // operands reads and (with the fused Transform, out_dchw) writes them; the folded-cost-volume Conv3D (fp32 maps in) writes
        // one; a fused Conv3DTranspose reads its skip tensor that way.  Channel-major (K, D, H, W) tensors stay planar.
        if (env_int("RT_NO_IL8", 0) != 0 || env_int("RT_NO_IL8_3D", 0) != 0 || plan->subs.empty()) return 0;
        const SubConv& sc = plan->subs[0];
        if (plan->is_conv3d && plan->subs.size() == 1 && plan->ff && !sc.x_f16 && !sc.f16mma)
            // the factored cost-volume fold: planar fp32 feature maps in (bit 5: keep them so), interleaved output in either storage type
            return 32 | 2;
        if (plan->is_conv3d && plan->subs.size() == 1) {
            // the fused Transform's (D, K/8, H, W, 8), or -- fp16 operands only -- the plain (K/8, D, H, W, 8) a Conv3DTranspose then reads
            const int out_dm = (plan->c3d_dchw && sc.Cout % 8 == 0) ? 2 : 0, out = sc.Cout % 8 == 0 ? 2 : 0;
            // both tensors fp16: conv_f16mma_kernel once the input is interleaved (its planar gathers move 4-byte pixel pairs, which the
            // odd plane sizes of dense 4-D tensors misalign) -- bit 3: an interleaved output needs an interleaved input
            const bool f16mma_ok = !sc.small3d && !sc.direct && (sc.split3 || sc.f16mma) && sc.KH == 3 && sc.KW == 3 &&
                                   (sc.S == 1 || sc.S == 2) && sc.TY == 4 && plan->c3d_C % 8 == 0 && env_int("RT_NO_F16MMA_3D", 0) == 0;
            // (bit 2: the skip tensor of a fused add may be interleaved like the output -- the fp16-operand kernels read it either way)
            if (sc.x_f16 && sc.y_f16) return f16mma_ok ? (1 | out | (out ? 8 : 0) | ((out && plan->has_resid) ? 4 : 0)) : 0;
            if (sc.split3 && sc.y_f16 && !sc.x_f16 && sc.TY == 4) return out_dm;         // fp32 (feature maps) in, fp16 volume out
            // fp32 tensors (round 4): the split kernel takes (D, C/4, H, W, 4) on either side -- one 16-byte load per pixel and group of 4
            // channels instead of four 4-byte ones, 16-byte stores.  The folded cost volume's two feature maps: (2F/4, H, W, 4), the right
            // half read shifted by the slice's disparity in whole 16-byte slots.
            if (sc.split3 && !sc.x_f16 && !sc.y_f16 && sc.TY == 4 && sc.KH == 3 && sc.KW == 3 && env_int("RT_NO_IL8_3D_F32", 0) == 0) {
                const int o4 = (plan->c3d_dchw && sc.Cout % 4 == 0) ? 2 : 0;
                return ((plan->c3d_C % 4 == 0 && plan->c3d_fold % 4 == 0) ? 1 : 0) | o4 | ((o4 && plan->has_resid) ? 4 : 0);
            }
            return 0;
        }
        if (plan->is_deconv3d) {
            if (sc.small3d) {    // last layer: (K/8, Dy, Hy, Wy, 8) fp16 in on the matrix cores (deconv3d_s2_il_kernel), fp32 volume out;
                                 // fp32 engines: (K/4, Dy, Hy, Wy, 4) fp32 in, split form (deconv3d_s2_il4_kernel)
                if (sc.small3d != 1 || sc.y_f16 || sc.s3.K % 32 != 0 || sc.small_w.empty() || env_int("RT_NO_SMALL_IL", 0) != 0) return 0;
                if (!sc.x_f16 && ((plan->flags & RT_CONV_EXACT_FP32) || env_int("RT_NO_IL8_3D_F32", 0) != 0 || env_int("RT_NO_SMALL_IL_F32", 0) != 0)) return 0;
                return 1;
            }
            bool all_f32 = true;
            for (const SubConv& q : plan->subs) all_f32 = all_f32 && q.split3 && !q.x_f16 && !q.y_f16 && q.TY == 4;
            if (all_f32) {         // fp32 tensors: the skip tensor (D, C/4, H, W, 4) (ZSlice::r_off_il4); the four-phase form also WRITES an
                                   // interleaved tensor (for the last layer's matrix-core kernel); the input stays planar
                if (plan->c3d_C % 4 != 0 || env_int("RT_NO_IL8_3D_F32", 0) != 0) return 0;
                return (plan->has_resid ? 4 : 0) | (sc.dp4 ? 2 : 0);
            }
            for (const SubConv& q : plan->subs)
                if (!(q.split3 || q.f16mma) || !q.y_f16 || q.TY != 4) return 0;
            // fp16 in and out: an interleaved INPUT moves the plan to fp16 operands (conv_f16mma_kernel, 2x2 phase windows), which also
            // writes interleaved outputs (bit 3: only then); the skip tensor may be interleaved with either kernel
            const bool f16mma_ok = sc.x_f16 && ((sc.KH == 2 && sc.KW == 2) || sc.dp4) && plan->desc3d.K % 8 == 0 && env_int("RT_NO_F16MMA_3D", 0) == 0 &&
                                   env_int("RT_NO_DECONV_IL", 0) == 0;
            int caps = (plan->has_resid && plan->c3d_C % 8 == 0) ? 4 : 0;
            if (f16mma_ok) caps |= 1 | (plan->c3d_C % 8 == 0 ? 2 | 8 : 0);
            return caps;
        }
        return 0;
    }
    if (plan->subs.size() != 1) return 0;
    // the general split-fp16 kernel (any 2-D window, transposed phases included) takes and writes them freely
    if (plan->subs[0].split3 && env_int("RT_NO_IL8", 0) == 0) {
        if (plan->subs[0].y_f16) return (plan->cin % 4 == 0 ? 1 : 0) | (plan->subs[0].Cout % 8 == 0 ? 2 : 0);      // fp16 output: groups of 8, no residual
        // bit 4 (16): an interleaved input whose channel count is PADDED to a multiple of 4 -- the gather reads whole 16-byte groups and
        // the pad channels meet zero weights, so they must hold finite values (the 33-channel input of conv2D_1: 32 feature channels
        // and the soft-argmax map in lane 0 of a ninth group)
        return (plan->cin % 4 == 0 ? 1 : 16) | (plan->subs[0].Cout % 4 == 0 ? 6 : 0);
    }
    if (plan->is_deconv) return 0;
    const SubConv& sc = plan->subs[0];
    if (env_int("RT_NO_IL8", 0) != 0 || sc.zs_dev || sc.y_xstride != 1 || sc.small3d || sc.direct) return 0;
    if (sc.s3p) return (plan->cin % 4 == 0 ? 1 : 0) | (sc.Cout % 4 == 0 ? 6 : 0);
    if (sc.rb) return (plan->cin % 4 == 0 ? 5 : 0) | (sc.Cout % 4 == 0 ? 2 : 0);      // the residual IS the input tensor
    if (sc.s3first) return sc.Cout % 4 == 0 ? 2 : 0;           // output only (its input is the image binding)
    if (sc.f16first) return sc.Cout % 8 == 0 ? 2 : 0;          // output only (its input is the fp32 image)
    // (bit 4: an interleaved input padded to a whole group of 8 -- conv2D_1's 33 channels in half2 mode; the gather loads whole groups,
    //  the pad channels meet zero weights and must hold finite values, as for the fp32 kernel above)
    if (sc.f16mma) return (sc.KH == 3 && sc.KW == 3 && sc.S == 1 && sc.Cout % 8 == 0) ? ((plan->cin % 8 == 0 ? 1 : 16) | 6) : 0;
    if (sc.x_f16 || sc.y_f16) return 0;
    // fp32 tensors: groups of 4 channels.  The Winograd kernel with the 4-wave tile takes all three.  (Round 3 took the interleaved
    // instantiations out of the product after tools/race_hunt.py saw one output in ~3000 of the exact engine deviate beside other contexts;
    // round 4 located it -- compiler-formed packed fp32 math in the interleaved epilogue beside co-resident fp16-MFMA waves -- and the
    // library is built without the SLP vectoriser since: profiles/r04_race.txt.  RT_NO_WINO_IL8=1: planar exact engines, for A/B runs.)
    if (sc.wino) return (env_int("RT_NO_WINO_IL8", 0) == 0 && sc.TY == 4 && plan->cin % 4 == 0 && sc.Cout % 4 == 0) ? 7 : 0;
    // (the direct-form kernel had an interleaved-output form for the first layer and the stride-2 layers behind RT_IL_DIRECT in round 1;
    //  whole networks were wrong with it on the GPU only -- tools/race_hunt.py reproduced that in round 2 even with one stream and one
    //  context -- and the split-fp16 kernels that now serve those layers write interleaved tensors themselves, so the form was removed)
    return 0;
}
// ... and the layout of each of them (0 = planar NCHW with a row pitch, 1 = (C/4, H, pitch, 4) fp32 / (C/8, H, pitch, 8) fp16)
extern "C" int rt_conv_plan_set_layouts(rtConvPlan* plan, int x_il8, int y_il8, int r_il8) {
    RT_REQUIRE(plan, "rt_conv_plan_set_layouts: null plan");
    ExactScope exact_scope(plan->flags);
    // A fused soft-argmax and pre-split tensors (rt_resblock_plan_set_split) are declared last, on the final types and layouts: a
    // SUCCESSFUL change of the layouts resets them; a refused one leaves the plan as it was.
    struct KeepOnFailure {
        rtConvPlan* p; int softarg, xs, ys; bool ok = false;
        ~KeepOnFailure() { if (!ok) { p->softarg = softarg; p->x_split = xs; p->y_split = ys; } }
    } keep{plan, plan->softarg, plan->x_split, plan->y_split};
    plan->softarg = 0;
    plan->x_split = plan->y_split = 0;
    if (!x_il8 && !y_il8 && !r_il8) {
        if (plan->is_conv3d && plan->subs.size() == 1 && plan->subs[0].f16mma)
            if (int rc = switch_conv3d_f16mma(plan, false)) return rc;
        if (plan->is_deconv3d && !plan->subs.empty() && plan->subs[0].f16mma) {
            const int xf = plan->subs[0].x_f16, yf = plan->subs[0].y_f16;
            if (int rc = build_deconv3d_subs(plan, false, !xf && !yf && deconv3d_p4f32_ok(plan))) return rc;
            for (SubConv& q : plan->subs) { q.x_f16 = xf; q.y_f16 = yf; }
        }
        for (SubConv& sc : plan->subs) sc.x_il8 = sc.y_il8 = sc.r_il8 = 0;
        keep.ok = true;
        return 0;
    }
    const int caps = rt_conv_plan_supports_il8(plan);
    if (plan->is_conv3d && plan->subs.size() == 1 && plan->subs[0].x_f16 && plan->subs[0].y_f16) {
        if ((caps & 8) && y_il8 && !x_il8) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_layouts: this Conv3D writes an interleaved tensor only when it reads one");
        if ((x_il8 && !(caps & 1)) || (y_il8 && !(caps & 2))) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_layouts: this Conv3D plan takes no interleaved tensors");
        // (every capability check before the plan is re-packed)
        if (r_il8 && (!(caps & 4) || !x_il8)) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_layouts: this Conv3D reads an interleaved skip tensor only on fp16 operands (interleaved input)");
        if (int rc = switch_conv3d_f16mma(plan, x_il8 != 0)) return rc;           // interleaved input: fp16 operands; planar input: the split kernel
        SubConv& sc = plan->subs[0];
        sc.x_il8 = x_il8 != 0; sc.y_il8 = y_il8 != 0; sc.r_il8 = r_il8 != 0;
        keep.ok = true;
        return 0;
    }
    if (plan->is_deconv3d && !plan->subs.empty()) {
        if ((x_il8 && !(caps & 1)) || (y_il8 && !(caps & 2)) || (r_il8 && !(caps & 4)) || ((caps & 8) && y_il8 && !x_il8))
            return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_layouts: this Conv3DTranspose plan does not take that combination of interleaved tensors (caps %d)", caps);
        RT_REQUIRE(!r_il8 || plan->has_resid, "rt_conv_plan_set_layouts: plan has no residual");
        if (plan->subs[0].small3d) {
            SubConv& sc = plan->subs[0];
            const bool f32in = !sc.x_f16;
            if (x_il8 && sc.small_il_dev && sc.small_il_f32 != (f32in ? 1 : 0)) { (void)hipFree(sc.small_il_dev); sc.small_il_dev = nullptr; }
            if (x_il8 && !sc.small_il_dev && f32in) {
                // fp32 input: the same lane images twice -- fp16 high parts, then the scaled fp16 low parts (split_f16)
                const int K = sc.s3.K, CO = sc.Cout, KC = K / 32;
                std::vector<uint16_t> slab((size_t)2 * 8 * KC * 64 * 8, 0);
                for (int j = 0; j < 8; j++)
                    for (int kc = 0; kc < KC; kc++)
                        for (int l = 0; l < 64; l++)
                            for (int e = 0; e < 8; e++) {
                                const int row = l % 16, co = row / 8, f = row % 8, k = kc * 32 + 8 * (l / 16) + e;
                                if (co >= CO) continue;
                                uint16_t hi, lo;
                                split_f16(sc.small_w[(((size_t)k * CO + co) * 8 + f) * 8 + j], hi, lo);
                                slab[(((size_t)j * KC + kc) * 64 + l) * 8 + e] = hi;
                                slab[(((size_t)(8 + j) * KC + kc) * 64 + l) * 8 + e] = lo;
                            }
                RT_HIP(hipMalloc(&sc.small_il_dev, slab.size() * 2));
                RT_HIP(hipMemcpy(sc.small_il_dev, slab.data(), slab.size() * 2, hipMemcpyHostToDevice));
                sc.small_il_f32 = 1;
            }
            if (x_il8 && !sc.small_il_dev) {
                sc.small_il_f32 = 0;
                // A operands of deconv3d_s2_il_kernel: per neighbour j and block of 32 input channels a 16 x 32 tile [row = co * 8 + phase][k],
                // stored as the MFMA's lane image: lane l holds row l % 16, channels 8 * (l / 16) .. + 7
                const int K = sc.s3.K, CO = sc.Cout, KC = K / 32;
                std::vector<uint16_t> slab((size_t)8 * KC * 64 * 8, 0);
                for (int j = 0; j < 8; j++)
                    for (int kc = 0; kc < KC; kc++)
                        for (int l = 0; l < 64; l++)
                            for (int e = 0; e < 8; e++) {
                                const int row = l % 16, co = row / 8, f = row % 8, k = kc * 32 + 8 * (l / 16) + e;
                                if (co >= CO) continue;
                                const _Float16 h = (_Float16)sc.small_w[(((size_t)k * CO + co) * 8 + f) * 8 + j];
                                std::memcpy(&slab[(((size_t)j * KC + kc) * 64 + l) * 8 + e], &h, 2);
                            }
                RT_HIP(hipMalloc(&sc.small_il_dev, slab.size() * 2));
                RT_HIP(hipMemcpy(sc.small_il_dev, slab.data(), slab.size() * 2, hipMemcpyHostToDevice));
            }
            sc.x_il8 = x_il8 != 0;
            keep.ok = true;
            return 0;
        }
        const bool f32 = !plan->subs[0].x_f16 && !plan->subs[0].y_f16;
        const bool want = x_il8 != 0, want_dp = want ? (y_il8 && deconv3d_dp4_ok(plan)) : (f32 && deconv3d_p4f32_ok(plan));
        if ((plan->subs[0].f16mma != 0) != want || (plan->subs[0].dp4 != 0) != want_dp) {
            const int xf = plan->subs[0].x_f16, yf = plan->subs[0].y_f16;
            if (int rc = build_deconv3d_subs(plan, want, want_dp)) return rc;
            for (SubConv& q : plan->subs) { q.x_f16 = xf; q.y_f16 = yf; }
        }
        for (SubConv& q : plan->subs) { q.x_il8 = x_il8 != 0; q.y_il8 = y_il8 != 0; q.r_il8 = r_il8 != 0; }
        keep.ok = true;
        return 0;
    }
    if ((x_il8 && !(caps & (1 | 16))) || (y_il8 && !(caps & 2)) || (r_il8 && !(caps & 4)))
        return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_layouts: this plan does not take an interleaved %s tensor (3x3 stride-1 plans take all three; "
                    "the first layer and stride-2 3x3 layers write one; channel counts must be multiples of 4 in fp32, 8 in fp16)",
                    x_il8 && !(caps & (1 | 16)) ? "input" : (y_il8 && !(caps & 2) ? "output" : "residual"));
    RT_REQUIRE(!r_il8 || plan->has_resid, "rt_conv_plan_set_layouts: plan has no residual");
    SubConv& sc = plan->subs[0];
    RT_REQUIRE(!sc.rb || (x_il8 != 0) == (r_il8 != 0), "rt_conv_plan_set_layouts: a residual block's residual is its input tensor");
    for (SubConv& q : plan->subs) { q.x_il8 = x_il8 != 0; q.y_il8 = y_il8 != 0; q.r_il8 = r_il8 != 0; }     // (3-D transposed plans have one sub per depth class)
    keep.ok = true;
    return 0;
}

extern "C" int rt_conv_plan_input_limit(const rtConvPlan* plan, float* limit) {
    RT_REQUIRE(plan && limit, "rt_conv_plan_input_limit: null pointer");
    bool f16_pipe = plan->rb_w1_dev != nullptr;
    for (const SubConv& sc : plan->subs) f16_pipe = f16_pipe || sc.split3 || sc.s3first || sc.s3p || sc.rb || sc.x_f16 || sc.f16mma;
    *limit = f16_pipe ? 65504.f : __builtin_inff();
    return 0;
}

extern "C" int rt_check_range(const void* x, int64_t rows, int64_t valid, int64_t pitch, int dtype, float limit, float* max_abs, int64_t* violations,
                              rtStream s) {
    RT_REQUIRE(x && rows >= 0 && valid >= 0 && pitch >= valid && (dtype == RT_F32 || dtype == RT_F16), "rt_check_range: bad arguments");
    unsigned* out = nullptr;
    RT_HIP(hipMalloc((void**)&out, 16));
    int rc = 0;
    if (hipMemsetAsync(out, 0, 16, S(s)) != hipSuccess) rc = fail(RT_E_RUNTIME, "rt_check_range: memset failed");
    const int64_t n = rows * valid;
    if (!rc && n > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>(rt::cdiv(n, 256), 2048);
        if (dtype == RT_F16) hipLaunchKernelGGL((rt::range_check_kernel<_Float16>), dim3(grid), dim3(256), 0, S(s), static_cast<const _Float16*>(x), rows, valid, pitch, limit, out);
        else hipLaunchKernelGGL((rt::range_check_kernel<float>), dim3(grid), dim3(256), 0, S(s), static_cast<const float*>(x), rows, valid, pitch, limit, out);
        if (hipGetLastError() != hipSuccess) rc = fail(RT_E_RUNTIME, "rt_check_range: launch failed");
    }
    unsigned host[4] = {0, 0, 0, 0};
    if (!rc && (hipMemcpyAsync(host, out, 16, hipMemcpyDeviceToHost, S(s)) != hipSuccess || hipStreamSynchronize(S(s)) != hipSuccess))
        rc = fail(RT_E_RUNTIME, "rt_check_range: read-back failed");
    (void)hipFree(out);
    if (rc) return rc;
    float m;
    std::memcpy(&m, &host[0], 4);
    if (max_abs) *max_abs = m;
    if (violations) *violations = (int64_t)(((unsigned long long)host[3] << 32) | host[2]);
    return 0;
}

extern "C" int rt_hash_buffer(const void* x, size_t bytes, unsigned long long* out_dev, rtStream s) {
    RT_REQUIRE(x && out_dev && bytes % 4 == 0, "rt_hash_buffer: null pointer or a size that is not a multiple of 4");
    RT_HIP(hipMemsetAsync(out_dev, 0, 8, S(s)));
    if (!bytes) return 0;
    const int64_t words = (int64_t)(bytes / 4);
    const unsigned grid = (unsigned)std::min<int64_t>(rt::cdiv(words, 256 * 8), 1024);
    hipLaunchKernelGGL(rt::hash_words_kernel, dim3(grid ? grid : 1), dim3(256), 0, S(s), static_cast<const unsigned*>(x), words, out_dev);
    RT_LAUNCH_CHECK("hash_words_kernel");
    return 0;
}

extern "C" int rt_conv_plan_out_dims(const rtConvPlan* plan, int dims[4]) {
    RT_REQUIRE(plan && dims, "rt_conv_plan_out_dims: null pointer");
    for (int i = 0; i < 4; i++) dims[i] = plan->out_dims[i];
    return 0;
}

extern "C" int rt_conv_enqueue(const rtConvPlan* plan, const void* x, void* y, const void* residual, int batch,
                               rtStream s) {
    return rt_conv_enqueue_hint(plan, x, y, residual, batch, s, 0);
}

extern "C" size_t rt_conv_plan_workspace_bytes(const rtConvPlan* plan, int batch) {
    if (!plan || batch <= 0 || !plan->ff) return 0;
    return fold_factor_bytes(plan, batch);           // (a plan that has the factored form may still launch the gather form: the bytes are then unused)
}

// The soft-argmax over the output depth that follows the last Conv3DTranspose of a 3-D model, inside the launch (deconv3d_s2_ilw_kernel<SA>):
// mode 1 = soft-argmax, 2 = soft-argmin, 0 = off.  rt_conv_enqueue then writes the (batch, 1, H, W) fp32 map to y (plain pitch W) and the
// volume does not exist.  Only the depth-walking last layer has the form: one output channel, 32 input channels, an fp16 channel-interleaved
// input, an fp32 output, no residual.  Anything else: RT_E_UNSUPPORTED and the plan is unchanged (the caller keeps rt_softargmax).
// Declared LAST: rt_conv_plan_set_io_types / _set_layouts switch it off again.
extern "C" int rt_conv_plan_set_softarg(rtConvPlan* plan, int mode) {
    RT_REQUIRE(plan, "rt_conv_plan_set_softarg: null plan");
    RT_REQUIRE(mode >= 0 && mode <= 2, "rt_conv_plan_set_softarg: mode must be 0, 1 (max) or 2 (min)");
    if (mode == 0) { plan->softarg = 0; return 0; }
    const bool ok = plan->is_deconv3d && plan->subs.size() == 1 && plan->subs[0].small3d == 1 && plan->subs[0].x_f16 && plan->subs[0].x_il8 &&
                    !plan->subs[0].y_f16 && plan->subs[0].s3.K == 32 && plan->subs[0].s3.C == 1 && !plan->has_resid &&
                    env_int("RT_SMALL_IL_WALK", -1) != 0 && env_int("RT_SOFTARG_FUSE", 1) != 0;
    if (!ok) return fail(RT_E_UNSUPPORTED, "rt_conv_plan_set_softarg: only the depth-walking last Conv3DTranspose (1 output channel, 32 interleaved fp16 "
                         "input channels, fp32 output, no residual) ends in a soft-argmax");
    plan->softarg = mode;
    return 0;
}

extern "C" int rt_conv_enqueue_hint(const rtConvPlan* plan, const void* x, void* y, const void* residual, int batch,
                                    rtStream s, int hints) {
    return rt_conv_enqueue_ws(plan, x, y, residual, batch, nullptr, 0, s, hints);
}

// The first layer of BOTH feature towers in one launch: samples 0 .. batch - 1 are read from x (the left images), samples batch .. 2 batch - 1
// from x2 (the right images, a separate binding: reference sample_app/main.cpp:290-300), y holds 2 * batch samples.  The towers share their
// weights (resnet18_2D_513x257_net.cpp:48-64 / 320-336 read the same tensors), so this is the launch over [left | right] the executor's
// siamese merge makes of every other tower layer.  Only the first-layer kernels have the form (5x5 stride 2 on <= 3 channels).
namespace { thread_local const void* tl_twin_x2 = nullptr; thread_local int tl_twin_from = 0; }
extern "C" int rt_conv_plan_supports_twin_input(const rtConvPlan* plan) {
    return plan && plan->subs.size() == 1 && (plan->subs[0].s3first || plan->subs[0].f16first) && !plan->has_resid && env_int("RT_NO_TWIN_INPUT", 0) == 0;
}
extern "C" int rt_conv_enqueue_twin_input(const rtConvPlan* plan, const void* x, const void* x2, void* y, int batch, rtStream s, int hints) {
    RT_REQUIRE(plan && x && x2 && y && batch > 0, "rt_conv_enqueue_twin_input: bad arguments");
    if (!rt_conv_plan_supports_twin_input(plan)) return fail(RT_E_UNSUPPORTED, "rt_conv_enqueue_twin_input: only the first-layer kernels read two input tensors");
    tl_twin_x2 = x2; tl_twin_from = batch;
    const int rc = rt_conv_enqueue_ws(plan, x, y, nullptr, 2 * batch, nullptr, 0, s, hints);
    tl_twin_x2 = nullptr; tl_twin_from = 0;
    return rc;
}

extern "C" int rt_conv_enqueue_ws(const rtConvPlan* plan, const void* x, void* y, const void* residual, int batch, void* workspace,
                                  size_t workspace_bytes, rtStream s, int hints) {
    RT_REQUIRE(plan && x && y, "rt_conv_enqueue: null pointer");
    RT_REQUIRE(batch > 0, "rt_conv_enqueue: batch must be positive");
    RT_REQUIRE(!plan->has_resid || residual || plan->rb_w1_dev, "rt_conv_enqueue: plan expects a residual tensor");
    RT_REQUIRE(!plan->rb_w1_dev || !residual || residual == x, "rt_conv_enqueue: a residual block's skip connection is its input tensor");
    std::call_once(plan->env_once, [&] {        // execution contexts of one engine share the plan and may launch it from different threads
        plan->opt_xcd = env_int("RT_CONV_XCD", 1); plan->opt_trace = env_int("RT_CONV_TRACE", 0);
        plan->opt_rb_tiles = exp_knob("RT_RB_TILES", 0); plan->opt_rbs_seg = env_int("RT_RBS_SEG", 0); plan->opt_ksplit = env_int("RT_S3_KSPLIT", -1);
        plan->opt_s3p_grid = env_int("RT_S3P_GRID", 0);
        plan->opt_zinner = env_int("RT_Z_INNER", 1);     // 3-D launches: depth slices fastest inside a tile (ConvArgs::z_inner); 0 = z outermost
        plan->opt_nbinner = env_int("RT_NB_INNER", 1);  // 3-D launches: blocks of 32 output channels fastest (ConvArgs::nb_inner); 0 = grid.y
        plan->opt_dw = env_int("RT_F16_DW", -1);         // -1: where it applies (3x3x3 stride-1 Conv3D between interleaved fp16 tensors), 0: never
        plan->opt_dw_nseg = env_int("RT_DW_NSEG", 0);   // depth segments per tile pair (0: chosen from the grid)
        plan->opt_fold_u = env_int("RT_FOLD_U", 2);           // factored cost-volume fold: depth slices per trip of the combining pass (MI355X, NVSmall b8: 1: 0.84, 2: 0.68, 4: 1.07 ms)
        plan->opt_f16p_classes = env_int("RT_F16P_CLASSES", 1);   // ... 1: both depth classes (even / odd output depths) in one walk, 0: a launch per class
        plan->opt_f16p_walk = env_int("RT_F16P_WALK", -1);        // transposed fp16 layers, four phases per workgroup: -1 walk down the class's depths (segments chosen), 0 one depth per workgroup, n > 0: n segments
        plan->opt_small_walk = env_int("RT_SMALL_IL_WALK", -1);   // last transposed layer on interleaved fp16 input: 0 = one depth block per workgroup
        plan->opt_r4 = env_int("RT_F16_R4", -1);         // -1: where it pays (3-D plans), 0: never, 1: every 3x3 stride-1 fp16 launch on interleaved tensors
    });
    if (fold_factor_active(plan)) return enqueue_fold_factor(plan, x, y, batch, workspace, workspace_bytes, s, hints);
    bool skip_sub = false;               // the previous sub-plan's launch covered this one too (two depth classes of a transposed layer in one walk)
    for (const SubConv& sc : plan->subs) {
        if (skip_sub) { skip_sub = false; continue; }
        if (sc.small3d) {
            rt::Deconv3dSmallArgs a = sc.s3;
            a.x = static_cast<const float*>(x); a.y = static_cast<float*>(y); a.w = sc.w_dev; a.bias = plan->bias_dev;
            a.resid = plan->has_resid ? static_cast<const float*>(residual) : nullptr;
            a.act = plan->act; a.x_bstride = plan->x_bstride; a.y_bstride = plan->y_bstride;
            const int64_t gz = (int64_t)batch * a.Mz;
            RT_REQUIRE(gz <= 65535 && (a.Hx + 1) / 2 <= 65535, "rt_conv_enqueue: grid limit exceeded");
            dim3 grid((unsigned)rt::cdiv((a.Wx + 1) / 2, 256), (unsigned)((a.Hx + 1) / 2), (unsigned)gz);
            if (sc.small3d == 2 && sc.x_f16) {          // half2 mode: fp16 activations in, fp32 (binding) or fp16 out
                if (sc.y_f16) {
                    if (a.C == 1) hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<1, false, _Float16, _Float16>), grid, dim3(256), 0, S(s), a);
                    else hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<2, false, _Float16, _Float16>), grid, dim3(256), 0, S(s), a);
                } else {
                    if (a.C == 1) hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<1, false, _Float16, float>), grid, dim3(256), 0, S(s), a);
                    else hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<2, false, _Float16, float>), grid, dim3(256), 0, S(s), a);
                }
            } else if (sc.small3d == 1 && sc.x_f16 && sc.x_il8) {     // ... with a channel-interleaved (K/8, D, H, W, 8) input: on the matrix cores
                a.w = static_cast<const float*>(sc.small_il_dev);
                const int groups = (int)rt::cdiv((a.Wx + 1) / 2, 16);
                RT_REQUIRE(batch <= 65535, "rt_conv_enqueue: grid limit exceeded");
                const int rowpairs = (int)rt::cdiv((a.Hx + 1) / 2, 2);
                if (plan->softarg) {
                    // the walk is the soft-argmax's reduction axis: one segment, y = the (batch, 1, Hx, Wx) map
                    RT_REQUIRE(a.K == 32 && a.C == 1 && !a.resid && rowpairs <= 65535, "rt_conv_enqueue: plan lost the form its fused soft-argmax needs");
                    a.y_bstride = (int64_t)a.Hx * a.Wx;
                    dim3 gw((unsigned)rt::cdiv(groups, 4), (unsigned)rowpairs, (unsigned)batch);
                    if (plan->softarg == 2) hipLaunchKernelGGL(rt::deconv3d_s2_ilw_kernel<2>, gw, dim3(256), 0, S(s), a, a.Mz, 1);
                    else hipLaunchKernelGGL(rt::deconv3d_s2_ilw_kernel<1>, gw, dim3(256), 0, S(s), a, a.Mz, 1);
                } else if (a.K == 32 && plan->opt_small_walk != 0) {
                    // the depth walk (deconv3d_s2_ilw_kernel): segments long enough to pay for their prologue (>= 8 depth blocks), as many as
                    // it takes to give every SIMD about two waves
                    const int64_t waves = (int64_t)groups * rowpairs * batch;
                    int nseg = (int)std::min<int64_t>(std::max<int64_t>(1, rt::cdiv((int64_t)8 * device_cus(), waves)), std::max(1, a.Mz / 8));
                    const int seg_len = (int)rt::cdiv(a.Mz, nseg);
                    nseg = (int)rt::cdiv(a.Mz, seg_len);
                    RT_REQUIRE(rowpairs <= 65535, "rt_conv_enqueue: grid limit exceeded");
                    dim3 gw((unsigned)(rt::cdiv(groups, 4) * nseg), (unsigned)rowpairs, (unsigned)batch);
                    hipLaunchKernelGGL(rt::deconv3d_s2_ilw_kernel<0>, gw, dim3(256), 0, S(s), a, seg_len, nseg);
                } else {
                    dim3 g2((unsigned)(rt::cdiv(groups, 4 * rt::kSmallIlIters) * a.Mz), (unsigned)rowpairs, (unsigned)batch);
                    hipLaunchKernelGGL(rt::deconv3d_s2_il_kernel, g2, dim3(256), 0, S(s), a);
                }
            } else if (sc.small3d == 1 && !sc.x_f16 && sc.x_il8) {    // fp32 engines: (K/4, D, H, W, 4) in, split form on the matrix cores
                RT_REQUIRE(sc.small_il_dev && sc.small_il_f32, "rt_conv_enqueue: interleaved fp32 input without its weight operands (rt_conv_plan_set_layouts)");
                a.w = static_cast<const float*>(sc.small_il_dev);
                const int groups = (int)rt::cdiv((a.Wx + 1) / 2, 16);
                RT_REQUIRE(batch <= 65535, "rt_conv_enqueue: grid limit exceeded");
                dim3 g2((unsigned)(rt::cdiv(groups, 4 * rt::kSmallIlIters) * a.Mz), (unsigned)rt::cdiv((a.Hx + 1) / 2, 2), (unsigned)batch);
                hipLaunchKernelGGL(rt::deconv3d_s2_il4_kernel, g2, dim3(256), 0, S(s), a);
            } else if (sc.small3d == 1 && sc.x_f16) {   // 3-D last layer in half2 mode: fp16 (K,D,H,W) in, fp32 volume out
                if (a.C == 1) hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<1, true, _Float16, float>), grid, dim3(256), 0, S(s), a);
                else hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<2, true, _Float16, float>), grid, dim3(256), 0, S(s), a);
            } else if (sc.small3d == 2) {
                if (a.C == 1) hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<1, false>), grid, dim3(256), 0, S(s), a);
                else hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<2, false>), grid, dim3(256), 0, S(s), a);
            } else {
                if (a.C == 1) hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<1, true>), grid, dim3(256), 0, S(s), a);
                else hipLaunchKernelGGL((rt::deconv3d_s2_small_kernel<2, true>), grid, dim3(256), 0, S(s), a);
            }
            RT_LAUNCH_CHECK("deconv3d_s2_small_kernel");
            continue;
        }
        rt::ConvArgs a;
        a.z_inner = 0; a.nb_inner = 0;
        a.x = static_cast<const float*>(x);
        a.x2 = static_cast<const float*>(tl_twin_x2); a.x2_from = tl_twin_from;      // rt_conv_enqueue_twin_input (first layers only)
        a.y = static_cast<float*>(y);
        a.w = sc.w_dev;
        a.bias = plan->bias_dev;
        a.zeros = plan->zeros_dev;
        a.dbg = nullptr;
#ifdef RT_KERNEL_TIMING
        if (const char* e = getenv("RT_DBG_PTR")) a.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
        a.resid = plan->has_resid ? static_cast<const float*>(residual) : nullptr;
        a.ch_off = sc.choff_dev;
        a.ch_shift = sc.shift_dev;
        a.w_exact = plan->w_f16;
        a.zs = sc.zs_dev;
        a.CinPad = sc.CinPad; a.Cout = sc.Cout;
        a.Hi = sc.Hi; a.Wi = sc.Wi; a.Ho = sc.Ho; a.Wo = sc.Wo;
        a.x_pitch = sc.x_pitch ? sc.x_pitch : sc.Wi;
        a.pad_y = sc.pad_y; a.pad_x = sc.pad_x; a.nz = sc.nz;
        a.tiles_x = (int)rt::cdiv(sc.Wo, 32 * sc.TXW);
        a.act = plan->act;
        a.xcd_order = plan->opt_xcd;
        a.x_bstride = plan->x_bstride; a.y_bstride = plan->y_bstride;
        a.y_cstride = sc.y_cstride; a.y_zstride = sc.y_zstride; a.y_off = sc.y_off;
        // Conv3D writing a channel-major interleaved tensor (K/8, D, H, W, 8): a depth slice is 8 * H * W elements apart
        if (plan->is_conv3d && !plan->c3d_dchw && sc.y_il8) a.y_zstride = 8 * sc.y_zstride;
        a.y_ystride = sc.y_ystride; a.y_xstride = sc.y_xstride;
        a.r_cstride = sc.r_cstride ? sc.r_cstride : sc.y_cstride;
        a.r_bstride = plan->r_bstride ? plan->r_bstride : plan->y_bstride;
        a.r_il8 = sc.r_il8;
        a.batch = batch; a.cin_real = sc.cin_real; a.x_cstride = (int64_t)sc.Hi * a.x_pitch;
        if (sc.dp4 && sc.split3) {     // fp32 tensors: four phases per workgroup in split-fp16 form (deconv_s3p.hip.h), planar input and output
            a.tiles_x = (int)rt::cdiv(sc.Wi, 32);
            const int tiles = a.tiles_x * (int)rt::cdiv(sc.Hi, 4), nblk = (int)rt::cdiv(sc.Cout, 32);
            RT_REQUIRE(!sc.x_il8 && !sc.x_f16 && !sc.y_f16, "rt_conv_enqueue: the fp32 four-phase transposed kernel takes a planar fp32 input and writes fp32");
            RT_REQUIRE((int64_t)batch * sc.nz <= 65535, "rt_conv_enqueue: grid limit exceeded");
            RT_REQUIRE(sc.y_cstride * (int64_t)sc.Cout < (1ll << 29), "rt_conv_enqueue: output sample exceeds 2 GB (32-bit buffer offsets)");
            dim3 g((unsigned)tiles, (unsigned)nblk, (unsigned)(batch * sc.nz));
            if (plan->opt_trace) fprintf(stderr, "[rt] deconv_s3p grid %u x %u x %u r%d\n", g.x, g.y, g.z, sc.r_il8);
            if (sc.y_il8) hipLaunchKernelGGL(rt::deconv_s3p_kernel<true>, g, dim3(256), 0, S(s), a);
            else hipLaunchKernelGGL(rt::deconv_s3p_kernel<false>, g, dim3(256), 0, S(s), a);
            RT_LAUNCH_CHECK("deconv_s3p_kernel");
            continue;
        }
        if (sc.dp4) {          // transposed 3-D convolution, four phases per workgroup (deconv_f16p.hip.h): tiles of the INPUT grid, one z per output depth
            a.tiles_x = (int)rt::cdiv(sc.Wi, 32);
            const int tiles = a.tiles_x * (int)rt::cdiv(sc.Hi, 4), nblk = (int)rt::cdiv(sc.Cout, 32);
            RT_REQUIRE(sc.x_il8 && sc.y_il8 && sc.x_f16 && sc.y_f16, "rt_conv_enqueue: the four-phase transposed kernel takes interleaved fp16 tensors");
            a.z_inner = plan->opt_zinner != 0 ? 1 : 0;
            RT_REQUIRE(batch <= 65535 && (int64_t)batch * sc.nz <= 65535, "rt_conv_enqueue: grid limit exceeded");
            dim3 g = a.z_inner ? dim3((unsigned)(tiles * sc.nz), (unsigned)nblk, (unsigned)batch) : dim3((unsigned)tiles, (unsigned)nblk, (unsigned)(batch * sc.nz));
            if (plan->opt_trace) fprintf(stderr, "[rt] deconv_f16p grid %u x %u x %u\n", g.x, g.y, g.z);
            if (plan->opt_f16p_walk != 0 && (!plan->has_resid || sc.r_il8)) {       // (a planar skip tensor -- not what the executor uses -- stays with the per-depth kernel)
                // the walk (deconv_f16pw_kernel): a workgroup keeps its tile and takes a segment of the class's output depths.  Segments as
                // long as the grid allows: cost = rounds over the chip's 3 workgroups per CU x (slices per segment + one slice's worth of
                // fill and drain)
                // Both depth classes in ONE walk (slice m of the even class, then slice m of the odd one: their input slices are the same and
                // the next): the second sub-plan rides along when it is the same launch in everything but weights, slices and gather table
                rt::DeconvWalkB wb{};
                const SubConv* sb = (&sc == &plan->subs[0] && plan->subs.size() == 2) ? &plan->subs[1] : nullptr;
                if (sb && plan->opt_f16p_classes != 0 && sb->dp4 && sb->f16mma && sb->x_il8 && sb->y_il8 && sb->x_f16 && sb->y_f16 && sb->r_il8 == sc.r_il8 &&
                    sb->Hi == sc.Hi && sb->Wi == sc.Wi && sb->Ho == sc.Ho && sb->Wo == sc.Wo && sb->Cout == sc.Cout && sb->x_pitch == sc.x_pitch &&
                    sb->y_cstride == sc.y_cstride && sb->y_ystride == sc.y_ystride && sb->r_cstride == sc.r_cstride) {
                    wb.w = static_cast<const float*>(sb->w_dev); wb.zs = sb->zs_dev; wb.ch_off = sb->choff_dev; wb.CinPad = sb->CinPad; wb.nz = sb->nz;
                    skip_sub = true;
                }
                const int nzw = std::max(sc.nz, wb.zs ? wb.nz : 0);                             // slice indices of the walk
                const double per_m = wb.zs ? (double)(sc.CinPad + wb.CinPad) / sc.CinPad : 1.0; // work per slice index, in slices of this class
                const int64_t slots = (int64_t)3 * device_cus(), base = (int64_t)tiles * nblk * batch;
                // (among segmentations within 5 % of the cheapest, the one with the most segments: rounds of long walks do not stay in step and
                //  their tail is a whole walk long -- NVSmall deconv3D_2 at batch 8: 1 segment 1.31 ms, 4 segments 1.26, both "cost" 146-152)
                int nseg = 1;
                double best = 1e30;
                auto seg_cost = [&](int ns) {
                    const int seg = (int)rt::cdiv(nzw, ns);
                    return (int)rt::cdiv(nzw, seg) != ns ? 1e30 : (double)rt::cdiv(base * ns, slots) * (seg * per_m + 1.0);
                };
                for (int ns = 1; ns <= nzw; ns++) best = std::min(best, seg_cost(ns));
                for (int ns = 1; ns <= nzw; ns++)
                    if (seg_cost(ns) <= 1.05 * best) nseg = ns;
                if (plan->opt_f16p_walk > 0) nseg = std::min(plan->opt_f16p_walk, nzw);            // (RT_F16P_WALK=<n>: n segments)
                a.dw_seg = (int)rt::cdiv(nzw, nseg); a.dw_nseg = nzw; a.dw_cpc = sc.nz; a.nz = (int)rt::cdiv(nzw, a.dw_seg);
                RT_REQUIRE((int64_t)batch * a.nz <= 65535, "rt_conv_enqueue: grid limit exceeded");
                dim3 gw = a.z_inner ? dim3((unsigned)(tiles * a.nz), (unsigned)nblk, (unsigned)batch) : dim3((unsigned)tiles, (unsigned)nblk, (unsigned)(batch * a.nz));
                if (plan->opt_trace) fprintf(stderr, "[rt] deconv_f16pw grid %u x %u x %u, %d slice indices per segment, %d class(es)\n", gw.x, gw.y, gw.z, a.dw_seg, wb.zs ? 2 : 1);
                hipLaunchKernelGGL(rt::deconv_f16pw_kernel, gw, dim3(256), 0, S(s), a, wb);
                RT_LAUNCH_CHECK("deconv_f16pw_kernel");
                continue;
            }
            hipLaunchKernelGGL(rt::deconv_f16p_kernel, g, dim3(256), 0, S(s), a);
            RT_LAUNCH_CHECK("deconv_f16p_kernel");
            continue;
        }
        const int tiles_y = (int)rt::cdiv(sc.Ho, sc.TY);
        const int64_t gz = (int64_t)batch * sc.nz;
        // split-fp16 / fp16-operand launches of 3-D plans fold the z-slices into grid.x, z fastest (conv_mfma.hip.h: ConvArgs::z_inner)
        // (fp16-operand launches on interleaved tensors only: measured on ResNet-18 3D fp32 the planar split kernel is 3 % SLOWER with z fastest,
        //  8.56 vs 8.27 ms per pair -- RT_Z_INNER=2 folds those too)
        const bool zin = sc.nz > 1 && plan->opt_zinner != 0 && (sc.f16mma || (sc.split3 && plan->opt_zinner == 2)) &&
                         (int64_t)rt::cdiv(sc.Wo, 32 * sc.TXW) * rt::cdiv(sc.Ho, sc.TY) * sc.nz < (1ll << 30);
        a.z_inner = zin ? 1 : 0;
        // ... and 3-D launches with several blocks of 32 output channels fold those in as well, fastest of all (ConvArgs::nb_inner)
        int nb_fold = 0;
        auto zfold = [&](dim3 g) {
            if (zin) g = dim3(g.x * (unsigned)sc.nz, g.y, (unsigned)batch);
            return nb_fold ? dim3(g.x * g.y, 1u, g.z) : g;
        };
        RT_REQUIRE(zin ? batch <= 65535 : gz <= 65535, "rt_conv_enqueue: batch * depth = %lld exceeds the grid limit", (long long)gz);
        RT_REQUIRE(sc.y_cstride * (int64_t)sc.Cout < (1ll << 29), "rt_conv_enqueue: output sample exceeds 2 GB (32-bit buffer offsets)");
        if (sc.direct) {
            RT_REQUIRE(sc.Ho <= 65535, "rt_conv_enqueue: output too tall for the direct kernel");
            dim3 dgrid((unsigned)rt::cdiv(sc.Wo, 256), (unsigned)sc.Ho, (unsigned)gz);
            bool launched = false;
#define RT_DIRECT(co, kh, kw)                                                                                         \
    if (!launched && sc.Cout <= co && sc.KH == kh && sc.KW == kw) {                                                   \
        hipLaunchKernelGGL((rt::conv_direct_f32_kernel<co, kh, kw>), dgrid, dim3(256), 0, S(s), a, sc.S, sc.cin_real); \
        launched = true;                                                                                              \
    }
            RT_DIRECT(1, 1, 1) RT_DIRECT(1, 1, 2) RT_DIRECT(1, 2, 1) RT_DIRECT(1, 2, 2) RT_DIRECT(1, 3, 3) RT_DIRECT(1, 5, 5)
            RT_DIRECT(2, 1, 1) RT_DIRECT(2, 1, 2) RT_DIRECT(2, 2, 1) RT_DIRECT(2, 2, 2) RT_DIRECT(2, 3, 3) RT_DIRECT(2, 5, 5)
#undef RT_DIRECT
            if (!launched) return fail(RT_E_UNSUPPORTED, "conv (direct): window %dx%d not instantiated", sc.KH, sc.KW);
            RT_LAUNCH_CHECK("conv_direct_f32_kernel");
            continue;
        }
#ifdef RT_EXPERIMENTAL
        if (sc.s3p) {
            // persistent: one 8-wave workgroup per CU walks a contiguous range of tiles (RT_S3P_GRID: test knob)
            const int64_t T = (int64_t)a.tiles_x * tiles_y * batch;
            int64_t g = std::min<int64_t>(T, (plan->opt_s3p_grid > 0 ? plan->opt_s3p_grid : device_cus()));
            if (g >= 8) g -= g % 8;
            dim3 pgrid((unsigned)std::max<int64_t>(g, 1));
            if (plan->opt_trace)
                fprintf(stderr, "[rt] conv_s3p x%d y%d r%d tiles %lld grid %u\n", sc.x_il8, sc.y_il8, sc.r_il8, (long long)T, pgrid.x);
            if (sc.x_il8 && sc.y_il8) hipLaunchKernelGGL((rt::conv_s3p_kernel<8, true, true>), pgrid, dim3(512), 0, S(s), a);
            else if (sc.x_il8) hipLaunchKernelGGL((rt::conv_s3p_kernel<8, true, false>), pgrid, dim3(512), 0, S(s), a);
            else if (sc.y_il8) hipLaunchKernelGGL((rt::conv_s3p_kernel<8, false, true>), pgrid, dim3(512), 0, S(s), a);
            else hipLaunchKernelGGL((rt::conv_s3p_kernel<8, false, false>), pgrid, dim3(512), 0, S(s), a);
            RT_LAUNCH_CHECK("conv_s3p_kernel");
            continue;
        }
#endif
        dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)rt::cdiv(sc.Cout, 32 * sc.NBW), (unsigned)gz);
        if (!plan->is2d && grid.y > 1 && plan->opt_nbinner != 0 && (sc.f16mma || sc.split3) && !sc.rb && sc.NBW == 1 &&
            (int64_t)grid.x * grid.y * (zin ? sc.nz : 1) < (1ll << 30))
            nb_fold = (int)grid.y;
        a.nb_inner = nb_fold;
        if (sc.rb) {
            rt::RBArgs ra;
            ra.c = a;
            ra.c.resid = static_cast<const float*>(x);          // the block's input is its skip connection
            ra.c.r_cstride = a.x_cstride; ra.c.r_bstride = plan->x_bstride; ra.c.r_il8 = sc.x_il8;
            ra.w1 = plan->rb_w1_dev; ra.bias1 = plan->rb_bias1_dev; ra.act1 = plan->rb_act1; ra.cmid = plan->rb_cmid; ra.seg = 0;
            // 32 -> 32 -> 32 channels, ELU after both, on interleaved tensors (the feature towers): the streaming form, strips of 30 columns x
            // segments of 16 rows (conv_rbs.hip.h); everything else: one 4 x 32 tile per workgroup
            if (sc.x_il8 && sc.y_il8 && plan->cin == 32 && plan->rb_cmid == 32 && sc.Cout == 32 && plan->rb_act1 == 1 && plan->act == 1 &&
                plan->opt_rb_tiles == 0) {
                ra.c.tiles_x = (int)rt::cdiv(sc.Wo, rt::S3RBSCfg::SW);
                // rows per workgroup: the pipeline's fill and drain steps and the prologue are paid per segment, so longer segments
                // cost fewer CU-cycles per row; shorter ones fill more CUs of an otherwise idle GPU (RT_RBS_SEG, default: see DESIGN.md 4.4)
                // Measured in the running network (ResNet-18 2D, 1257x369, four contexts): 16 rows 2129, 24: 2178, 32: 2185, 48: 2067,
                // 64: 1957 pairs/s (two launches of the layer-by-layer kernels: 2057); alone 16 rows are fastest (24.5 vs 32.4 us).
                // Round 3, siamese batch-2 launches, six one-stream contexts: 32 rows 2315, 48: 2432, 64: 2457-2539, 96: 2441, 128: 2305 pairs/s;
                // one context, synchronous execute(): 32 rows 0.60 ms per pair, 64: 0.74-0.76 ms.  So: throughput hint -> 64 rows while
                // that leaves >= 120 workgroups; otherwise 32 rows while that leaves >= 120; otherwise 16.
                int seg = plan->opt_rbs_seg;
                const auto wgs = [&](int rows) { return ra.c.tiles_x * (int)rt::cdiv(sc.Ho, rows) * batch; };
                if (seg <= 0) seg = ((hints & RT_HINT_THROUGHPUT) && wgs(64) >= 120) ? 64 : (wgs(32) >= 120 ? 32 : rt::S3RBSCfg::SEG);
                seg = seg < 4 ? 4 : (seg > 240 ? 240 : (seg + 3) / 4 * 4);
                ra.seg = seg;
                dim3 sgrid((unsigned)(ra.c.tiles_x * (int)rt::cdiv(sc.Ho, seg)), 1u, (unsigned)batch);
                if (plan->opt_trace) fprintf(stderr, "[rt] conv_s3rb%c grid %u x %u split %d -> %d\n", plan->x_split ? 'd' : 's', sgrid.x, sgrid.z, plan->x_split, plan->y_split);
                if (sc.x_f16) {                         // half2 mode: fp16 tensors, fp16 operands
                    ra.w1 = plan->rbh_w1_dev; ra.c.w = plan->rbh_w2_dev;
                    hipLaunchKernelGGL(rt::conv_f16rbd_kernel, sgrid, dim3(512), 0, S(s), ra);
                    RT_LAUNCH_CHECK("conv_f16rbd_kernel");
                    continue;
                }
                if (plan->x_split) {                    // pre-split input: the DMA-fed block (its own row order of the weight slabs)
                    ra.w1 = plan->rbd_w1_dev; ra.c.w = plan->rbd_w2_dev;
                    if (plan->y_split) hipLaunchKernelGGL(rt::conv_s3rbd_kernel<true>, sgrid, dim3(512), 0, S(s), ra);
                    else hipLaunchKernelGGL(rt::conv_s3rbd_kernel<false>, sgrid, dim3(512), 0, S(s), ra);
                    RT_LAUNCH_CHECK("conv_s3rbd_kernel");
                    continue;
                }
                if (plan->y_split) hipLaunchKernelGGL(rt::conv_s3rbs_kernel<true>, sgrid, dim3(512), 0, S(s), ra);
                else hipLaunchKernelGGL(rt::conv_s3rbs_kernel<false>, sgrid, dim3(512), 0, S(s), ra);      // ELU / ELU, as in every tower block
                RT_LAUNCH_CHECK("conv_s3rbs_kernel");
                continue;
            }
            if (sc.x_f16 || sc.y_f16)
                return fail(RT_E_UNSUPPORTED, "rt_conv_enqueue: the half2 form of the residual block takes channel-interleaved fp16 tensors only (rt_conv_plan_set_layouts(1, 1, 1))");
#ifdef RT_EXPERIMENTAL
            dim3 rgrid((unsigned)(a.tiles_x * tiles_y), 1u, (unsigned)batch);
            if (plan->opt_trace) fprintf(stderr, "[rt] conv_s3rb x%d y%d grid %u x %u\n", sc.x_il8, sc.y_il8, rgrid.x, rgrid.z);
            if (sc.x_il8 && sc.y_il8) hipLaunchKernelGGL((rt::conv_s3rb_kernel<true, true>), rgrid, dim3(256), 0, S(s), ra);
            else if (sc.x_il8) hipLaunchKernelGGL((rt::conv_s3rb_kernel<true, false>), rgrid, dim3(256), 0, S(s), ra);
            else if (sc.y_il8) hipLaunchKernelGGL((rt::conv_s3rb_kernel<false, true>), rgrid, dim3(256), 0, S(s), ra);
            else hipLaunchKernelGGL((rt::conv_s3rb_kernel<false, false>), rgrid, dim3(256), 0, S(s), ra);
            RT_LAUNCH_CHECK("conv_s3rb_kernel");
            continue;
#else
            return fail(RT_E_UNSUPPORTED, "rt_conv_enqueue: this build plans residual blocks for the streaming kernel only (32 -> 32 -> 32 channels, "
                                           "ELU / ELU, interleaved tensors); the per-tile form is compiled with RT_EXPERIMENTAL");
#endif
        }
        if (sc.s3first) {
            if (sc.y_il8) hipLaunchKernelGGL((rt::conv_s3_first_kernel<true>), grid, dim3(256), 0, S(s), a);
            else hipLaunchKernelGGL((rt::conv_s3_first_kernel<false>), grid, dim3(256), 0, S(s), a);
            RT_LAUNCH_CHECK("conv_s3_first_kernel");
            continue;
        }
        if (sc.split3) {
            if (plan->opt_trace)
                fprintf(stderr, "[rt] conv_s3 %dx%d s%d x%d y%d r%d grid %u x %u x %u\n", sc.KH, sc.KW, sc.S, sc.x_il8, sc.y_il8, sc.r_il8, grid.x, grid.y, grid.z);
            bool launched = false;
            // split-K for launches that leave SIMDs with a single wave (the low-resolution layers; conv_split.hip.h): KS groups of
            // 4 waves per workgroup, as many as the CU's 16 wave slots at this kernel's register budget allow.  Chosen from the
            // per-sample grid so that a sample's result does not depend on the batch it travels in.
            int ks = 1;
            const int64_t per_cu = rt::cdiv((int64_t)grid.x * grid.y * (gz / batch), (int64_t)device_cus());
            // Not with the throughput hint: there other contexts' launches fill the SIMDs, and a 16-wave workgroup with 139 KB of LDS
            // keeps them off its CU (six one-stream contexts: 2502 against 2589 pairs/s; one context, synchronous: 0.541 against 0.557 ms).
            if (plan->opt_ksplit > 0 || (plan->opt_ksplit < 0 && !(hints & RT_HINT_THROUGHPUT))) {
                const int nch = a.CinPad / 16;
                ks = plan->opt_ksplit > 0 ? plan->opt_ksplit : (per_cu == 1 && nch >= 4 ? 4 : (per_cu <= 2 && nch >= 2 ? 2 : 1));
                if (plan->opt_trace) fprintf(stderr, "[rt] conv_s3 split-K request %d\n", ks);
            }
            if (sc.x_f16 || sc.y_f16) {                 // fp16 storage (3-D tensors of half2 mode): planar, 4-row tiles
                // (the residual's layout is a run-time flag of the kernel: r_il8; an interleaved fp16 OUTPUT exists for fp32 planar input)
                if (!sc.x_f16 && sc.x_il8 && sc.KH == 3 && sc.KW == 3 && sc.S == 1 && sc.TY == 4 && sc.y_f16) {      // fp32 interleaved in (2-D tower tensor)
                    if (sc.y_il8) launch_s3<3, 3, 1, true, true, 4, float, _Float16>(zfold(grid), ks, per_cu, S(s), a);
                    else launch_s3<3, 3, 1, true, false, 4, float, _Float16>(zfold(grid), ks, per_cu, S(s), a);
                    RT_LAUNCH_CHECK("conv_s3_kernel<float, f16>");
                    continue;
                }
                RT_REQUIRE(!sc.x_il8 && !(sc.y_il8 && sc.x_f16) && sc.TY == 4 && sc.y_f16, "rt_conv_enqueue: fp16-storage variant of the split kernel not instantiated");
#define RT_S3H(kh, kw, st)                                                                                                         \
    if (!launched && sc.KH == kh && sc.KW == kw && sc.S == st) {                                                                   \
        if (sc.x_f16) launch_s3<kh, kw, st, false, false, 4, _Float16, _Float16>(zfold(grid), ks, per_cu, S(s), a); \
        else if (sc.y_il8) launch_s3<kh, kw, st, false, true, 4, float, _Float16>(zfold(grid), ks, per_cu, S(s), a); \
        else launch_s3<kh, kw, st, false, false, 4, float, _Float16>(zfold(grid), ks, per_cu, S(s), a);   \
        launched = true;                                                                                                           \
    }
                RT_S3H(3, 3, 1) RT_S3H(3, 3, 2) RT_S3H(1, 1, 1) RT_S3H(1, 2, 1) RT_S3H(2, 1, 1) RT_S3H(2, 2, 1)
#undef RT_S3H
            }
#ifdef RT_EXPERIMENTAL
            if (!launched && sc.KH == 3 && sc.KW == 3 && sc.S == 1 && sc.TY == 8) {          // 8-row tiles, 8 waves
                if (sc.x_il8 && sc.y_il8) launch_s3<3, 3, 1, true, true, 8>(zfold(grid), ks, per_cu, S(s), a);
                else if (sc.x_il8) launch_s3<3, 3, 1, true, false, 8>(zfold(grid), ks, per_cu, S(s), a);
                else if (sc.y_il8) launch_s3<3, 3, 1, false, true, 8>(zfold(grid), ks, per_cu, S(s), a);
                else launch_s3<3, 3, 1, false, false, 8>(zfold(grid), ks, per_cu, S(s), a);
                launched = true;
            }
#endif
#define RT_S3(kh, kw, st)                                                                                                   \
    if (!launched && sc.KH == kh && sc.KW == kw && sc.S == st) {                                                            \
        if (sc.x_il8 && sc.y_il8) launch_s3<kh, kw, st, true, true>(zfold(grid), ks, per_cu, S(s), a);        \
        else if (sc.x_il8) launch_s3<kh, kw, st, true, false>(zfold(grid), ks, per_cu, S(s), a);              \
        else if (sc.y_il8) launch_s3<kh, kw, st, false, true>(zfold(grid), ks, per_cu, S(s), a);              \
        else launch_s3<kh, kw, st, false, false>(zfold(grid), ks, per_cu, S(s), a);                           \
        launched = true;                                                                                                    \
    }
            RT_S3(3, 3, 1) RT_S3(3, 3, 2) RT_S3(1, 1, 1) RT_S3(1, 2, 1) RT_S3(2, 1, 1) RT_S3(2, 2, 1)
#undef RT_S3
            if (!launched) return fail(RT_E_UNSUPPORTED, "conv (split fp16): window %dx%d stride %d not instantiated", sc.KH, sc.KW, sc.S);
            RT_LAUNCH_CHECK("conv_s3_kernel");
            continue;
        }
        if (sc.f16first) {
            if (plan->opt_trace) fprintf(stderr, "[rt] conv_f16_first il8 y%d grid %u x %u x %u\n", sc.y_il8, grid.x, grid.y, grid.z);
            if (sc.y_il8) hipLaunchKernelGGL((rt::conv_f16_first_kernel<true>), grid, dim3(256), 0, S(s), a, plan->cin);
            else hipLaunchKernelGGL((rt::conv_f16_first_kernel<false>), grid, dim3(256), 0, S(s), a, plan->cin);
            RT_LAUNCH_CHECK("conv_f16_first_kernel");
            continue;
        }
        if (sc.f16mma) {
            if (plan->opt_trace)
                fprintf(stderr, "[rt] conv_f16mma %dx%d s%d rows %d il8 x%d y%d r%d grid %u x %u x %u\n", sc.KH, sc.KW, sc.S, sc.TY, sc.x_il8,
                        sc.y_il8, sc.r_il8, grid.x, grid.y, grid.z);
            // Conv3D 3x3x3 stride 1 between interleaved fp16 tensors: the workgroup walks down the depth axis (conv_f16dw.hip.h)
            if (plan->c3d_dw && plan->opt_dw != 0 && sc.KH == 3 && sc.KW == 3 && sc.S == 1 && sc.x_il8 && sc.y_il8 && !sc.zs_dev && sc.y_xstride == 1 &&
                (!plan->has_resid || sc.r_il8) && (plan->act == RT_ACT_NONE || plan->act == RT_ACT_ELU) && !sc.shift_dev) {
                using Dw = rt::ConvF16DwCfg;
                const int dw_tiles_x = (int)rt::cdiv(sc.Wo, Dw::TX), dw_nkb = (int)rt::cdiv(sc.Cout, 32);
                const int64_t npairs = rt::cdiv(dw_tiles_x * (int)rt::cdiv(sc.Ho, Dw::TY), 2);
                // depth segments: every segment pays two steps' worth of MFMAs for its neighbours' slices and a prologue, every round of
                // workgroups over the CUs costs a whole segment -- minimise rounds x (segment + overhead); results do not depend on it.
                // MI355X, NVSmall half2 (profiles/r05_conv3d_layers.txt): batch 8 -- one segment everywhere (conv3D_2 0.206 ms per pair, two
                // segments 0.215, four 0.228); batch 1 -- conv3D_2 two segments (0.232; one: 0.340), conv3D_4 four (0.137; one: 0.298).
                int nseg = plan->opt_dw_nseg;
                double best = 1e30, fill = 1.0;
                for (int ns = 1; ns <= sc.nz; ns++) {
                    const int seg = (int)rt::cdiv(sc.nz, ns);
                    if (ns > 1 && seg < 4) break;
                    if ((int)rt::cdiv(sc.nz, seg) != ns || (plan->opt_dw_nseg > 0 && ns != std::min(plan->opt_dw_nseg, sc.nz))) continue;
                    const int64_t wgs = npairs * ns * dw_nkb * batch, rounds = rt::cdiv(wgs, (int64_t)device_cus());
                    const double steps = seg + (ns > 1 ? 3.0 : 1.7), cost = (double)rounds * steps;
                    if (cost < best - 1e-9) { best = cost; nseg = ns; fill = (double)wgs / (double)(rounds * device_cus()) * seg / steps; }
                }
                // a launch that would leave most of the chip idle or spend its time on segment ends stays with the per-slice kernel (the
                // 128-channel layers at 12 x 41 x 129: 0.155 against 0.098 ms at batch 1, a tie at batch 8)
                if (plan->opt_dw < 0 && fill < 0.6) goto dw_declined;
                nseg = std::max(1, std::min(nseg, sc.nz));
                a.tiles_x = dw_tiles_x;
                a.dw_ntiles = a.tiles_x * (int)rt::cdiv(sc.Ho, Dw::TY);
                a.dw_cpc = plan->c3d_C / 16;
                a.nb_inner = dw_nkb;
                a.dw_seg = (int)rt::cdiv(sc.nz, nseg);
                a.dw_nseg = (int)rt::cdiv(sc.nz, a.dw_seg);
                const int64_t gx = npairs * a.dw_nseg * a.nb_inner;
                RT_REQUIRE(gx < (1ll << 31) && batch <= 65535, "rt_conv_enqueue: grid limit exceeded");
                dim3 gd((unsigned)gx, 1u, (unsigned)batch);
                if (plan->opt_trace) fprintf(stderr, "[rt] conv_f16dw grid %u x %u segments %d x %d slices, %d chunks per slice\n", gd.x, gd.z, a.dw_nseg, a.dw_seg, a.dw_cpc);
                const bool resident = a.dw_cpc <= 2, elu = plan->act == RT_ACT_ELU;
#define RT_DW_LAUNCH(res, hr, el) hipLaunchKernelGGL((rt::conv_f16dw_kernel<res, hr, el>), gd, dim3(512), 0, S(s), a)
                if (resident) {
                    if (plan->has_resid) { if (elu) RT_DW_LAUNCH(true, true, true); else RT_DW_LAUNCH(true, true, false); }
                    else { if (elu) RT_DW_LAUNCH(true, false, true); else RT_DW_LAUNCH(true, false, false); }
                } else {
                    if (plan->has_resid) { if (elu) RT_DW_LAUNCH(false, true, true); else RT_DW_LAUNCH(false, true, false); }
                    else { if (elu) RT_DW_LAUNCH(false, false, true); else RT_DW_LAUNCH(false, false, false); }
                }
#undef RT_DW_LAUNCH
                RT_LAUNCH_CHECK("conv_f16dw_kernel");
                continue;
            }
            dw_declined:
            // Conv3D between interleaved fp16 tensors: four output rows per wave, operands reused from registers (conv_f16r4.hip.h) -- the
            // 4 x 32-tile kernel below reads 2 KB of LDS per MFMA and is LDS-bound at 0.3 of the matrix peak on these layers
            if (sc.KH == 3 && sc.KW == 3 && sc.S == 1 && sc.x_il8 && sc.y_il8 && !sc.zs_dev && sc.TY == 4 && sc.y_xstride == 1 &&
                (plan->opt_r4 > 0 || (plan->opt_r4 < 0 && plan->is_conv3d && sc.Ho >= 12))) {
                dim3 g4 = zfold(dim3((unsigned)(a.tiles_x * (int)rt::cdiv(sc.Ho, rt::ConvF16R4Cfg::TY)), grid.y, grid.z));
                if (plan->opt_trace) fprintf(stderr, "[rt] conv_f16r4 grid %u x %u x %u\n", g4.x, g4.y, g4.z);
                hipLaunchKernelGGL(rt::conv_f16r4_kernel, g4, dim3(256), 0, S(s), a);
                RT_LAUNCH_CHECK("conv_f16r4_kernel");
                continue;
            }
            if (sc.KH == 3 && sc.KW == 3 && sc.S == 1) {     // the tower layers: tensor layouts x rows per workgroup
#define RT_F16_331(xi, yi, nw)                                                                                   \
    if (sc.x_il8 == xi && sc.y_il8 == yi && sc.TY == nw) {                                                       \
        hipLaunchKernelGGL((rt::conv_f16mma_kernel<3, 3, 1, xi != 0, yi != 0, nw>), zfold(grid), dim3(64 * nw), 0, S(s), a); \
        RT_LAUNCH_CHECK("conv_f16mma_kernel<3,3,1>");                                                            \
        continue;                                                                                                \
    }
                RT_F16_331(0, 0, 4) RT_F16_331(1, 0, 4) RT_F16_331(0, 1, 4) RT_F16_331(1, 1, 4)
#undef RT_F16_331
            }
            if (sc.KH == 3 && sc.KW == 3 && sc.S == 2 && sc.TY == 4 && (sc.x_il8 || sc.y_il8)) {      // stride-2 Conv3D on interleaved 4-D tensors
                if (sc.x_il8 && sc.y_il8) hipLaunchKernelGGL((rt::conv_f16mma_kernel<3, 3, 2, true, true>), zfold(grid), dim3(256), 0, S(s), a);
                else if (sc.x_il8) hipLaunchKernelGGL((rt::conv_f16mma_kernel<3, 3, 2, true, false>), zfold(grid), dim3(256), 0, S(s), a);
                else hipLaunchKernelGGL((rt::conv_f16mma_kernel<3, 3, 2, false, true>), zfold(grid), dim3(256), 0, S(s), a);
                RT_LAUNCH_CHECK("conv_f16mma_kernel<3,3,2>");
                continue;
            }
            if (sc.KH == 2 && sc.KW == 2 && sc.S == 1 && sc.TY == 4 && sc.x_il8) {      // phases of a transposed 3-D convolution on interleaved input
                if (sc.y_il8) hipLaunchKernelGGL((rt::conv_f16mma_kernel<2, 2, 1, true, true>), zfold(grid), dim3(256), 0, S(s), a);
                else hipLaunchKernelGGL((rt::conv_f16mma_kernel<2, 2, 1, true, false>), zfold(grid), dim3(256), 0, S(s), a);
                RT_LAUNCH_CHECK("conv_f16mma_kernel<2,2,1,il>");
                continue;
            }
            RT_REQUIRE(!(sc.x_il8 || sc.y_il8 || sc.r_il8) && sc.TY == 4, "rt_conv_enqueue: fp16-arithmetic variant not instantiated");
#define RT_F16CASE(kh, kw, st)                                                                              \
    if (sc.KH == kh && sc.KW == kw && sc.S == st) {                                                         \
        hipLaunchKernelGGL((rt::conv_f16mma_kernel<kh, kw, st>), zfold(grid), dim3(256), 0, S(s), a);              \
        RT_LAUNCH_CHECK("conv_f16mma_kernel<" #kh "," #kw "," #st ">");                                     \
        continue;                                                                                           \
    }
            RT_F16CASE(3, 3, 2) RT_F16CASE(1, 1, 1) RT_F16CASE(1, 2, 1) RT_F16CASE(2, 1, 1) RT_F16CASE(2, 2, 1)
#undef RT_F16CASE
            return fail(RT_E_UNSUPPORTED, "conv (fp16 arithmetic): window %dx%d stride %d not instantiated", sc.KH, sc.KW, sc.S);
        }
        if (sc.wino) {
            RT_REQUIRE(!sc.zs_dev && sc.y_xstride == 1, "rt_conv_enqueue: Winograd kernel takes uniform, x-contiguous slices only");
            if (sc.x_f16 || sc.y_f16) {
                RT_REQUIRE(sc.x_f16 && sc.y_f16, "rt_conv_enqueue: the Winograd kernel takes fp16 on both sides or on neither");
                hipLaunchKernelGGL((rt::conv_wino_f32_kernel<4, _Float16, _Float16>), grid, dim3(256), 0, S(s), a);
            }
#ifdef RT_EXPERIMENTAL
            else if (sc.NW == 8) {
                RT_REQUIRE(!(sc.x_il8 || sc.y_il8 || sc.r_il8), "rt_conv_enqueue: interleaved tensors need the 4-wave Winograd tile");
                hipLaunchKernelGGL((rt::conv_wino_f32_kernel<8>), grid, dim3(512), 0, S(s), a);
            }
#else
            else if (sc.NW != 4) return fail(RT_E_UNSUPPORTED, "rt_conv_enqueue: Winograd kernel: 4-wave tile (8 waves: RT_EXPERIMENTAL builds)");
#endif
            else if (sc.x_il8 && sc.y_il8) hipLaunchKernelGGL((rt::conv_wino_f32_kernel<4, float, float, true, true>), grid, dim3(256), 0, S(s), a);
            else if (sc.x_il8) hipLaunchKernelGGL((rt::conv_wino_f32_kernel<4, float, float, true, false>), grid, dim3(256), 0, S(s), a);
            else if (sc.y_il8) hipLaunchKernelGGL((rt::conv_wino_f32_kernel<4, float, float, false, true>), grid, dim3(256), 0, S(s), a);
            else if (sc.r_il8) hipLaunchKernelGGL((rt::conv_wino_f32_kernel<4, float, float, false, false, true>), grid, dim3(256), 0, S(s), a);
            else hipLaunchKernelGGL((rt::conv_wino_f32_kernel<4>), grid, dim3(256), 0, S(s), a);
            RT_LAUNCH_CHECK("conv_wino_f32_kernel");
            continue;
        }
        if (int rc = launch_sub(sc, a, grid, S(s))) return rc;
    }
    return 0;
}

extern "C" int rt_has_experimental(void) {
#ifdef RT_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}

extern "C" int rt_conv_plan_destroy(rtConvPlan* plan) {
    free_plan(plan);
    return 0;
}

// ---- multi-GPU: RCCL communicator + byte broadcast (include/rt_stereo.h) ------------------------------------------------------
struct rtComm {
    void* nccl = nullptr;      // ncclComm_t
    bool owned = false;
    int world = 1, rank = 0, device = 0;
    void* fabric = nullptr;    // emulator build: the in-process stand-in for RCCL that the communicators of one rt_comm_init_all share
};

#ifndef HIPEMU
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// librccl is loaded on first use: a single-GPU process never maps it
RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
#define RT_RCCL_SYM(n) api.n = reinterpret_cast<decltype(api.n)>(dlsym(api.lib, "nccl" #n))
        RT_RCCL_SYM(GetUniqueId); RT_RCCL_SYM(CommInitRank); RT_RCCL_SYM(CommInitAll); RT_RCCL_SYM(CommDestroy); RT_RCCL_SYM(CommCount);
        RT_RCCL_SYM(CommUserRank); RT_RCCL_SYM(CommCuDevice); RT_RCCL_SYM(Broadcast); RT_RCCL_SYM(GroupStart); RT_RCCL_SYM(GroupEnd);
        RT_RCCL_SYM(GetErrorString);
#undef RT_RCCL_SYM
    });
    const bool ok = api.lib && api.GetUniqueId && api.CommInitRank && api.CommInitAll && api.CommDestroy && api.CommCount && api.CommUserRank &&
                    api.CommCuDevice && api.Broadcast && api.GroupStart && api.GroupEnd && api.GetErrorString;
    return ok ? &api : nullptr;
}
#define RT_RCCL(api, call, what)                                                                          \
    do {                                                                                                  \
        ncclResult_t r_ = (call);                                                                         \
        if (r_ != ncclSuccess) return fail(RT_E_RUNTIME, "%s: RCCL: %s", what, (api)->GetErrorString(r_)); \
    } while (0)
#define RT_NEED_RCCL(api, what)   \
    RcclApi* api = rccl();        \
    if (!api) return fail(RT_E_UNSUPPORTED, "%s: librccl.so could not be loaded (%s)", what, dlerror() ? dlerror() : "symbols missing")
}  // namespace

extern "C" int rt_comm_unique_id(void* id_bytes) {
    RT_REQUIRE(id_bytes, "rt_comm_unique_id: null pointer");
    RT_NEED_RCCL(api, "rt_comm_unique_id");
    ncclUniqueId id;
    RT_RCCL(api, api->GetUniqueId(&id), "rt_comm_unique_id");
    static_assert(sizeof(id) == RT_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id_bytes, &id, sizeof(id));
    return 0;
}

static int fill_comm(RcclApi* api, rtComm* c) {
    RT_RCCL(api, api->CommCount(static_cast<ncclComm_t>(c->nccl), &c->world), "rt_comm");
    RT_RCCL(api, api->CommUserRank(static_cast<ncclComm_t>(c->nccl), &c->rank), "rt_comm");
    RT_RCCL(api, api->CommCuDevice(static_cast<ncclComm_t>(c->nccl), &c->device), "rt_comm");
    return 0;
}

extern "C" int rt_comm_init_rank(rtComm** comm, int world, int rank, const void* id_bytes) {
    RT_REQUIRE(comm && id_bytes && world >= 1 && rank >= 0 && rank < world, "rt_comm_init_rank: bad arguments");
    RT_NEED_RCCL(api, "rt_comm_init_rank");
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    RT_RCCL(api, api->CommInitRank(&c, world, id, rank), "rt_comm_init_rank");
    auto* out = new rtComm();
    out->nccl = c; out->owned = true;
    if (int rc = fill_comm(api, out)) { api->CommDestroy(c); delete out; return rc; }
    *comm = out;
    return 0;
}

extern "C" int rt_comm_init_all(rtComm** comms, int ndev, const int* devices) {
    RT_REQUIRE(comms && ndev >= 1 && ndev <= 64, "rt_comm_init_all: bad arguments");
    RT_NEED_RCCL(api, "rt_comm_init_all");
    std::vector<ncclComm_t> c((size_t)ndev, nullptr);
    RT_RCCL(api, api->CommInitAll(c.data(), ndev, devices), "rt_comm_init_all");
    for (int i = 0; i < ndev; i++) {
        comms[i] = new rtComm();
        comms[i]->nccl = c[i]; comms[i]->owned = true;
        if (int rc = fill_comm(api, comms[i])) return rc;
    }
    return 0;
}

extern "C" int rt_comm_adopt(rtComm** comm, void* nccl_comm) {
    RT_REQUIRE(comm && nccl_comm, "rt_comm_adopt: null pointer");
    RT_NEED_RCCL(api, "rt_comm_adopt");
    auto* out = new rtComm();
    out->nccl = nccl_comm; out->owned = false;
    if (int rc = fill_comm(api, out)) { delete out; return rc; }
    *comm = out;
    return 0;
}

extern "C" int rt_comm_broadcast(rtComm* comm, void* host_buf, size_t bytes, int root, rtStream s) {
    RT_REQUIRE(comm && (host_buf || !bytes) && root >= 0 && root < comm->world, "rt_comm_broadcast: bad arguments");
    if (!bytes) return 0;
    RT_NEED_RCCL(api, "rt_comm_broadcast");
    int prev = 0;
    RT_HIP(hipGetDevice(&prev));
    RT_HIP(hipSetDevice(comm->device));
    void* dev = nullptr;
    RT_HIP(hipMalloc(&dev, bytes));
    int rc = 0;
    if (comm->rank == root && hipMemcpyAsync(dev, host_buf, bytes, hipMemcpyHostToDevice, S(s)) != hipSuccess) rc = fail(RT_E_RUNTIME, "rt_comm_broadcast: upload failed");
    if (!rc) {
        ncclResult_t r = api->Broadcast(dev, dev, bytes, ncclUint8, root, static_cast<ncclComm_t>(comm->nccl), S(s));
        if (r != ncclSuccess) rc = fail(RT_E_RUNTIME, "rt_comm_broadcast: RCCL: %s", api->GetErrorString(r));
    }
    if (!rc && comm->rank != root && hipMemcpyAsync(host_buf, dev, bytes, hipMemcpyDeviceToHost, S(s)) != hipSuccess) rc = fail(RT_E_RUNTIME, "rt_comm_broadcast: download failed");
    if (hipStreamSynchronize(S(s)) != hipSuccess && !rc) rc = fail(RT_E_RUNTIME, "rt_comm_broadcast: stream synchronisation failed");
    (void)hipFree(dev);
    (void)hipSetDevice(prev);
    return rc;
}

extern "C" int rt_comm_group_start(void) {
    RT_NEED_RCCL(api, "rt_comm_group_start");
    RT_RCCL(api, api->GroupStart(), "rt_comm_group_start");
    return 0;
}

extern "C" int rt_comm_group_end(void) {
    RT_NEED_RCCL(api, "rt_comm_group_end");
    RT_RCCL(api, api->GroupEnd(), "rt_comm_group_end");
    return 0;
}

extern "C" int rt_comm_destroy(rtComm* comm) {
    if (!comm) return 0;
    int rc = 0;
    if (comm->owned && comm->nccl) {
        RcclApi* api = rccl();
        if (api && api->CommDestroy(static_cast<ncclComm_t>(comm->nccl)) != ncclSuccess) rc = fail(RT_E_RUNTIME, "rt_comm_destroy: RCCL error");
    }
    delete comm;
    return rc;
}
#else
// SIMT-emulator build (CPU test tier): no RCCL.  rt_comm_init_rank serves a world of one rank; rt_comm_init_all -- the one-process,
// one-thread-per-device model of apps/stereo_throughput.cpp (ncclCommInitAll) -- returns communicators that share an in-process
// "fabric": rt_comm_broadcast then really is a collective (the root's bytes reach every rank, every rank must call it, the call returns
// when all have), so the threading of the native multi-GPU start-up can be exercised without a GPU (tests/test_multi_gpu.py).
#include <condition_variable>
namespace {
struct EmuFabric {
    std::mutex m;
    std::condition_variable cv;
    int world = 1, arrived = 0, refs = 0;
    unsigned long long gen = 0;
    bool ready = false;
    std::vector<char> data;
};
}  // namespace
extern "C" int rt_comm_unique_id(void* id_bytes) {
    RT_REQUIRE(id_bytes, "rt_comm_unique_id: null pointer");
    std::memset(id_bytes, 0x5a, RT_COMM_ID_BYTES);
    return 0;
}
extern "C" int rt_comm_init_rank(rtComm** comm, int world, int rank, const void* id_bytes) {
    RT_REQUIRE(comm && id_bytes && world == 1 && rank == 0, "rt_comm_init_rank: the emulator build has no RCCL (world size 1 only)");
    *comm = new rtComm();
    (*comm)->owned = true;
    return 0;
}
extern "C" int rt_comm_init_all(rtComm** comms, int ndev, const int*) {
    RT_REQUIRE(comms && ndev >= 1 && ndev <= 64, "rt_comm_init_all: bad arguments");
    EmuFabric* fab = ndev > 1 ? new EmuFabric() : nullptr;
    if (fab) { fab->world = ndev; fab->refs = ndev; }
    for (int i = 0; i < ndev; i++) {
        comms[i] = new rtComm();
        comms[i]->owned = true; comms[i]->world = ndev; comms[i]->rank = i; comms[i]->fabric = fab;
    }
    return 0;
}
extern "C" int rt_comm_adopt(rtComm**, void*) { return fail(RT_E_UNSUPPORTED, "rt_comm_adopt: the emulator build has no RCCL"); }
extern "C" int rt_comm_broadcast(rtComm* comm, void* host_buf, size_t bytes, int root, rtStream) {
    RT_REQUIRE(comm && (host_buf || !bytes) && root >= 0 && root < comm->world, "rt_comm_broadcast: bad arguments");
    auto* fab = static_cast<EmuFabric*>(comm->fabric);
    if (!fab || !bytes) return 0;
    std::unique_lock<std::mutex> lk(fab->m);
    const unsigned long long gen = fab->gen;
    int rc = 0;
    if (comm->rank == root) {
        fab->data.assign(static_cast<const char*>(host_buf), static_cast<const char*>(host_buf) + bytes);
        fab->ready = true;
        fab->cv.notify_all();
    } else {
        fab->cv.wait(lk, [&] { return fab->ready && fab->gen == gen; });
        if (fab->data.size() != bytes) rc = fail(RT_E_BADARG, "rt_comm_broadcast: rank %d expects %zu bytes, the root sent %zu", comm->rank, bytes, fab->data.size());
        else std::memcpy(host_buf, fab->data.data(), bytes);
    }
    if (++fab->arrived == fab->world) {                      // the last rank to arrive releases everybody (and the buffer)
        fab->arrived = 0; fab->ready = false; fab->gen++;
        fab->cv.notify_all();
    } else {
        fab->cv.wait(lk, [&] { return fab->gen != gen; });
    }
    return rc;
}
extern "C" int rt_comm_group_start(void) { return 0; }
extern "C" int rt_comm_group_end(void) { return 0; }
extern "C" int rt_comm_destroy(rtComm* comm) {
    if (comm && comm->fabric) {
        auto* fab = static_cast<EmuFabric*>(comm->fabric);
        bool last;
        { std::lock_guard<std::mutex> lk(fab->m); last = --fab->refs == 0; }
        if (last) delete fab;
    }
    delete comm;
    return 0;
}
#endif

extern "C" int rt_comm_info(const rtComm* comm, int* world, int* rank) {
    RT_REQUIRE(comm, "rt_comm_info: null pointer");
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    return 0;
}
