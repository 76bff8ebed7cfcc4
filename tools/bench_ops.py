#!/usr/bin/env python3
"""Op-level microbenchmarks through the C ABI at the BASELINE shapes.

Prints one line per op: average time (HIP events on the launch stream), algorithmic GB/s or TFLOP/s
and the fraction of the gfx950 roofline that bounds it: HBM 8 TB/s, or -- ONE roof convention with bench.py (VERDICT r04 weak #8b) -- the
fp16 matrix pipe the fp32 convolutions execute on: direct-form FLOPs x 3 fp16 products per multiply / 2.5 PFLOP/s dense.
    python tools/bench_ops.py [--iters 50] [--only conv]
"""
import argparse
import ctypes
import json
import os
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi  # noqa: E402

HBM_PEAK = 8.0e12
MFMA_F16_PEAK = 2.5e15          # dense fp16; an fp32 multiply is three fp16 products (3-term split)
SPLIT_TERMS = 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default="")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    k = capi.KernelLib()
    print("backend:", k.backend())
    dev = "cuda"
    results = []

    def timeit(fn, iters=None):
        iters = iters or args.iters
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        k.lib.rt_event_create(ctypes.byref(e0)); k.lib.rt_event_create(ctypes.byref(e1))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        k.lib.rt_event_record(e0, None)
        for _ in range(iters):
            fn()
        k.lib.rt_event_record(e1, None)
        ms = ctypes.c_float()
        k.check(k.lib.rt_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "elapsed")
        return ms.value * 1e-3 / iters

    def report(name, t, flops=0.0, nbytes=0.0, bound="hbm"):
        tf, gb = flops / t / 1e12, nbytes / t / 1e9
        frac = (SPLIT_TERMS * flops / t / MFMA_F16_PEAK) if bound == "mfma" else (nbytes / t / HBM_PEAK)
        print("%-46s %9.1f us  %8.2f TFLOP/s  %8.1f GB/s  %5.1f%% of %s roof" % (name, t * 1e6, tf, gb, 100 * frac, "fp16-mfma (x3 products)" if bound == "mfma" else bound))
        results.append(dict(name=name, us=t * 1e6, tflops=tf, gbps=gb, bound=bound, frac=frac))

    def want(name):
        return not args.only or args.only in name

    def rnd(*s):
        return torch.randn(*s, device=dev)

    # ---- 2-D convolutions of ResNet-18 2D at 1257x369 (half-res 629x185) ---------------------------------
    conv_cases = [
        ("conv3x3 32->32 @185x629 +res+ELU b1", 32, 32, 185, 629, 3, 1, 1, True, 1, False),
        ("conv3x3 32->32 @185x629 +res+ELU b2", 32, 32, 185, 629, 3, 1, 1, True, 2, False),
        ("conv3x3 32->32 @185x629 +res+ELU b8", 32, 32, 185, 629, 3, 1, 1, True, 8, False),
        ("conv3x3 32->32 @185x629 +ELU (no residual) b8", 32, 32, 185, 629, 3, 1, 1, False, 8, False),
        ("conv3x3 32->32 @185x640 (aligned rows) b8", 32, 32, 185, 640, 3, 1, 1, True, 8, False),
        ("conv3x3 32->32 @185x640 (aligned rows) b1", 32, 32, 185, 640, 3, 1, 1, True, 1, False),
        ("conv5x5s2 3->32 @369x1257 +ELU b2", 3, 32, 369, 1257, 5, 2, 2, False, 2, False),
        ("conv3x3 33->32 @185x629 +ELU b1", 33, 32, 185, 629, 3, 1, 1, False, 1, False),
        ("conv3x3s2 32->64 @185x629 +ELU b1", 32, 64, 185, 629, 3, 2, 1, False, 1, False),
        ("conv3x3 64->64 @93x315 +ELU b1", 64, 64, 93, 315, 3, 1, 1, False, 1, False),
        ("conv3x3s2 64->128 @93x315 +ELU b1", 64, 128, 93, 315, 3, 2, 1, False, 1, False),
        ("conv3x3 128->128 @47x158 +ELU b1", 128, 128, 47, 158, 3, 1, 1, False, 1, False),
        ("deconv3x3s2 128->64 @47x158 +res+ELU b1", 128, 64, 47, 158, 3, 2, 1, True, 1, True),
        ("deconv3x3s2 64->32 @93x315 +res+ELU b1", 64, 32, 93, 315, 3, 2, 1, True, 1, True),
        ("deconv3x3s2 32->1 @185x629 +sigmoid b1", 32, 1, 185, 629, 3, 2, 1, False, 1, True),
    ]
    for name, cin, cout, h, w, ks, st, pad, res, b, tr in conv_cases:
        if not want(name):
            continue
        wt = (np.random.randn(cin * cout * ks * ks).astype(np.float32) / np.float32(np.sqrt(cin * ks * ks)))
        bias = np.random.randn(cout).astype(np.float32)
        plan = k.conv2d_plan(wt, bias, cin, cout, h, w, ks, st, pad, act=capi.RT_ACT_ELU, has_residual=res, transposed=tr)
        co, ho, wo, _ = plan.out_dims
        x, y = rnd(b, cin, h, w), torch.empty(b, co, ho, wo, device=dev)
        r = rnd(b, co, ho, wo) if res else None
        t = timeit(lambda: plan.enqueue(x, y, r, b))
        if tr:
            flops = 2.0 * b * cin * cout * 9 * h * w
        else:
            flops = 2.0 * b * cin * cout * ks * ks * ho * wo
        nbytes = 4.0 * (x.numel() + y.numel() + (r.numel() if res else 0))
        report(name, t, flops, nbytes, "mfma")
        plan.destroy()

    # ---- cost volume / softargmax / element-wise at C2 shapes -----------------------------------------------
    C, H, W, D = 32, 185, 629, 48
    for b in (1, 8):
        l, r = rnd(b, C, H, W), rnd(b, C, H, W)
        cv, sa = torch.empty(b, D, H, W, device=dev), torch.empty(b, 1, H, W, device=dev)
        if want("corr"):
            t = timeit(lambda: k.corr_cost_volume(l, r, cv, b, C, H, W, D))
            report("corr cost volume C32 D48 @185x629 b%d" % b, t, 2.0 * b * C * D * H * W, 4.0 * b * (2 * C + D) * H * W)
            os.environ["RT_NO_CORR_MFMA_PLANAR"] = "1"      # the fp32 fmaf kernel (small maps, D > 64, RT_CONV_EXACT_FP32)
            t = timeit(lambda: k.corr_cost_volume(l, r, cv, b, C, H, W, D))
            del os.environ["RT_NO_CORR_MFMA_PLANAR"]
            report("  the same on the vector ALU (corr_f32_kernel) b%d" % b, t, 2.0 * b * C * D * H * W, 4.0 * b * (2 * C + D) * H * W)
            if b == 1:
                # the op-level CPU baseline of BASELINE.md section 2: the reference's kernel as a plain single-thread C loop
                # (oracle/corr_cpu.c restates lib/kernels.cu:168-200), same tensors, one core, result checked against the GPU's
                from redtail_amd import build
                import time
                lib = ctypes.CDLL(build.build_oracle_c())
                lh, rh = l.cpu().numpy(), r.cpu().numpy()
                ch = np.empty((1, D, H, W), np.float32)
                fp = ctypes.POINTER(ctypes.c_float)
                args_ = (lh.ctypes.data_as(fp), rh.ctypes.data_as(fp), 1, C, H, W, D, ch.ctypes.data_as(fp))
                lib.corr_cost_volume_cpu(*args_)
                n_, t0_ = 0, time.perf_counter()
                while n_ < 2 or time.perf_counter() - t0_ < 3.0:
                    lib.corr_cost_volume_cpu(*args_)
                    n_ += 1
                tc = (time.perf_counter() - t0_) / n_
                err = float(np.abs(cv.cpu().numpy() - ch).max())
                print("%-46s %9.1f us  (%d runs, 1 core, plain C restatement of corrCostVolumeKernel; GPU / CPU = %.0fx; max |GPU - CPU| = %.2g)" % (
                    "  CPU corr cost volume C32 D48 @185x629 b1", tc * 1e6, n_, tc / t, err))
                results.append(dict(name="cpu corr cost volume C32 D48 @185x629 b1 (oracle/corr_cpu.c, 1 core)", us=tc * 1e6, kind="port", cores=1,
                                    max_abs_diff_vs_gpu=err))
            t = timeit(lambda: k.corr_softargmax(l, r, sa, b, C, H, W, D, False))
            report("corr+softargmax fused, planar fp32 maps (corr_softargmax_mfma_kernel<planar>) b%d" % b, t, 2.0 * b * C * D * H * W, 4.0 * b * (2 * C + 1) * H * W)
            t = timeit(lambda: k.corr_softargmax_pitched(l, r, sa, b, C, H, W, D, False, 0, 0))
            report("  the same on the vector ALU (corr_f32_kernel: exact-fp32 engines) b%d" % b, t, 2.0 * b * C * D * H * W, 4.0 * b * (2 * C + 1) * H * W)
            # the form the engine runs: channel-interleaved (C/4, H, pitch, 4) feature maps, Gram band on the matrix cores (corr_mfma.hip.h)
            P = (W + 31) // 32 * 32
            li, ri = rnd(b, C // 4, H, P, 4), rnd(b, C // 4, H, P, 4)
            so = torch.empty(b, 1, H, P, device=dev)
            t = timeit(lambda: k.corr_softargmax_il(li, ri, so, b, C, H, W, D, False, P, P))
            report("corr+softargmax fused, interleaved (corr_softargmax_mfma_kernel) b%d" % b, t, 2.0 * b * C * D * H * W, 4.0 * b * (2 * C + 1) * H * W)
        if want("softargmax"):
            t = timeit(lambda: k.softargmax(cv, sa, b, D, H, W, False))
            report("softargmax D48 @185x629 b%d" % b, t, 0, 4.0 * b * (D + 1) * H * W)
        if want("elu"):
            y = torch.empty_like(l)
            t = timeit(lambda: k.elu(l, y, l.numel()))
            report("ELU 32x185x629 b%d" % b, t, 0, 8.0 * l.numel())
            t = timeit(lambda: k.add_act(l, r, y, l.numel(), capi.RT_ACT_ELU))
            report("add+ELU 32x185x629 b%d" % b, t, 0, 12.0 * l.numel())

    # ---- 3-D models ---------------------------------------------------------------------------------------
    if want("3d"):
        for name, Dd, Cc, Kk, Hh, Ww in (("conv3D_1 NVTiny (24,16,81,257)->16", 24, 16, 16, 81, 257),
                                         ("conv3D_1 NVSmall (48,64,161,513)->32", 48, 64, 32, 161, 513)):
            wt = (np.random.randn(Kk * 3 * Cc * 9).astype(np.float32) / np.float32(np.sqrt(27 * Cc)))
            bias = np.random.randn(Kk).astype(np.float32)
            plan = k.conv3d_plan(wt, bias, Cc, Kk, (Dd, Hh, Ww), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1),
                                 act=capi.RT_ACT_ELU, out_dchw=True)
            x, y = rnd(1, Dd, Cc, Hh, Ww), torch.empty((1,) + plan.out_dims, device=dev)
            t = timeit(lambda: plan.enqueue(x, y, None, 1), iters=max(3, args.iters // 10))
            report(name, t, 2.0 * 27 * Cc * Kk * Dd * Hh * Ww, 4.0 * (x.numel() + y.numel()), "mfma")
            plan.destroy()
        Cc, Hh, Ww, Dd = 32, 161, 513, 48
        l, r = rnd(1, Cc, Hh, Ww), rnd(1, Cc, Hh, Ww)
        cv = torch.empty(1, Dd, 2 * Cc, Hh, Ww, device=dev)
        t = timeit(lambda: k.cost_volume(l, r, cv, 1, Cc, Hh, Ww, Dd), iters=max(3, args.iters // 10))
        report("default cost volume NVSmall (48,64,161,513)", t, 0, 4.0 * (2 * Cc + 2 * Cc * Dd) * Hh * Ww)
        vol, out = rnd(1, 96, 321, 1025), torch.empty(1, 1, 321, 1025, device=dev)
        t = timeit(lambda: k.softargmax(vol, out, 1, 96, 321, 1025, True))
        report("softargmin D96 @321x1025 (NVSmall, C5)", t, 0, 4.0 * 97 * 321 * 1025)
        # layout plugins at the NVSmall sizes (lib/transform_plugin.cpp, padding_plugin.cpp, slice_plugin.cpp; fused away in the engine)
        Kk, Dd2, Hh2, Ww2 = 32, 48, 161, 513
        x4, y4 = rnd(1, Kk, Dd2, Hh2, Ww2), torch.empty(1, Dd2, Kk, Hh2, Ww2, device=dev)
        t = timeit(lambda: k.permute4d(x4, y4, 1, (Kk, Dd2, Hh2, Ww2), (1, 0, 2, 3)), iters=max(3, args.iters // 10))
        report("transform (K,D,H,W)->(D,K,H,W) NVSmall (32,48,161,513)", t, 0, 8.0 * x4.numel())
        inner = Kk * Hh2 * Ww2
        xp, yp = rnd(1, Dd2, inner), torch.empty(1, Dd2 + 1, inner, device=dev)
        t = timeit(lambda: k.pad_d(xp, yp, 1, Dd2, inner, 1), iters=max(3, args.iters // 10))
        report("pad D 48->49 NVSmall (48,32,161,513)", t, 0, 4.0 * (xp.numel() + yp.numel()))
        xs, ys = rnd(1, 97, 321 * 1025), torch.empty(1, 96, 321 * 1025, device=dev)
        t = timeit(lambda: k.slice_d(xs, ys, 1, 97, 321 * 1025, 0, 96))
        report("slice D 97->96 NVSmall (97,1,321,1025)", t, 0, 8.0 * ys.numel())

    if args.json:
        with open(args.json, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
