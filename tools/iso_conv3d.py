#!/usr/bin/env python3
"""One Conv3D layer of the half2 3-D trunk alone (BASELINE C5 shapes): conv_f16dw_kernel (depth walk) against conv_f16r4_kernel, and
instrumented builds of the former (RT_DW_ABL masks: 1 no patch loads, 2 no stores, 4 no MFMAs; RT_DW_VALU_PER_MFMA).
    python tools/iso_conv3d.py build             cross-compile the variants (CPU box) into tools/build/dw_<name>/
    python tools/iso_conv3d.py run [batch]       time them (GPU box)"""
import ctypes
import os
import sys
os.environ.setdefault("RT_DEV_KNOBS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import build, capi  # noqa: E402

VARIANTS = [("abl2", ["-DRT_DW_ABL=2"], "no stores"), ("abl16", ["-DRT_DW_ABL=16"], "patch ring filled once (real data)"),
            ("abl18", ["-DRT_DW_ABL=18"], "ring filled once, no stores"), ("abl4", ["-DRT_DW_ABL=4"], "no MFMAs"),
            ("abl3", ["-DRT_DW_ABL=3"], "no patch loads (zeros), no stores")]
# NVSmall 1025x321 (nvsmall_1025x321_net.cpp:331-399): conv3D_2, conv3D_4/5, conv3D_7/8; ResNet-18 3D conv3D_1b
SHAPES = [("conv3D_2   32->32  48x161x513", 32, 32, 48, 161, 513), ("conv3D_4   64->64  24x81x257", 64, 64, 24, 81, 257),
          ("conv3D_7  128->128 12x41x129", 128, 128, 12, 41, 129), ("r18 1b    32->32  68x161x513", 32, 32, 68, 161, 513)]


def klib(path):
    k = capi.KernelLib.__new__(capi.KernelLib)
    k.path = path
    k.lib = ctypes.CDLL(path)
    for sym, (res, args) in capi.KERNEL_SYMBOLS.items():
        fn = getattr(k.lib, sym)
        fn.restype, fn.argtypes = res, args
    return k


def time_layer(k, c, kk, d, h, w, batch, env, iters=10):
    import numpy as np
    import torch
    for key, val in env.items():
        os.environ[key] = val
    rng = np.random.default_rng(1)
    wt = (rng.standard_normal((kk, 3, c, 3, 3)) / np.sqrt(27 * c)).astype(np.float16)
    b = rng.standard_normal(kk).astype(np.float16)
    plan = k.conv3d_plan(wt, b, c, kk, (d, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU, out_dchw=True, dtype=capi.RT_F16)
    plan.set_io_types(capi.RT_F16, capi.RT_F16)
    plan.set_layouts(1, 1, 0)
    x = (torch.randn(batch, d, c // 8, h, w, 8, device="cuda") * 0.5).half()
    y = torch.empty(batch, d, kk // 8, h, w, 8, device="cuda", dtype=torch.float16)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    k.lib.rt_event_create(ctypes.byref(e0)); k.lib.rt_event_create(ctypes.byref(e1))
    for _ in range(3):
        plan.enqueue(x, y, None, batch)
    torch.cuda.synchronize()
    k.lib.rt_event_record(e0, None)
    for _ in range(iters):
        plan.enqueue(x, y, None, batch)
    k.lib.rt_event_record(e1, None)
    torch.cuda.synchronize()
    ms = ctypes.c_float()
    k.lib.rt_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    plan.destroy()
    for key in env:
        os.environ.pop(key, None)
    return ms.value / iters / batch, float(y.float().abs().mean())


FOLD_VARIANTS = [("fabl1", ["-DRT_FOLD_ABL=1"], "no T loads"), ("fabl2", ["-DRT_FOLD_ABL=2"], "no activation"), ("fabl4", ["-DRT_FOLD_ABL=4"], "no stores"),
                 ("fabl3", ["-DRT_FOLD_ABL=3"], "no T loads, no activation"), ("fnolds", ["-DRT_FOLD_T_LDS=0"], "T from memory (no LDS window)")]


def time_fold(k, f, kk, d, h, w, batch, iters=10):
    """the first Conv3D of a 3-D model over the folded cost volume (factored form): NVSmall 2 x 32 -> 32, 48 x 161 x 513, fp16 interleaved out"""
    import numpy as np
    import torch
    rng = np.random.default_rng(1)
    wt = (rng.standard_normal((kk, 3, 2 * f, 3, 3)) / np.sqrt(27 * 2 * f)).astype(np.float32)
    b = rng.standard_normal(kk).astype(np.float32)
    plan = k.conv3d_plan(wt, b, 2 * f, kk, (d, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU, out_dchw=True, cv_fold=f)
    plan.set_io_types(capi.RT_F32, capi.RT_F16)
    plan.set_layouts(0, 1, 0)
    x = torch.randn(batch, 2 * f, h, w, device="cuda") * 0.5
    y = torch.empty(batch, d, kk // 8, h, w, 8, device="cuda", dtype=torch.float16)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    k.lib.rt_event_create(ctypes.byref(e0)); k.lib.rt_event_create(ctypes.byref(e1))
    for _ in range(3):
        plan.enqueue(x, y, None, batch)
    torch.cuda.synchronize()
    k.lib.rt_event_record(e0, None)
    for _ in range(iters):
        plan.enqueue(x, y, None, batch)
    k.lib.rt_event_record(e1, None)
    torch.cuda.synchronize()
    ms = ctypes.c_float()
    k.lib.rt_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    plan.destroy()
    return ms.value / iters / batch


def main():
    if sys.argv[1] == "buildfold":
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(lambda v: build.build_variant("dw_" + v[0], v[1], kernels_only=True), FOLD_VARIANTS))
        return
    if sys.argv[1] == "fold":
        import torch
        torch.zeros(1, device="cuda")
        batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        prod = os.path.join(build.ROOT, "redtail_amd", "lib", "librt_stereo_hip.so")
        for u in ("1", "2", "4"):
            os.environ["RT_FOLD_U"] = u
            print("conv3D_1 (factored fold) 2x32->32 48x161x513, batch %d, %d depth slice(s) per trip of the combining pass: %.4f ms per pair" % (
                batch, int(u), time_fold(klib(prod), 32, 32, 48, 161, 513, batch)), flush=True)
        os.environ.pop("RT_FOLD_U")
        rows = [("product", prod)]
        rows += [(what, os.path.join(build.ROOT, "tools", "build", "dw_" + v, "librt_stereo_hip.so")) for v, _, what in FOLD_VARIANTS]
        for what, pth in rows:
            if os.path.exists(pth):
                print("conv3D_1 (factored fold) 2x32->32 48x161x513, batch %d, %-28s %.4f ms per pair" % (batch, what, time_fold(klib(pth), 32, 32, 48, 161, 513, batch)), flush=True)
        return
    if sys.argv[1] == "build":
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(lambda v: build.build_variant("dw_" + v[0], v[1], kernels_only=True), VARIANTS))
        return
    import torch
    torch.zeros(1, device="cuda")            # torch's HIP runtime first: a kernel library loaded before it leaves torch without a device
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    if sys.argv[1] == "one":                 # python tools/iso_conv3d.py one <batch> <shape index> <dw 0|1> [variant]: a few launches for a profiler
        name, c, kk, d, h, w = SHAPES[int(sys.argv[3])]
        var = sys.argv[5] if len(sys.argv) > 5 else None
        k = klib(os.path.join(build.ROOT, "tools", "build", "dw_" + var, "librt_stereo_hip.so") if var else os.path.join(build.ROOT, "redtail_amd", "lib", "librt_stereo_hip.so"))
        ms, _ = time_layer(k, c, kk, d, h, w, batch, {"RT_F16_DW": sys.argv[4]}, iters=5)
        print(name, "dw" + sys.argv[4], var, "%.4f ms per pair" % ms)
        return
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    base = klib(os.path.join(build.ROOT, "redtail_amd", "lib", "librt_stereo_hip.so"))
    print("per pair at batch %d: ms, TFLOP/s of the direct form, fraction of 2.5 PF" % batch)
    for name, c, kk, d, h, w in SHAPES:
        gf = 2.0 * 27 * c * kk * d * h * w / 1e9
        rows = [("4 x 32 tiles (conv_f16mma_kernel)", base, {"RT_F16_DW": "0", "RT_F16_R4": "0"}), ("r4 (per slice)", base, {"RT_F16_DW": "0"}),
                ("depth walk", base, {"RT_F16_DW": "1"})]
        for ns in (1, 2, 3, 4):
            rows.append(("depth walk, %d segment(s)" % ns, base, {"RT_F16_DW": "1", "RT_DW_NSEG": str(ns)}))
        for v, _, what in VARIANTS:
            if only and v not in only:
                continue
            pth = os.path.join(build.ROOT, "tools", "build", "dw_" + v, "librt_stereo_hip.so")
            if os.path.exists(pth):
                rows.append(("depth walk, " + what, klib(pth), {"RT_F16_DW": "1"}))
        for label, k, env in rows:
            ms, chk = time_layer(k, c, kk, d, h, w, batch, env)
            print("%-30s %-44s %7.4f ms  %7.1f TF  %.3f   (mean |y| %.4f)" % (name, label, ms, gf / ms, gf / ms / 2500.0, chk), flush=True)


if __name__ == "__main__":
    main()
