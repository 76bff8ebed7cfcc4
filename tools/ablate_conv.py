#!/usr/bin/env python3
"""What bounds conv_mfma_f32_kernel?  Times the 3x3 32->32 @185x629 (+residual+ELU) layer with parts of the
kernel compiled out (RT_ABLATE mask, see conv_mfma.hip.h).  `build` cross-compiles the variants (CPU box),
`run` times them (GPU box)."""
import ctypes
import os
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import build, capi  # noqa: E402

VARIANTS = [(0, "full kernel"), (1, "no input gathers"), (2, "no weight loads"), (16, "no residual loads"), (32, "no stores"),
            (51, "no global memory traffic"), (128, "no input transform (Winograd)"), (179, "no memory, no transform")]


# conv_f16mma_kernel (half2 mode): `build16` / `run16 [batch]`
VARIANTS16 = [(0, "full kernel"), (1, "no input gathers"), (16, "no residual loads"), (32, "no stores"), (51, "no global memory traffic"),
              (64, "no LDS reads"), (256, "no MFMA"), (4 + 8 + 64, "no LDS, no barriers"), (51 + 4 + 8 + 64, "MFMA only")]


# occupancy of conv_f16mma_kernel: `buildocc16` / `runocc16 [batch]`
VARIANTSOCC = [(10000 + w, "%d waves per SIMD" % w) for w in (4, 5, 6, 7, 8)]
if os.environ.get("RT_ABL_AB"):          # A/B against a library built from another revision (tools/build/..._abl20000.so)
    VARIANTSOCC = [(20000, "reference revision"), (10008, "working tree")]


def main():
    f16 = sys.argv[1].endswith("16")
    if sys.argv[1].endswith("occ16"):
        global VARIANTS16
        VARIANTS16 = VARIANTSOCC
    global VARIANTS
    if f16:
        VARIANTS = VARIANTS16
    if os.environ.get("RT_ABL_AB") and not f16:          # fp32 A/B of two builds: `RT_ABL_AB=1 python tools/ablate_conv.py run [batch]`
        VARIANTS = [(20000, "reference revision"), (10008, "working tree")]
    if sys.argv[1].startswith("build"):
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(build.build_hip_ablation, [m for m, _ in VARIANTS]))
        return
    import numpy as np
    import torch
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cin = cout = 32
    h, w = 185, 629
    wt = (np.random.randn(cout * cin * 9).astype(np.float32) / np.float32(np.sqrt(cin * 9)))
    bias = np.random.randn(cout).astype(np.float32)
    pitch = 640 if (f16 or os.environ.get("RT_ABL_IL8")) else w
    x = torch.randn(b, cin, h, pitch, device="cuda", dtype=torch.float16 if f16 else torch.float32)
    y = torch.empty_like(x)
    r = torch.randn_like(x)
    for m, name in VARIANTS:
        k = capi.KernelLib.__new__(capi.KernelLib)
        k.path = os.path.join(build.ROOT, "tools", "build", "librt_stereo_hip_abl%d.so" % m)
        k.lib = ctypes.CDLL(k.path)
        for sym, (res, args) in capi.KERNEL_SYMBOLS.items():
            fn = getattr(k.lib, sym)
            fn.restype, fn.argtypes = res, args
        plan = k.conv2d_plan(wt, bias, cin, cout, h, w, 3, 1, 1, act=capi.RT_ACT_ELU, has_residual=True)
        if not f16 and os.environ.get("RT_ABL_IL8"):      # fp32 Winograd on channel-interleaved tensors
            plan.set_pitch(pitch, pitch)
            plan.set_layouts(1, 1, 1)
        if f16:
            plan.set_pitch(pitch, pitch)
            plan.set_io_types(capi.RT_F16, capi.RT_F16)
            if os.environ.get("RT_ABL_IL8"):          # channel-interleaved tensors (same bytes, other addressing)
                plan.set_layouts(1, 1, 1)
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        k.lib.rt_event_create(ctypes.byref(e0)); k.lib.rt_event_create(ctypes.byref(e1))
        for _ in range(3):
            plan.enqueue(x, y, r, b)
        torch.cuda.synchronize()
        k.lib.rt_event_record(e0, None)
        for _ in range(20):
            plan.enqueue(x, y, r, b)
        k.lib.rt_event_record(e1, None)
        ms = ctypes.c_float()
        k.lib.rt_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        us = ms.value * 1e3 / 20
        print("mask %3d %-34s %8.1f us  %6.1f TFLOP/s" % (m, name, us, 2.0 * b * 32 * 32 * 9 * h * w / us / 1e6))


if __name__ == "__main__":
    main()
