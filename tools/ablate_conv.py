#!/usr/bin/env python3
"""What bounds conv_mfma_f32_kernel?  Times the 3x3 32->32 @185x629 (+residual+ELU) layer with parts of the
kernel compiled out (RT_ABLATE mask, see conv_mfma.hip.h).  `build` cross-compiles the variants (CPU box),
`run` times them (GPU box)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import build, capi  # noqa: E402

VARIANTS = [(0, "full kernel"), (1, "no input gathers"), (2, "no weight loads"), (16, "no residual loads"), (32, "no stores"),
            (51, "no global memory traffic"), (128, "no input transform (Winograd)"), (179, "no memory, no transform")]


def main():
    if sys.argv[1] == "build":
        for m, _ in VARIANTS:
            build.build_hip_ablation(m)
        return
    import numpy as np
    import torch
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cin = cout = 32
    h, w = 185, 629
    wt = (np.random.randn(cout * cin * 9).astype(np.float32) / np.sqrt(cin * 9))
    bias = np.random.randn(cout).astype(np.float32)
    x = torch.randn(b, cin, h, w, device="cuda")
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
