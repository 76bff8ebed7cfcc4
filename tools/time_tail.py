#!/usr/bin/env python3
"""Per-phase timing of conv_s3_kernel on the low-resolution layers of ResNet-18 2D at 1257x369 (64->64 @93x315, 128->128 @47x158;
s_memtime stamps of thread 0 of every workgroup; instrumented library from redtail_amd.build.build_hip_timing)."""
import ctypes
import os
os.environ.setdefault("RT_DEV_KNOBS", "1")
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import build, capi  # noqa: E402

k = capi.KernelLib.__new__(capi.KernelLib)
k.path = os.environ.get("RT_TIMING_LIB", os.path.join(build.ROOT, "tools", "build", "librt_stereo_hip_timing.so"))
k.lib = ctypes.CDLL(k.path)
for name, (res, args) in capi.KERNEL_SYMBOLS.items():
    fn = getattr(k.lib, name)
    fn.restype, fn.argtypes = res, args
for c, h, w in ((64, 93, 315), (128, 47, 158)):
    wt = (np.random.randn(c * c * 9).astype(np.float32) / np.float32(np.sqrt(9 * c)))
    plan = k.conv2d_plan(wt, np.random.randn(c).astype(np.float32), c, c, h, w, 3, 1, 1, act=capi.RT_ACT_ELU, has_residual=False)
    pitch = (w + 7) // 8 * 8
    plan.set_pitch(pitch, pitch)
    plan.set_layouts(1, 1, 0)
    x = torch.randn(1, c, h, pitch, device="cuda")
    y = torch.empty_like(x)
    nwg = ((w + 31) // 32) * ((h + 3) // 4) * (c // 32)
    dbg = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        plan.enqueue(x, y, None, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.enqueue(x, y, None, 1)
    e1.record()
    torch.cuda.synchronize()
    print("%d->%d @%dx%d: %d workgroups, %.1f us per launch back to back" % (c, c, h, w, nwg, e0.elapsed_time(e1) * 50))
    os.environ["RT_DBG_PTR"] = str(dbg.data_ptr())
    plan.enqueue(x, y, None, 1)
    torch.cuda.synchronize()
    os.environ.pop("RT_DBG_PTR")
    t = dbg.cpu().numpy().reshape(nwg, 16).astype(np.float64)
    t = t[t[:, 0] > 0]
    nch = c // 16
    last = min(2 + 2 * nch, 13)
    names = ["start", "loads issued"] + sum((["c%d in LDS" % i, "c%d MFMAs issued" % i] for i in range(nch)), []) + ["stores issued"]
    d = np.diff(t[:, :last + 1], axis=1)
    for i in range(last):
        print("  %-18s -> %-18s %9.1f %9.1f %9.1f" % (names[i], names[i + 1], d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    life = t[:, 14] - t[:, 0]
    print("  workgroup lifetime mean %.1f cycles; shader clock %.0f MHz; kernel span %.1f cycles; starts p90 %.1f after the first" % (
        life.mean(), life.sum() / t[:, 15].sum() * 100.0, t[:, 14].max() - t[:, 0].min(), np.percentile(t[:, 0] - t[:, 0].min(), 90)))
