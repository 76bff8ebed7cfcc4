#!/usr/bin/env python3
"""Per-phase timing of conv_mfma_f32_kernel on the 3x3 32->32 @185x629 layer (s_memtime stamps of
thread 0 of every workgroup; instrumented library built by redtail_amd.build.build_hip_timing)."""
import os
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import build, capi  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
F16 = len(sys.argv) > 2 and sys.argv[2] == "f16"      # conv_f16mma_kernel (half2 mode) instead of the fp32 kernel
k = capi.KernelLib.__new__(capi.KernelLib)
import ctypes  # noqa: E402
k.path = os.environ.get("RT_TIMING_LIB", os.path.join(build.ROOT, "tools", "build", "librt_stereo_hip_timing.so"))
k.lib = ctypes.CDLL(k.path)
for name, (res, args) in capi.KERNEL_SYMBOLS.items():
    fn = getattr(k.lib, name)
    fn.restype, fn.argtypes = res, args
cin = cout = 32
h, w = 185, 629
wt = (np.random.randn(cout * cin * 9).astype(np.float32) / np.float32(np.sqrt(cin * 9)))
bias = np.random.randn(cout).astype(np.float32)
plan = k.conv2d_plan(wt, bias, cin, cout, h, w, 3, 1, 1, act=capi.RT_ACT_ELU, has_residual=True)
x = torch.randn(b, cin, h, 640 if F16 else w, device="cuda", dtype=torch.float16 if F16 else torch.float32)
y = torch.empty_like(x)
r = torch.randn_like(x)
if F16:
    plan.set_pitch(640, 640)
    plan.set_io_types(capi.RT_F16, capi.RT_F16)
    if len(sys.argv) > 3 and sys.argv[3] == "il8":      # channel-interleaved tensors (same bytes, other addressing)
        plan.set_layouts(1, 1, 1)
nwg = 47 * 20 * b          # 4 x 32 pixel tiles
dbg = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    plan.enqueue(x, y, r, b)
torch.cuda.synchronize()
os.environ["RT_DBG_PTR"] = str(dbg.data_ptr())
plan.enqueue(x, y, r, b)
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(nwg, 16).astype(np.float64)
names = ["start", "c0 sync1", "c0 ready", "c1 sync1", "c1 ready", "c2 sync1", "c2 ready", "c3 sync1", "c3 ready",
         "epilogue start", "end"]
if F16:
    names = ["start", "loads issued", "residual here", "c0 in LDS", "c0 MFMAs issued", "c1 in LDS", "c1 MFMAs issued",
             "epilogue start", "stores done"]
    t[:, 9:11] = t[:, 8:9]
d = np.diff(t[:, :11], axis=1)
print("batch %d: phase durations in shader cycles (mean / p10 / p90 over %d workgroups)" % (b, nwg))
for i in range(len(names) - 1):
    print("  %-16s -> %-16s %9.1f %9.1f %9.1f" % (names[i], names[i + 1], d[:, i].mean(), np.percentile(d[:, i], 10),
                                                 np.percentile(d[:, i], 90)))
print("  workgroup lifetime mean %.1f" % (t[:, 10] - t[:, 0]).mean())
print("  shader clock during the kernel: %.0f MHz (s_memtime ticks per 100 MHz s_memrealtime tick)" % (
    (t[:, 10] - t[:, 0]).sum() / t[:, 15].sum() * 100.0))
print("  kernel span (first start .. last end) %.1f; starts: p50 %.1f p90 %.1f max %.1f after the first" % (
    t[:, 10].max() - t[:, 0].min(), np.percentile(t[:, 0] - t[:, 0].min(), 50), np.percentile(t[:, 0] - t[:, 0].min(), 90),
    (t[:, 0] - t[:, 0].min()).max()))
print("  ends: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(t[:, 10] - t[:, 0].min(), q) for q in (10, 50, 90, 100)))
