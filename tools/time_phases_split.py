#!/usr/bin/env python3
"""Per-phase timing of the split-fp16 kernels on the 3x3 32->32 @185x629 layer (s_memtime stamps of thread 0 of every
workgroup; instrumented library from redtail_amd.build.build_hip_timing).  RT_S3P=1 selects the persistent kernel
instead of the general one (one workgroup per 4x32 tile)."""
import ctypes
import os
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import build, capi  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
general = os.environ.get("RT_S3P", "0") == "0"
block = os.environ.get("RT_TIME_BLOCK", "0") != "0"      # the fused residual block (1: conv_s3rbs_kernel, streaming; with RT_RB_TILES=1 conv_s3rb_kernel)
streaming = block and os.environ.get("RT_RB_TILES", "0") == "0"
k = capi.KernelLib.__new__(capi.KernelLib)
k.path = os.environ.get("RT_TIMING_LIB", os.path.join(build.ROOT, "tools", "build", "librt_stereo_hip_timing.so"))
k.lib = ctypes.CDLL(k.path)
for name, (res, args) in capi.KERNEL_SYMBOLS.items():
    fn = getattr(k.lib, name)
    fn.restype, fn.argtypes = res, args
h, w = 185, 629
wt = (np.random.randn(32 * 32 * 9).astype(np.float32) / np.float32(np.sqrt(288)))
if block:
    plan = k.resblock_plan(wt, np.random.randn(32).astype(np.float32), wt[::-1].copy(), np.random.randn(32).astype(np.float32), 32, 32, h, w)
else:
    plan = k.conv2d_plan(wt, np.random.randn(32).astype(np.float32), 32, 32, h, w, 3, 1, 1, act=capi.RT_ACT_ELU, has_residual=True)
plan.set_pitch(640, 640)
plan.set_layouts(1, 1, 1)
x = torch.randn(b, 32, h, 640, device="cuda")
y, r = torch.empty_like(x), (x if block else torch.randn_like(x))
nwg = 2 * 21 * 12 * b if streaming else 47 * 20 * b if block else (47 * 20 * b if general else 256)
dbg = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    plan.enqueue(x, y, r, b)
torch.cuda.synchronize()
os.environ["RT_DBG_PTR"] = str(dbg.data_ptr())
plan.enqueue(x, y, r, b)
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(nwg, 16).astype(np.float64)
t = t[t[:, 0] > 0]
if streaming:
    for role, nm in ((0, "conv1 wave 0"), (1, "conv2 wave 4")):
        tr = dbg.cpu().numpy().reshape(-1, 2, 16)[:, role].astype(np.float64)
        tr = tr[(tr[:, 0] > 0) & (tr[:, 13] > 0)]            # full segments (6 steps)
        names = ["start", "prologue loads issued", "prologue done (barrier)", "step 0 done", "s1 MFMAs issued", "s1 epilogue issued",
                 "s1 next rows in LDS", "s1 barrier", "s2 MFMAs issued", "s2 epilogue issued", "s2 next rows in LDS", "s2 barrier",
                 "step 3 done", "step 4 done", "end (step 5, stores acknowledged)"]
        d = np.diff(tr[:, :15], axis=1)
        print("streaming residual block, %s: phase durations in shader cycles (mean / p10 / p90 over %d workgroups)" % (nm, len(tr)))
        for i in range(14):
            print("  %-26s -> %-34s %9.1f %9.1f %9.1f" % (names[i], names[i + 1], d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
        life = tr[:, 14] - tr[:, 0]
        print("  workgroup lifetime mean %.1f cycles; shader clock %.0f MHz; kernel span %.1f cycles" % (
            life.mean(), life.sum() / tr[:, 15].sum() * 100.0, tr[:, 14].max() - tr[:, 0].min()))
    sys.exit(0)
if block:
    names = ["start", "gathers issued", "c0 in LDS", "conv1 c0 MFMAs", "c1 in LDS", "conv1 c1 MFMAs", "t written", "conv2 c0 MFMAs",
             "conv2 c1 MFMAs", "stores issued"]
    last = 9
elif general:
    names = ["start", "loads issued", "c0 in LDS", "c0 MFMAs issued", "c1 in LDS", "c1 MFMAs issued", "stores issued"]
    last = 6
else:
    names = ["start", "weights+gather0 issued", "t0 staged", "t0 barrier", "t0 MFMAs issued", "t0 stores issued",
             "t1 staged", "t1 barrier", "t1 MFMAs issued", "t1 stores issued"]
    t = t[t[:, 9] > 0]           # workgroups with (at least) two tiles
    last = 9
d = np.diff(t[:, :last + 1], axis=1)
print("%s kernel, batch %d: phase durations in shader cycles (mean / p10 / p90 over %d workgroups)" % ("residual-block" if block else "general" if general else "persistent", b, len(t)))
for i in range(last):
    print("  %-24s -> %-24s %9.1f %9.1f %9.1f" % (names[i], names[i + 1], d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("  last stamp -> all stores acknowledged %9.1f" % (t[:, 14] - t[:, last]).mean())
life = t[:, 14] - t[:, 0]
print("  workgroup lifetime mean %.1f cycles; shader clock %.0f MHz" % (life.mean(), life.sum() / t[:, 15].sum() * 100.0))
print("  kernel span (first start .. last end) %.1f cycles; starts p50 %.1f p90 %.1f max %.1f after the first; ends p10 %.1f p50 %.1f p90 %.1f" % (
    t[:, 14].max() - t[:, 0].min(), *(np.percentile(t[:, 0] - t[:, 0].min(), q) for q in (50, 90, 100)),
    *(np.percentile(t[:, 14] - t[:, 0].min(), q) for q in (10, 50, 90))))
