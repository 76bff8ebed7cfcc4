#!/usr/bin/env python3
"""The dominant layer alone: 3x3 32->32 @629x185 (+bias, +residual, +ELU) on the executor's tensor layouts, or the fused
residual block, launched back to back on an idle GPU.  For rocprofv3 passes (tools/pmc_layer.sh) and quick timings.

    python tools/iso_layer.py [conv|block] [launches] [batch] [hints] [half2]      (hints 1 = RT_HINT_THROUGHPUT: 64-row segments of the block)
"""
import ctypes
import os
os.environ.setdefault("RT_DEV_KNOBS", "1")      # the RT_* switches this tool uses are development knobs
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
b = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hints = int(sys.argv[4]) if len(sys.argv) > 4 else 0
half2 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
k = capi.KernelLib()
H, W = 185, 629
rng = np.random.default_rng(1)
wt = (rng.standard_normal((32, 32, 3, 3)) / np.sqrt(288)).astype(np.float32)
bias = rng.standard_normal(32).astype(np.float32)
if kind == "block":
    plan = k.resblock_plan(wt, bias, wt[::-1].copy(), bias, 32, 32, H, W)
else:
    plan = k.conv2d_plan(wt, bias, 32, 32, H, W, 3, 1, 1, act=capi.RT_ACT_ELU, has_residual=True)
plan.set_pitch(640, 640)
if half2:
    plan.set_io_types(capi.RT_F16, capi.RT_F16)
plan.set_layouts(1, 1, 1)
x = torch.randn(b, 32, H, 640, device="cuda")
if half2:
    x = x.half()
elif kind == "block" and plan.supports_split() and not os.environ.get("ISO_NO_SPLIT"):
    # the typical tower block reads and writes pre-split tensors (conv_s3rbd_kernel): (C/8, H, pitch, [8 hi | 8 lo]) fp16 pairs
    plan.set_split(1, 1)
    g = x.reshape(b, 4, 8, H, 640).permute(0, 1, 3, 4, 2)
    hi = g.half()
    x = torch.cat([hi, ((g - hi.float()) * 2048.0).half()], dim=-1).contiguous().view(torch.float32)
y = torch.empty_like(x)
r = x if kind == "block" else torch.randn_like(x)
for _ in range(3):
    plan.enqueue(x, y, r, b, hints=hints)
torch.cuda.synchronize()
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
k.lib.rt_event_create(ctypes.byref(e0)); k.lib.rt_event_create(ctypes.byref(e1))
k.lib.rt_event_record(e0, None)
for _ in range(launches):
    plan.enqueue(x, y, r, b, hints=hints)
k.lib.rt_event_record(e1, None)
torch.cuda.synchronize()
ms = ctypes.c_float()
k.lib.rt_event_elapsed_ms(e0, e1, ctypes.byref(ms))
print("%s batch %d hints %d half2 %d: %.2f us per launch (%d launches back to back)" % (kind, b, hints, half2, ms.value * 1e3 / launches, launches))
