#!/usr/bin/env python3
"""What does a launch pay when its CODE is cold (evicted from L2 / MALL by the ~1.5 GB a stereo pair streams through the chip)?
Two plans of the same shape (same kernel instantiation, different weights and tensors): after flushing the caches with a 2 GB
fill, time   B (code + data cold)   against   A then B (A warms the code only; B's weights and input are still cold)."""
import os
import sys
os.environ.setdefault("RT_DEV_KNOBS", "1")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from redtail_amd import capi  # noqa: E402

k = capi.KernelLib()
big = torch.empty(1 << 29, dtype=torch.float32, device="cuda")         # 2 GB


def mk(cin, cout, h, w, stride):
    wt = (np.random.randn(cout * cin * 9).astype(np.float32) / np.float32(np.sqrt(9 * cin)))
    plan = k.conv2d_plan(wt, np.random.randn(cout).astype(np.float32), cin, cout, h, w, 3, stride, 1, act=capi.RT_ACT_ELU, has_residual=False)
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    ip, op = (w + 7) // 8 * 8, (wo + 7) // 8 * 8
    plan.set_pitch(ip, op)
    plan.set_layouts(1, 1, 0)
    return plan, torch.randn(1, cin, h, ip, device="cuda"), torch.empty(1, cout, ho, op, device="cuda")


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for name, shp in (("128->128 s1 @47x158", (128, 128, 47, 158, 1)), ("64->128 s2 @93x315", (64, 128, 93, 315, 2)), ("32->32 s1 @185x629", (32, 32, 185, 629, 1))):
    A, B = mk(*shp), mk(*shp)
    for p in (A, B):
        for _ in range(3):
            p[0].enqueue(p[1], p[2], None, 1)
    torch.cuda.synchronize()
    res = {"cold": [], "code warm": [], "all warm": []}
    for _ in range(12):
        big.fill_(1.0); torch.cuda.synchronize()
        res["cold"].append(timed(lambda: B[0].enqueue(B[1], B[2], None, 1)))
        big.fill_(2.0); torch.cuda.synchronize()
        A[0].enqueue(A[1], A[2], None, 1); torch.cuda.synchronize()
        res["code warm"].append(timed(lambda: B[0].enqueue(B[1], B[2], None, 1)))
        res["all warm"].append(timed(lambda: B[0].enqueue(B[1], B[2], None, 1)))
    print("%-22s " % name + "  ".join("%s %.1f us (min %.1f)" % (n, float(np.median(v)), min(v)) for n, v in res.items()))
