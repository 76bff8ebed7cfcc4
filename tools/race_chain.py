#!/usr/bin/env python3
"""Producer -> consumer determinism under contention: a residual block as TWO dependent launches of the same plan family in one stream
(t = conv1(x); y = conv2(t) + x), NS streams with their own plans and tensors, t and y zeroed before every pass and y checked after it.
What tools/race_kernel.py cannot see: a consumer that reads its input before the producer's stores are visible.
    python tools/race_chain.py [iters] [streams]      (RT_KLIB: another build of the kernel library; CASE: substring filter)"""
import os
import sys
os.environ.setdefault("RT_DEV_KNOBS", "1")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
k = capi.KernelLib(os.environ.get("RT_KLIB"))
EX = capi.RT_CONV_EXACT_FP32
H, W, C = 185, 629, 32
P = (W + 31) // 32 * 32
rng = np.random.default_rng(7)
CASES = [("exact wino il chain", EX, (1, 1, 0), (1, 1, 1)), ("exact wino planar chain", EX, (0, 0, 0), (0, 0, 0)), ("split il chain", 0, (1, 1, 0), (1, 1, 1))]
only = os.environ.get("CASE")
for name, flags, lay1, lay2 in CASES:
    if only and only not in name:
        continue
    sets = []
    x0 = torch.randn(1, C, H, P, device="cuda")
    for s in range(NS):
        plans = []
        for lay, res in ((lay1, False), (lay2, True)):
            wt = (rng.standard_normal(C * C * 9) / np.sqrt(C * 9)).astype(np.float32)
            p = k.conv2d_plan(wt, rng.standard_normal(C).astype(np.float32), C, C, H, W, 3, 1, 1, act=capi.RT_ACT_ELU, has_residual=res, flags=flags)
            p.set_pitch(P, P)
            if any(lay):
                p.set_layouts(*lay)
            plans.append(p)
        sets.append((plans, x0.clone(), torch.zeros(1, C, H, P, device="cuda"), torch.zeros(1, C, H, P, device="cuda"), torch.cuda.Stream()))
    refs = []
    for plans, x, t, y, s in sets:          # every stream has its own weights: its own reference
        plans[0].enqueue(x, t, None, 1)
        plans[1].enqueue(t, y, x, 1)
        torch.cuda.synchronize()
        refs.append(y.clone())
    cnt = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in sets]
    worst = [torch.zeros((), device="cuda") for _ in sets]
    for it in range(iters):
        for i, (plans, x, t, y, s) in enumerate(sets):
            with torch.cuda.stream(s):
                t.zero_()
                y.zero_()
            plans[0].enqueue(x, t, None, 1, stream=s.cuda_stream)
            plans[1].enqueue(t, y, x, 1, stream=s.cuda_stream)
            with torch.cuda.stream(s):
                d = (y - refs[i]).abs().max()
                cnt[i] += (d > 0)
                worst[i] = torch.maximum(worst[i], d)
        if it % 256 == 255:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("%-26s %d passes on %d streams: %d deviating, worst %.3g" % (name, iters * NS, NS, sum(int(c) for c in cnt), max(float(w_) for w_ in worst)), flush=True)
    for plans, *_ in sets:
        for p in plans:
            p.destroy()
