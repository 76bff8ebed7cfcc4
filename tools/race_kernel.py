#!/usr/bin/env python3
"""Run-to-run determinism of ONE convolution plan under contention: NS streams, each with its own plan instance and tensors of the
same layer, launch it back to back; every output is compared bit for bit with the first result.  Narrows a whole-network
nondeterminism (tools/race_hunt.py) down to a kernel.   python tools/race_kernel.py [iters] [streams]"""
import os
import sys
os.environ.setdefault("RT_DEV_KNOBS", "1")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi  # noqa: E402

POISON = os.environ.get("RACE_POISON", "0") != "0"      # NaN instead of zeros (then the never-written padding columns show as well)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
k = capi.KernelLib(os.environ.get("RT_KLIB"))          # RT_KLIB: another build of the kernel library (probes)
EX = capi.RT_CONV_EXACT_FP32
H, W = 185, 629
rng = np.random.default_rng(5)
# name, cin, cout, h, w, ksize, stride, transposed, resid, (x_il, y_il, r_il), flags
CASES = [
    ("exact wino 32->32 il +res", 32, 32, H, W, 3, 1, False, True, (1, 1, 1), EX),
    ("exact wino 32->32 il", 32, 32, H, W, 3, 1, False, False, (1, 1, 0), EX),
    ("exact wino 32->32 planar +res", 32, 32, H, W, 3, 1, False, True, (0, 0, 0), EX),
    ("exact 5x5 s2 3->32 ->il", 3, 32, 369, 1257, 5, 2, False, False, (0, 1, 0), EX),
    ("exact 3x3 s2 32->64 il", 32, 64, H, W, 3, 2, False, False, (1, 1, 0), EX),
    ("exact wino 64->64 il", 64, 64, 93, 315, 3, 1, False, False, (1, 1, 0), EX),
    ("exact deconv 64->32 il +res", 64, 32, 93, 315, 3, 2, True, True, (1, 1, 1), EX),
    ("split 32->32 il +res", 32, 32, H, W, 3, 1, False, True, (1, 1, 1), 0),
    ("exact wino 32->32 il, planar res", 32, 32, H, W, 3, 1, False, True, (1, 1, 0), EX),
    ("exact wino 32->32 planar->il", 32, 32, H, W, 3, 1, False, False, (0, 1, 0), EX),
    ("exact wino 64->64 il->planar", 64, 64, 93, 315, 3, 1, False, False, (1, 0, 0), EX),
    ("exact wino 128->128 planar->il", 128, 128, 47, 158, 3, 1, False, False, (0, 1, 0), EX),
]
only = os.environ.get("CASE")
for name, cin, cout, h, w, ks, st, tr, res, (xi, yi, ri), flags in CASES:
    if only and only not in name:
        continue
    wt = (rng.standard_normal(cout * cin * ks * ks).astype(np.float32) / np.float32(np.sqrt(cin * ks * ks)))
    bias = rng.standard_normal(cout).astype(np.float32)
    if tr:
        ho, wo = 2 * h - 1 + 0, 2 * w - 1 + 0
        ho, wo = 185, 629
    else:
        ho, wo = (h + st - 1) // st, (w + st - 1) // st
    ip, op = (w + 31) // 32 * 32, (wo + 31) // 32 * 32
    sets = []
    x0 = torch.randn(1, cin, h, ip, device="cuda")
    r0 = torch.randn(1, cout, ho, op, device="cuda") if res else None
    for s in range(NS):
        plan = k.conv2d_plan(wt, bias, cin, cout, h, w, ks, st, 1 if tr else ks // 2, act=capi.RT_ACT_ELU, has_residual=res, transposed=tr, flags=flags)
        plan.set_pitch(ip, op)
        if xi or yi or ri:
            try:
                plan.set_layouts(xi, yi, ri)
            except capi.RtError as e:
                print("%-34s skipped: %s" % (name, str(e)[:80]))
                sets = None
                break
        sets.append((plan, x0.clone(), (r0.clone() if res else None), torch.zeros(1, cout, ho, op, device="cuda"), torch.cuda.Stream()))
    if sets is None:
        continue
    p, x, r, y, s = sets[0]
    p.enqueue(x, y, r, 1)
    torch.cuda.synchronize()
    ref = torch.nan_to_num(y.clone())
    cnt = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in sets]
    worst = [torch.zeros((), device="cuda") for _ in sets]
    keep = [None] * NS
    for it in range(iters):
        for i, (p, x, r, y, s) in enumerate(sets):
            with torch.cuda.stream(s):                      # poison the output first: a store that never lands must show (the previous launch left the same values)
                y.fill_(float("nan") if POISON else 0.0)
            p.enqueue(x, y, r, 1, stream=s.cuda_stream)
            with torch.cuda.stream(s):                      # every output is checked, on its own stream, without a host sync
                d = (torch.nan_to_num(y, nan=1e30) - ref).abs().max()
                cnt[i] += (d > 0)
                worst[i] = torch.maximum(worst[i], d)
        if it % 256 == 255:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("%-34s %d launches on %d streams: %d mismatching outputs, worst %.3g" % (
        name, iters * NS, NS, sum(int(c) for c in cnt), max(float(w_) for w_ in worst)), flush=True)
    for p, *_ in sets:
        p.destroy()
