#!/usr/bin/env python3
"""Reproducer of the multi-context deviation of the interleaved Winograd kernel (profiles/r04_race.txt): ONE victim plan -- exact-fp32
3x3 32 -> 32 @185x629 on interleaved tensors, conv_wino_f32_kernel<4, float, float, il, il> -- runs back to back on one stream while an
AGGRESSOR plan runs back to back on three other streams; every victim output is compared bit for bit with its first result.
    python tools/race_pair.py [iterations]          (needs an RT_EXPERIMENTAL build: RT_VARIANT_DIR=tools/build/<dir>)"""
import os
import sys
os.environ.setdefault("RT_DEV_KNOBS", "1")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
vdir = os.environ.get("RT_VARIANT_DIR")
k = capi.KernelLib(os.path.join(vdir, "librt_stereo_hip.so") if vdir else None)
EX, H, W, P, C = capi.RT_CONV_EXACT_FP32, 185, 629, 640, 32
rng = np.random.default_rng(5)


def conv(cin, cout, ks, st, flags, il, res=True, h=H, w=W, p=P):
    wt = (rng.standard_normal(cout * cin * ks * ks) / np.sqrt(cin * ks * ks)).astype(np.float32)
    plan = k.conv2d_plan(wt, rng.standard_normal(cout).astype(np.float32), cin, cout, h, w, ks, st, ks // 2, act=capi.RT_ACT_ELU, has_residual=res, flags=flags)
    ho, wo = (h + st - 1) // st, (w + st - 1) // st
    po = (wo + 31) // 32 * 32
    plan.set_pitch(p if cin > 3 else 0, po)
    if il:
        plan.set_layouts(*il)
    x = torch.randn(1, cin, h, p if cin > 3 else w, device="cuda")
    y = torch.zeros(1, cout, ho, po, device="cuda")
    r = torch.randn(1, cout, ho, po, device="cuda") if res else None
    return lambda s, hints=0: plan.enqueue(x, y, r, 1, stream=s, hints=hints), y, plan


def block():
    w1 = (rng.standard_normal((32, 32, 3, 3)) / np.sqrt(288)).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    plan = k.resblock_plan(w1, b, w1[::-1].copy(), b, 32, 32, H, W)
    plan.set_pitch(P, P)
    plan.set_layouts(1, 1, 1)
    x = torch.randn(2, 32, H, P, device="cuda")
    y = torch.zeros_like(x)
    return lambda s, hints=capi.RT_HINT_THROUGHPUT: plan.enqueue(x, y, x, 2, stream=s, hints=hints), y, plan


def fill():
    t = torch.zeros(8 << 20, device="cuda")
    return lambda s: t.add_(1.0), t, None


AGGRESSORS = [
    ("nothing (victim alone)", None),
    ("the same kernel (exact Winograd, interleaved)", lambda: conv(32, 32, 3, 1, EX, (1, 1, 1))),
    ("exact Winograd, planar tensors", lambda: conv(32, 32, 3, 1, EX, None)),
    ("exact 3x3 stride 2 32->64 (conv_mfma_f32_kernel)", lambda: conv(32, 64, 3, 2, EX, None, res=False)),
    ("split-fp16 3x3 32->32 interleaved (conv_s3_kernel)", lambda: conv(32, 32, 3, 1, 0, (1, 1, 1))),
    ("split-fp16 residual block (conv_s3rbs_kernel, 146 KB LDS)", block),
    ("split-fp16 first layer 5x5 s2 (conv_s3_first_kernel)", lambda: conv(3, 32, 5, 2, 0, (0, 1, 0), res=False, h=369, w=1257, p=0)),
    ("torch elementwise add (memory traffic only)", fill),
]
VICTIMS = {
    "il": ("exact Winograd, interleaved in / out / residual", lambda: conv(32, 32, 3, 1, EX, (1, 1, 1))),
    "planar": ("exact Winograd, planar tensors", lambda: conv(32, 32, 3, 1, EX, None)),
    "il_in": ("exact Winograd, interleaved input only", lambda: conv(32, 32, 3, 1, EX, (1, 0, 0))),
    "il_out": ("exact Winograd, interleaved output only", lambda: conv(32, 32, 3, 1, EX, (0, 1, 0), res=False)),
    "il_nores": ("exact Winograd, interleaved in / out, no residual", lambda: conv(32, 32, 3, 1, EX, (1, 1, 0), res=False)),
    "mfma32": ("exact 3x3 stride 2 32->64 (conv_mfma_f32_kernel, v_mfma_f32_32x32x2_f32)", lambda: conv(32, 64, 3, 2, EX, None, res=False)),
    "split": ("split-fp16 3x3 32->32 interleaved (conv_s3_kernel)", lambda: conv(32, 32, 3, 1, 0, (1, 1, 1))),
}
vkey = os.environ.get("VICTIM", "il")
print("victim:", VICTIMS[vkey][0], flush=True)
only = os.environ.get("CASE")
for name, make in AGGRESSORS:
    if only and only not in name:
        continue
    victim, vy, vplan = VICTIMS[vkey][1]()
    vs = torch.cuda.Stream()
    ags = []
    if make:
        for _ in range(3):
            fn, _y, pl = make()
            ags.append((fn, torch.cuda.Stream(), pl))
    victim(vs.cuda_stream)
    torch.cuda.synchronize()
    ref = vy.clone()
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    words = torch.zeros((), dtype=torch.int64, device="cuda")
    for it in range(iters):
        for fn, s, _pl in ags:
            if _pl is None:
                with torch.cuda.stream(s):
                    fn(s)
            else:
                fn(s.cuda_stream)
        victim(vs.cuda_stream)
        with torch.cuda.stream(vs):
            d = (vy.view(torch.int32) != ref.view(torch.int32)).sum()
            bad += (d > 0)
            words += d
        if it % 128 == 127:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("aggressor: %-58s victim launches %d, deviating %d, differing words %d" % (name, iters, int(bad), int(words)), flush=True)
