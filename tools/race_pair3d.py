#!/usr/bin/env python3
"""Cross-kernel determinism of the half2 3-D kernels (same protocol as tools/race_pair.py): a victim plan back to back on one stream, an
aggressor plan on three others, every victim output compared bit for bit with its first result.  The kernels mix MFMA shapes
(v_mfma_f32_16x16x32_f16 in deconv3d_s2_il_kernel, v_mfma_f32_32x32x16_f16 everywhere else): profiles/r04_race.txt shows what a mix of
fp32 16x16x4 and fp16 32x32x16 MFMAs on one CU does.     python tools/race_pair3d.py [iterations] [--f32: the fp32 engines' 3-D kernels]"""
import os
import sys
os.environ.setdefault("RT_DEV_KNOBS", "1")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redtail_amd import capi  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000
k = capi.KernelLib()
rng = np.random.default_rng(3)
h16 = lambda a: np.ascontiguousarray(a.astype(np.float16))
F16 = capi.RT_F16


def small_il():          # last layer: (32/8, 12, 41, 129, 8) fp16 -> (24, 1, 81, 257) fp32 on v_mfma_f32_16x16x32_f16
    K, C, dy, hy, wy = 32, 1, 12, 41, 129
    w = h16(rng.standard_normal((K, 3, C, 3, 3)) / np.sqrt(27 * K / 8))
    p = k.conv3d_plan(w, h16(rng.standard_normal(C)), C, K, (2 * dy + 1, 2 * hy - 1, 2 * wy - 1), (3, 3, 3), (2, 2, 2), (0, 1, 1), (0, 1, 1),
                      dtype=F16, transposed_in_dims=(dy, hy, wy), out_depth=2 * dy)
    p.set_io_types(F16, capi.RT_F32)
    p.set_layouts(1, 0, 0)
    x = torch.randn(1, K // 8, dy, hy, wy, 8, device="cuda").half()
    y = torch.zeros(1, 2 * dy, C, 2 * hy - 1, 2 * wy - 1, device="cuda")
    return (lambda s: p.enqueue(x, y, None, 1, stream=s)), y, p


def conv3d(r4):          # 32 -> 32 @ (12, 81, 257) interleaved fp16: conv_f16r4_kernel (r4 = 1) / conv_f16mma_kernel (0), v_mfma_f32_32x32x16_f16
    os.environ["RT_F16_R4"] = str(r4)
    c, kk, d, h, w = 32, 32, 12, 81, 257
    wt = h16(rng.standard_normal((kk, 3, c, 3, 3)) / np.sqrt(27 * c))
    p = k.conv3d_plan(wt, h16(rng.standard_normal(kk)), c, kk, (d, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU, out_dchw=True, dtype=F16)
    p.set_io_types(F16, F16)
    p.set_layouts(1, 1, 0)
    x = torch.randn(1, d, c // 8, h, w, 8, device="cuda").half()
    y = torch.zeros(1, d, kk // 8, h, w, 8, device="cuda").half()
    fn = lambda s: p.enqueue(x, y, None, 1, stream=s)
    fn(None)                        # the knob is read at the plan's first launch
    torch.cuda.synchronize()
    return fn, y, p


def deconv():            # 64 -> 32 transposed, interleaved: conv_f16mma_kernel<2,2,1> phases
    K, C, dy, hy, wy = 64, 32, 6, 41, 129
    w = h16(rng.standard_normal((K, 3, C, 3, 3)) / np.sqrt(27 * K / 8))
    p = k.conv3d_plan(w, h16(rng.standard_normal(C)), C, K, (2 * dy + 1, 2 * hy - 1, 2 * wy - 1), (3, 3, 3), (2, 2, 2), (0, 1, 1), (0, 1, 1),
                      act=capi.RT_ACT_ELU, out_dchw=True, dtype=F16, transposed_in_dims=(dy, hy, wy), out_depth=2 * dy)
    p.set_io_types(F16, F16)
    p.set_layouts(1, 1, 0)
    x = torch.randn(1, K // 8, dy, hy, wy, 8, device="cuda").half()
    y = torch.zeros(1, C // 8, 2 * dy, 2 * hy - 1, 2 * wy - 1, 8, device="cuda").half()
    return (lambda s: p.enqueue(x, y, None, 1, stream=s)), y, p


def deconv_f32(il_out):  # fp32 engines: 64 -> 32 transposed, four phases per workgroup in split form (deconv_s3p_kernel), planar or interleaved output
    K, C, dy, hy, wy = 64, 32, 6, 41, 129
    w = (rng.standard_normal((K, 3, C, 3, 3)) / np.sqrt(27 * K / 8)).astype(np.float32)
    p = k.conv3d_plan(w, rng.standard_normal(C).astype(np.float32), C, K, (2 * dy + 1, 2 * hy - 1, 2 * wy - 1), (3, 3, 3), (2, 2, 2), (0, 1, 1), (0, 1, 1),
                      act=capi.RT_ACT_ELU, out_dchw=True, transposed_in_dims=(dy, hy, wy), out_depth=2 * dy)
    if il_out:
        p.set_layouts(0, 1, 0)
    x = torch.randn(1, K, dy, hy, wy, device="cuda")
    y = torch.zeros(1, C, 2 * dy, 2 * hy - 1, 2 * wy - 1, device="cuda")
    return (lambda s: p.enqueue(x, y, None, 1, stream=s)), y, p


def small_il4():         # fp32 engines: last layer from (32/4, 12, 41, 129, 4) fp32, split form on v_mfma_f32_16x16x32_f16
    K, C, dy, hy, wy = 32, 1, 12, 41, 129
    w = (rng.standard_normal((K, 3, C, 3, 3)) / np.sqrt(27 * K / 8)).astype(np.float32)
    p = k.conv3d_plan(w, rng.standard_normal(C).astype(np.float32), C, K, (2 * dy + 1, 2 * hy - 1, 2 * wy - 1), (3, 3, 3), (2, 2, 2), (0, 1, 1), (0, 1, 1),
                      transposed_in_dims=(dy, hy, wy), out_depth=2 * dy)
    p.set_layouts(1, 0, 0)
    x = torch.randn(1, K // 4, dy, hy, wy, 4, device="cuda")
    y = torch.zeros(1, 2 * dy, C, 2 * hy - 1, 2 * wy - 1, device="cuda")
    return (lambda s: p.enqueue(x, y, None, 1, stream=s)), y, p


def conv3d_f32():        # fp32 engines: 32 -> 32 @ (12, 81, 257) on (D, C/4, H, W, 4) tensors: conv_s3_kernel, 3-term split
    c, kk, d, h, w = 32, 32, 12, 81, 257
    wt = (rng.standard_normal((kk, 3, c, 3, 3)) / np.sqrt(27 * c)).astype(np.float32)
    p = k.conv3d_plan(wt, rng.standard_normal(kk).astype(np.float32), c, kk, (d, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU, out_dchw=True)
    p.set_layouts(1, 1, 0)
    x = torch.randn(1, d, c // 4, h, w, 4, device="cuda")
    y = torch.zeros(1, d, kk // 4, h, w, 4, device="cuda")
    return (lambda s: p.enqueue(x, y, None, 1, stream=s)), y, p


KERNELS = {"last layer (16x16x32 MFMA)": small_il, "Conv3D, 4 rows per wave": lambda: conv3d(1), "Conv3D, 4 x 32 tiles": lambda: conv3d(0), "Conv3DTranspose phases": deconv}
if "--f32" in sys.argv:  # the fp32 engines' 3-D kernels (round 4), each beside each and beside an fp16-operand Conv3D
    KERNELS = {"fp32 Conv3DTranspose, 4 phases": lambda: deconv_f32(False), "... interleaved output": lambda: deconv_f32(True),
               "fp32 last layer (split, 16x16x32)": small_il4, "fp32 Conv3D (split, interleaved)": conv3d_f32, "fp16 Conv3D, 4 rows per wave": lambda: conv3d(1)}
for vname, vmake in KERNELS.items():
    for aname, amake in KERNELS.items():
        victim, vy, vp = vmake()
        ags = [amake() + (torch.cuda.Stream(),) for _ in range(3)]
        vs = torch.cuda.Stream()
        victim(vs.cuda_stream)
        torch.cuda.synchronize()
        ref = vy.clone()
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        for it in range(iters):
            for fn, _y, _p, s in ags:
                fn(s.cuda_stream)
            victim(vs.cuda_stream)
            with torch.cuda.stream(vs):
                bad += (vy.view(torch.int16 if vy.dtype == torch.float16 else torch.int32) != ref.view(torch.int16 if vy.dtype == torch.float16 else torch.int32)).any()
            if it % 64 == 63:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        print("victim %-28s aggressor %-28s launches %d deviating %d" % (vname, aname, iters, int(bad)), flush=True)
