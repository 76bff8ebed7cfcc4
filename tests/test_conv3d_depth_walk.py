"""conv_f16dw_kernel (round 5): Conv3D 3x3x3 stride 1 between channel-interleaved fp16 tensors with the workgroup walking down the depth
axis -- input slices through LDS once, three live accumulator sets, weights resident (C <= 32) or streamed (C >= 48).  Against the oracle
(lib/conv3d_plugin.cpp:187-216 semantics on fp16-rounded operands, fp32 accumulation, one rounding of the output) and BIT FOR BIT against
conv_f16r4_kernel, whose summation order per output element it keeps -- whatever the depth segmentation."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi
from test_ops_parity import T, rnd
from test_deconv3d_half2 import q16, h16, dev16, empty, host, il_cm, un_il_cm, il_dm, un_il_dm

DW_CASES = [
    # c, k, d, h, w, channel-major output, residual (0 none, 1 interleaved), depth segments (0 = chosen by the plan)
    (16, 32, 5, 13, 37, False, 0, 0),      # one chunk per slice (resident), two 12-row tiles (the second with one row), two column tiles
    (32, 32, 6, 14, 33, False, 1, 2),      # two chunks per slice (resident), skip tensor, 2 x 2 tiles; two depth segments
    (32, 64, 4, 12, 40, True, 0, 4),       # two blocks of output channels, channel-major (K/8, D, H, W, 8) output, one slice per segment
    (48, 24, 3, 9, 70, False, 0, 0),       # three chunks per slice (streamed weights), Cout % 32 != 0, three column tiles: an odd tile count
    (64, 32, 5, 13, 40, False, 1, 2),      # four chunks per slice (streamed), skip tensor, 2 x 2 tiles (an odd row), segments of 3 + 2
    (16, 8, 1, 5, 9, True, 0, 0),          # a single depth slice
    (32, 32, 2, 3, 34, False, 0, 2),       # two slices, two segments of one
]


@pytest.mark.parametrize("c,k,d,h,w,cm,resid,nseg", DW_CASES)
def test_conv3d_depth_walk(backend, monkeypatch, c, k, d, h, w, cm, resid, nseg):
    n = 2
    x = q16(rnd(n, d, c, h, w))
    wt, b = q16(rnd(k, 3, c, 3, 3) * np.float32(1 / np.sqrt(27 * c))), q16(rnd(k))
    ref = O.conv3d_tf(T(x).double(), T(wt).double(), T(b).double(), (1, 1, 1), (1, 1, 1), (1, 1, 1))      # (N, K, D, H, W)
    if not cm:
        ref = O.transform(ref)                                                                        # (N, D, K, H, W)
    res = q16(rnd(*ref.shape)) if resid else None
    if resid:
        ref = ref + T(res).double()
    ref = O.elu(ref).numpy()
    outs = []
    for dw in ("1", "0"):
        monkeypatch.setenv("RT_F16_DW", dw)
        monkeypatch.setenv("RT_F16_R4", "1")
        monkeypatch.setenv("RT_DW_NSEG", str(nseg))
        plan = backend.klib.conv3d_plan(h16(wt), h16(b), c, k, (d, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU,
                                        out_dchw=not cm, has_residual=bool(resid), dtype=capi.RT_F16)
        plan.set_io_types(capi.RT_F16, capi.RT_F16)
        plan.set_layouts(1, 1, 0)
        oshape = (n, k // 8, d, h, w, 8) if cm else (n, d, k // 8, h, w, 8)
        out = empty(backend, oshape, True)
        rin = None
        if resid:
            rin = dev16(backend, il_cm(res) if cm else il_dm(res))
            plan.set_layouts(1, 1, 1) if plan.il_caps() & 4 else pytest.skip("no interleaved residual for this plan")
        plan.enqueue(dev16(backend, il_dm(x)), out, rin, n)
        got = host(backend, out)
        assert not np.isnan(got).any()
        outs.append(un_il_cm(got) if cm else un_il_dm(got))
        plan.destroy()
    tol = 2e-3 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[0] - ref).max() <= tol, np.abs(outs[0] - ref).max()
    assert np.abs(outs[1] - ref).max() <= tol
    assert np.array_equal(outs[0], outs[1]), np.abs(outs[0] - outs[1]).max()      # same summation order as the per-slice kernel


def test_conv3d_depth_walk_without_activation_and_any_segmentation(backend, monkeypatch):
    """no activation; every segmentation of the depth axis gives the same bits"""
    n, c, k, d, h, w = 1, 32, 32, 9, 7, 21
    x = q16(rnd(n, d, c, h, w))
    wt, b = q16(rnd(k, 3, c, 3, 3) * np.float32(1 / np.sqrt(27 * c))), q16(rnd(k))
    ref = O.transform(O.conv3d_tf(T(x).double(), T(wt).double(), T(b).double(), (1, 1, 1), (1, 1, 1), (1, 1, 1))).numpy()
    outs = []
    for nseg in (1, 2, 3, 5, 9):
        monkeypatch.setenv("RT_F16_DW", "1")
        monkeypatch.setenv("RT_DW_NSEG", str(nseg))
        plan = backend.klib.conv3d_plan(h16(wt), h16(b), c, k, (d, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_NONE,
                                        out_dchw=True, dtype=capi.RT_F16)
        plan.set_io_types(capi.RT_F16, capi.RT_F16)
        plan.set_layouts(1, 1, 0)
        out = empty(backend, (n, d, k // 8, h, w, 8), True)
        plan.enqueue(dev16(backend, il_dm(x)), out, None, n)
        outs.append(un_il_dm(host(backend, out)))
        plan.destroy()
    assert np.abs(outs[0] - ref).max() <= 2e-3 * max(1.0, float(np.abs(ref).max()))
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
