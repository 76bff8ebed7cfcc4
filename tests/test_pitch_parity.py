"""Row-pitched planes (rt_conv_plan_set_pitch, rt_corr_softargmax_pitched): the executor keeps its internal
activations 128-byte aligned per row; the padding columns must be neither read as data nor written."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi
from test_ops_parity import T, near, rnd


def pitched(a, pitch, fill=np.nan):
    """(..., H, W) -> (..., H, pitch) with `fill` in the padding columns"""
    out = np.full(a.shape[:-1] + (pitch,), fill, np.float32)
    out[..., :a.shape[-1]] = a
    return out


CASES = [
    # cin, cout, h, w, k, stride, transposed, act, resid, batch, in_pitch, out_pitch
    (32, 32, 9, 37, 3, 1, False, capi.RT_ACT_ELU, True, 2, 64, 64),        # Winograd kernel
    (16, 8, 6, 21, 3, 1, False, capi.RT_ACT_NONE, False, 1, 32, 0),        # direct MFMA kernel, dense output
    (8, 40, 9, 33, 3, 2, False, capi.RT_ACT_ELU, False, 1, 0, 32),         # stride 2, dense input
    (3, 32, 11, 29, 5, 2, False, capi.RT_ACT_ELU, False, 2, 32, 32),
    (16, 24, 5, 9, 3, 2, True, capi.RT_ACT_ELU, True, 1, 32, 32),          # transposed: 4 phases, ZSlice offsets
    (32, 1, 4, 13, 3, 2, True, capi.RT_ACT_SIGMOID, False, 1, 32, 0),      # direct VALU kernel
]


@pytest.mark.parametrize("cin,cout,h,w,k,stride,tr,act,resid,batch,ip,op", CASES)
def test_conv2d_pitched(backend, cin, cout, h, w, k, stride, tr, act, resid, batch, ip, op):
    pad = k // 2 if not tr else 1
    x, b = rnd(batch, cin, h, w), rnd(cout)
    if tr:
        wt = rnd(cin, cout, k, k) * np.float32(1 / np.sqrt(cin * k * k))
        ref = O.deconv2d(T(x), T(wt), T(b), stride, pad)
    else:
        wt = rnd(cout, cin, k, k) * np.float32(1 / np.sqrt(cin * k * k))
        ref = O.conv2d(T(x), T(wt), T(b), stride, pad)
    res = rnd(*ref.shape) if resid else None
    if resid:
        ref = ref + T(res)
    ref = O.elu(ref) if act == capi.RT_ACT_ELU else (torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref)
    ref = ref.numpy()
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, k, stride, pad, act=act, has_residual=resid, transposed=tr)
    plan.set_pitch(ip, op)
    wo = ref.shape[-1]
    xin = pitched(x, ip) if ip else x
    rin = (pitched(res, op) if op else res) if resid else None
    y = backend.empty(ref.shape[:-1] + (op if op else wo,))
    plan.enqueue(backend.dev(xin), y, backend.dev(rin) if resid else None, batch)
    out = backend.host(y)
    near(out[..., :wo], ref, 2e-5)
    if op:
        assert np.isnan(out[..., wo:]).all(), "padding columns were written"
    plan.destroy()


def test_corr_softargmax_pitched(backend):
    n, c, h, w, d = 2, 16, 7, 45, 12
    l, r = rnd(n, c, h, w), rnd(n, c, h, w)
    ref = O.softargmax(O.corr_cost_volume(T(l), T(r), d), False).numpy()
    out = backend.empty((n, 1, h, 64))
    backend.klib.corr_softargmax_pitched(backend.dev(pitched(l, 64)), backend.dev(pitched(r, 64)), out, n, c, h, w, d,
                                         False, 64, 64)
    res = backend.host(out)
    near(res[..., :w], ref, 1e-4)
    assert np.isnan(res[..., w:]).all()
