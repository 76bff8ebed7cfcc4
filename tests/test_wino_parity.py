"""Parity of the Winograd F(2x2,3x3) MFMA kernel (conv_wino_f32_kernel) against the oracle and against the
direct implicit-GEMM kernel on the same inputs: image edges that cut tiles and pixel pairs (odd widths),
channel tails (Cout not a multiple of 4 / 16 / 32), several chunks and 32-channel blocks, batch > 1, and the
(D*C)-merged Conv3DPlugin form (reference lib/conv3d_plugin.cpp:187-216).  RT_CONV_NO_WINO selects the path
when the plan is created.  Runs on the SIMT emulator (CPU tier) and on the GPU (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi
from test_ops_parity import T, near, rnd

@pytest.fixture(autouse=True)
def _winograd_only(monkeypatch):
    """these tests are about the fp32 Winograd / direct-form kernels: keep small layers off the split-fp16 kernel
    (tests/test_split_parity.py covers that one)"""
    monkeypatch.setenv("RT_NO_S3", "1")


WINO_CASES = [
    # cin, cout, h, w, act, resid, batch
    (32, 32, 9, 33, capi.RT_ACT_ELU, True, 1),        # one pixel past a 32-wide tile
    (32, 32, 5, 63, capi.RT_ACT_ELU, True, 2),        # odd width: last pixel pair straddles the edge
    (8, 32, 3, 5, capi.RT_ACT_NONE, False, 1),        # image smaller than one tile
    (33, 32, 7, 64, capi.RT_ACT_ELU, False, 1),       # padded input channels (5 chunks)
    (16, 26, 6, 35, capi.RT_ACT_ELU, True, 1),        # Cout % 4 != 0: per-channel validity inside a lane
    (24, 40, 8, 31, capi.RT_ACT_SIGMOID, False, 2),   # two 32-channel blocks, second half empty past 40
    (64, 64, 10, 37, capi.RT_ACT_ELU, False, 1),
    (128, 128, 4, 18, capi.RT_ACT_ELU, False, 1),
]


def run_conv(backend, monkeypatch, no_wino, x, wt, b, act, res, batch):
    monkeypatch.setenv("RT_CONV_NO_WINO", "1" if no_wino else "0")
    cout, cin = wt.shape[:2]
    h, w = x.shape[2:]
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, 3, 1, 1, act=act, has_residual=res is not None)
    y = backend.empty((batch, cout, h, w))
    plan.enqueue(backend.dev(x), y, backend.dev(res) if res is not None else None, batch)
    out = backend.host(y).copy()
    plan.destroy()
    return out


@pytest.mark.parametrize("cin,cout,h,w,act,resid,batch", WINO_CASES)
def test_wino_conv2d(backend, monkeypatch, cin, cout, h, w, act, resid, batch):
    x, wt, b = rnd(batch, cin, h, w), rnd(cout, cin, 3, 3) * np.float32(1 / np.sqrt(cin * 9)), rnd(cout)
    res = rnd(batch, cout, h, w) if resid else None
    ref = O.conv2d(T(x), T(wt), T(b), 1, 1)
    if resid:
        ref = ref + T(res)
    ref = O.elu(ref) if act == capi.RT_ACT_ELU else (torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref)
    y_w = run_conv(backend, monkeypatch, False, x, wt, b, act, res, batch)
    y_d = run_conv(backend, monkeypatch, True, x, wt, b, act, res, batch)
    near(y_w, ref.numpy(), 2e-5)
    near(y_d, ref.numpy(), 2e-5)
    near(y_w, y_d, 2e-5)


def test_wino_large_values(backend, monkeypatch):
    """The transforms add before they multiply: outputs of a few hundred keep a relative error of ~1e-6."""
    x, wt, b = rnd(1, 32, 12, 40) * np.float32(30), rnd(32, 32, 3, 3), rnd(32)
    ref = O.conv2d(T(x), T(wt), T(b), 1, 1).numpy()
    y = run_conv(backend, monkeypatch, False, x, wt, b, capi.RT_ACT_NONE, None, 1)
    assert np.abs(ref).max() > 300
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max()


@pytest.mark.parametrize("out_dchw", [False, True])
def test_wino_conv3d(backend, monkeypatch, out_dchw):
    """3x3x3 stride-1 Conv3D with K = 32: depth taps ride the gather table, the 3x3 window is Winograd."""
    n, d, c, h, w, k = 1, 5, 8, 9, 35, 32
    x, wt, b = rnd(n, d, c, h, w), rnd(k, 3, c, 3, 3) * np.float32(1 / np.sqrt(27 * c)), rnd(k)
    ref = O.conv3d_tf(T(x), T(wt), T(b), (1, 1, 1), (1, 1, 1), (1, 1, 1))           # N K D H W
    if out_dchw:
        ref = O.transform(ref)
    ref = O.elu(ref).numpy()
    outs = []
    for no_wino in (False, True):
        monkeypatch.setenv("RT_CONV_NO_WINO", "1" if no_wino else "0")
        plan = backend.klib.conv3d_plan(wt, b, c, k, (d, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1),
                                        act=capi.RT_ACT_ELU, out_dchw=out_dchw)
        y = backend.empty((n,) + plan.out_dims)
        plan.enqueue(backend.dev(x), y, None, n)
        outs.append(backend.host(y).copy())
        plan.destroy()
    near(outs[0], ref, 2e-5)
    near(outs[1], ref, 2e-5)


def to_il(a, g):
    """(N, C, H, P) planar -> (N, C/g, H, P, g) channel-interleaved"""
    n, c, h, p = a.shape
    return np.ascontiguousarray(a.reshape(n, c // g, g, h, p).transpose(0, 1, 3, 4, 2))


def from_il(a):
    n, q, h, p, g = a.shape
    return a.transpose(0, 1, 4, 2, 3).reshape(n, q * g, h, p)


IL_CASES = [
    # cin, cout, h, w, act, resid, batch, pitch
    (32, 32, 9, 33, capi.RT_ACT_ELU, True, 1, 64),
    (32, 32, 5, 63, capi.RT_ACT_ELU, True, 2, 64),        # odd width
    (8, 32, 3, 5, capi.RT_ACT_NONE, False, 1, 32),        # image smaller than one tile
    (36, 28, 7, 64, capi.RT_ACT_ELU, True, 1, 64),        # channel counts that are multiples of 4 only; padded chunks
    (64, 64, 10, 37, capi.RT_ACT_SIGMOID, True, 1, 0),    # dense rows
    (128, 40, 4, 18, capi.RT_ACT_ELU, False, 2, 32),
]


@pytest.mark.parametrize("x_il,y_il,r_il", [(1, 1, 1), (1, 1, 0), (0, 1, 0), (1, 0, 1), (0, 0, 1), (1, 0, 0)])
@pytest.mark.parametrize("cin,cout,h,w,act,resid,batch,pitch", IL_CASES)
def test_wino_interleaved(backend, cin, cout, h, w, act, resid, batch, pitch, x_il, y_il, r_il):
    """fp32 Winograd kernel on channel-interleaved (C/4, H, pitch, 4) tensors, every mix with planar ones; results
    must equal the planar form bit for bit (same operands, same order of operations)"""
    if r_il and not resid:
        pytest.skip("no residual")
    from test_pitch_parity import pitched
    x, wt, b = rnd(batch, cin, h, w), rnd(cout, cin, 3, 3) * np.float32(1 / np.sqrt(cin * 9)), rnd(cout)
    res = rnd(batch, cout, h, w) if resid else None
    P = pitch or w
    outs = []
    for il in (False, True):
        xi, yi, ri = (x_il, y_il, r_il) if il else (0, 0, 0)
        plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, 3, 1, 1, act=act, has_residual=resid)
        if pitch:
            plan.set_pitch(pitch, pitch)
        if il:
            assert plan.supports_il8()
            plan.set_layouts(xi, yi, ri)
        lay = lambda a, f: to_il(a, 4) if f else a
        xin = backend.dev(lay(pitched(x, P), xi))                 # NaN in the padding columns
        rin = backend.dev(lay(pitched(res, P), ri)) if resid else None
        y = backend.empty((batch, cout // 4, h, P, 4) if yi else (batch, cout, h, P))
        plan.enqueue(xin, y, rin, batch)
        out = backend.host(y).copy()
        out = from_il(out) if yi else out
        if P > w:
            assert np.isnan(out[..., w:]).all(), "padding columns were written"
        outs.append(out[..., :w])
        plan.destroy()
    ref = O.conv2d(T(x), T(wt), T(b), 1, 1)
    if resid:
        ref = ref + T(res)
    ref = O.elu(ref) if act == capi.RT_ACT_ELU else (torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref)
    near(outs[1], ref.numpy(), 2e-5)
    assert np.array_equal(outs[0], outs[1])


def _il_fuzz_cases(n=20, seed=424242):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        cin, cout = int(rng.choice([4, 8, 12, 32, 36, 64, 100])), int(rng.choice([24, 28, 32, 40, 64, 96]))
        h, w = int(rng.integers(1, 13)), int(rng.integers(1, 75))
        resid = bool(rng.integers(0, 2))
        cases.append((cin, cout, h, w, resid, int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2)) if resid else 0,
                      int(rng.choice([capi.RT_ACT_NONE, capi.RT_ACT_ELU, capi.RT_ACT_SIGMOID])), int(rng.integers(1, 4)),
                      int(rng.choice([0, 32]))))
    return cases


@pytest.mark.parametrize("cin,cout,h,w,resid,x_il,y_il,r_il,act,batch,align", _il_fuzz_cases())
def test_wino_interleaved_fuzz(backend, cin, cout, h, w, resid, x_il, y_il, r_il, act, batch, align):
    """seeded random shapes (single pixels and rows, widths below a tile, channel counts that are multiples of 4 only),
    layout mixes, epilogues and row pitches for the fp32 Winograd kernel on channel-interleaved tensors"""
    from test_pitch_parity import pitched
    x, wt, b = rnd(batch, cin, h, w), rnd(cout, cin, 3, 3) * np.float32(1 / np.sqrt(cin * 9)), rnd(cout)
    res = rnd(batch, cout, h, w) if resid else None
    pitch = (w + align - 1) // align * align if align else 0
    P = pitch or w
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, 3, 1, 1, act=act, has_residual=resid)
    if pitch:
        plan.set_pitch(pitch, pitch)
    plan.set_layouts(x_il, y_il, r_il)
    lay = lambda a, f: to_il(a, 4) if f else a
    xin = backend.dev(lay(pitched(x, P), x_il))
    rin = backend.dev(lay(pitched(res, P), r_il)) if resid else None
    y = backend.empty((batch, cout // 4, h, P, 4) if y_il else (batch, cout, h, P))
    plan.enqueue(xin, y, rin, batch)
    out = backend.host(y).copy()
    out = from_il(out) if y_il else out
    plan.destroy()
    ref = O.conv2d(T(x), T(wt), T(b), 1, 1)
    if resid:
        ref = ref + T(res)
    ref = O.elu(ref) if act == capi.RT_ACT_ELU else (torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref)
    near(out[..., :w], ref.numpy(), 2e-5)
    if P > w:
        assert np.isnan(out[..., w:]).all(), "padding columns were written"
