"""Parity of the split-fp16 convolution kernels (conv_split.hip.h: fp32 tensors, three fp16 MFMAs per product term,
fp32 accumulation) against an fp64 evaluation of the same convolution, next to the fp32 fmaf-chain kernels they replace:
the split form must be as accurate as fp32 arithmetic, not merely inside the 1e-3 budget.  Persistent tile loop
(several tiles per workgroup, both LDS buffers), image edges, channel tails, every mix of planar and channel-interleaved
tensors, batch > 1.  CPU tier: SIMT emulator; GPU tier: -m gpu."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi
from test_ops_parity import rnd
from test_pitch_parity import pitched
from test_wino_parity import from_il, to_il


def ref64(x, wt, b, res, act):
    y = O.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), 1, 1)
    if res is not None:
        y = y + torch.from_numpy(res).double()
    y = O.elu(y) if act == capi.RT_ACT_ELU else (torch.sigmoid(y) if act == capi.RT_ACT_SIGMOID else y)
    return y.numpy()


def run(backend, x, wt, b, res, act, batch, pitch=0, x_il=0, y_il=0, r_il=0):
    cout, cin = wt.shape[:2]
    h, w = x.shape[2:]
    P = pitch or w
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, 3, 1, 1, act=act, has_residual=res is not None)
    if pitch:
        plan.set_pitch(pitch, pitch)
    if x_il or y_il or r_il:
        plan.set_layouts(x_il, y_il, r_il)
    lay = lambda a, f: to_il(a, 4) if f else a
    xin = backend.dev(lay(pitched(x, P), x_il))                       # NaN in the padding columns
    rin = backend.dev(lay(pitched(res, P), r_il)) if res is not None else None
    y = backend.empty((batch, cout // 4, h, P, 4) if y_il else (batch, cout, h, P))
    plan.enqueue(xin, y, rin, batch)
    out = backend.host(y).copy()
    out = from_il(out) if y_il else out
    plan.destroy()
    if P > w:
        assert np.isnan(out[..., w:]).all(), "padding columns were written"
    return out[..., :w]


CASES = [
    # cin, cout, h, w, act, resid, batch, pitch, (x_il, y_il, r_il)
    (32, 32, 19, 70, capi.RT_ACT_ELU, True, 2, 96, (1, 1, 1)),     # the resblock layer: 3 x 3 tiles per image, 2 images
    (32, 32, 9, 33, capi.RT_ACT_ELU, True, 1, 64, (1, 1, 0)),      # one row / one pixel past a tile
    (32, 32, 8, 32, capi.RT_ACT_NONE, False, 1, 0, (0, 0, 0)),     # exactly one tile, planar, dense rows
    (32, 32, 17, 45, capi.RT_ACT_ELU, False, 1, 64, (1, 0, 0)),    # encoder2D_out: interleaved in, planar out
    (32, 32, 11, 37, capi.RT_ACT_ELU, True, 1, 64, (0, 1, 1)),     # resblock1_conv1-like: planar in, interleaved out
    (16, 24, 10, 40, capi.RT_ACT_SIGMOID, True, 2, 0, (1, 1, 1)),  # one chunk only, Cout < 32
    (20, 30, 7, 35, capi.RT_ACT_ELU, True, 1, 0, (1, 0, 0)),       # Cout % 4 != 0 -> planar output / residual
    (13, 9, 5, 9, capi.RT_ACT_NONE, False, 3, 0, (0, 0, 0)),       # odd channel counts, image smaller than a tile
    (8, 8, 3, 5, capi.RT_ACT_ELU, False, 1, 32, (1, 1, 0)),        # NVTiny's 2-D encoder width
    (3, 32, 1, 1, capi.RT_ACT_NONE, False, 2, 0, (0, 1, 0)),       # single pixel
]


@pytest.mark.parametrize("grid", [0, 3])
@pytest.mark.parametrize("cin,cout,h,w,act,resid,batch,pitch,il", CASES)
def test_split_conv2d(backend, monkeypatch, cin, cout, h, w, act, resid, batch, pitch, il, grid):
    """grid = 3: three workgroups walk all tiles (persistent loop, double-buffered patch, prefetch across tiles and
    images); grid = 0: the default one-workgroup-per-CU launch"""
    if not backend.klib.has_experimental():
        pytest.skip("conv_s3p_kernel is compiled with RT_EXPERIMENTAL only (the emulator build of the CPU tier)")
    monkeypatch.setenv("RT_S3P", "1")
    if grid:
        monkeypatch.setenv("RT_S3P_GRID", str(grid))
    x, wt, b = rnd(batch, cin, h, w), rnd(cout, cin, 3, 3) * np.float32(1 / np.sqrt(cin * 9)), rnd(cout)
    res = rnd(batch, cout, h, w) if resid else None
    ref = ref64(x, wt, b, res, act)
    out = run(backend, x, wt, b, res, act, batch, pitch, *il)
    err = np.abs(out - ref).max()
    assert err <= 5e-6, err
    # the exact-fp32 kernels on the same input: the split form is not allowed to be (noticeably) less accurate
    monkeypatch.setenv("RT_CONV_EXACT_FP32", "1")
    exact = run(backend, x, wt, b, res, act, batch, pitch, 0, 0, 0)
    assert err <= max(3.0 * np.abs(exact - ref).max(), 1.5e-6), (err, np.abs(exact - ref).max())


def test_split_is_fp32_accurate_on_large_and_tiny_values(backend):
    """dynamic range: activations of a few hundred next to ones of 1e-4 (low parts are pre-scaled, so nothing relies on
    fp16 subnormals); error relative to the magnitude of the sum stays in the fp32-roundoff class"""
    rng = np.random.default_rng(5)
    x = rnd(1, 32, 12, 40)
    x *= np.where(rng.random(x.shape) < 0.5, np.float32(300.0), np.float32(1e-4))
    wt, b = rnd(32, 32, 3, 3) * np.float32(0.1), rnd(32)
    ref = ref64(x, wt, b, None, capi.RT_ACT_NONE)
    out = run(backend, x, wt, b, None, capi.RT_ACT_NONE, 1)
    mag = O.conv2d(torch.from_numpy(np.abs(x)).double(), torch.from_numpy(np.abs(wt)).double(), None, 1, 1).numpy()
    assert np.abs(ref).max() > 100
    assert (np.abs(out - ref) / mag).max() <= 4e-7          # ~ 2^-22 + accumulation roundoff, relative to sum |x w|


def test_split_overflow_is_loud(backend):
    """outside the fp16 range of the high part (|x| >= 65520) the result is inf / NaN, never a silently wrong number"""
    x = rnd(1, 32, 8, 32)
    x[0, 3, 4, 5] = 7.0e4
    wt, b = rnd(32, 32, 3, 3), rnd(32)
    out = run(backend, x, wt, b, None, capi.RT_ACT_NONE, 1)
    assert not np.isfinite(out[0, :, 3:6, 4:7]).all()


# ---- the general kernel (conv_s3_kernel): other windows, more channels, transposed phases ------------------------------
def ref64_g(x, wt, b, res, act, stride, tr):
    X, Wt, B = torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double()
    y = O.deconv2d(X, Wt, B, stride, 1) if tr else O.conv2d(X, Wt, B, stride, wt.shape[-1] // 2)
    if res is not None:
        y = y + torch.from_numpy(res).double()
    y = O.elu(y) if act == capi.RT_ACT_ELU else (torch.sigmoid(y) if act == capi.RT_ACT_SIGMOID else y)
    return y.numpy()


G_CASES = [
    # cin, cout, h, w, k, stride, transposed, act, resid, batch, (x_il, y_il, r_il)
    (33, 32, 9, 37, 3, 1, False, capi.RT_ACT_ELU, False, 1, (0, 1, 0)),      # conv2D_1: 33 channels -> 3 chunks, 15 of them padding
    (64, 64, 10, 37, 3, 1, False, capi.RT_ACT_ELU, True, 2, (1, 1, 1)),      # conv2D_4/5: two 32-channel blocks, 4 chunks
    (128, 128, 5, 18, 3, 1, False, capi.RT_ACT_ELU, False, 1, (1, 1, 0)),    # conv2D_7/8
    (32, 64, 11, 37, 3, 2, False, capi.RT_ACT_ELU, False, 2, (1, 1, 0)),     # conv2D_3ds: stride 2, even / odd patch columns
    (64, 128, 8, 66, 3, 2, False, capi.RT_ACT_ELU, False, 1, (0, 0, 0)),     # conv2D_6ds, planar
    (20, 36, 9, 35, 3, 2, False, capi.RT_ACT_NONE, True, 1, (1, 1, 1)),      # channel tails, stride 2 with residual
    (128, 64, 5, 9, 3, 2, True, capi.RT_ACT_ELU, True, 2, (1, 1, 1)),        # deconv2D_1: 4 phases in one launch, interleaved everywhere
    (64, 32, 6, 13, 3, 2, True, capi.RT_ACT_ELU, True, 1, (1, 0, 0)),        # deconv2D_2: planar output + residual
    (24, 20, 7, 33, 3, 2, True, capi.RT_ACT_SIGMOID, False, 1, (0, 1, 0)),   # transposed, odd sizes, interleaved output only
    (16, 40, 6, 35, 1, 1, False, capi.RT_ACT_NONE, False, 2, (1, 1, 0)),     # 1x1
]


@pytest.mark.parametrize("cin,cout,h,w,k,stride,tr,act,resid,batch,il", G_CASES)
def test_split_general(backend, monkeypatch, cin, cout, h, w, k, stride, tr, act, resid, batch, il):
    x, b = rnd(batch, cin, h, w), rnd(cout)
    wt = rnd(*((cin, cout, k, k) if tr else (cout, cin, k, k))) * np.float32(1 / np.sqrt(cin * k * k / (stride * stride if tr else 1)))
    shape = ref64_g(x, wt, b, None, capi.RT_ACT_NONE, stride, tr).shape
    res = rnd(*shape) if resid else None
    ref = ref64_g(x, wt, b, res, act, stride, tr)
    ho, wo = shape[-2:]
    ip, op = (w + 31) // 32 * 32, (wo + 31) // 32 * 32
    outs = []
    # the interleaved form once as shipped (low-resolution layers split their contraction over wave groups: another summation
    # order) and once with that switched off, where layouts change addressing only
    for (x_il, y_il, r_il), ksplit in ((il, None), (il, "0"), ((0, 0, 0), "0"), ((0, 0, 0), None)):
        if ksplit is None:
            monkeypatch.delenv("RT_S3_KSPLIT", raising=False)
        else:
            monkeypatch.setenv("RT_S3_KSPLIT", ksplit)
        plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, k, stride, 1 if tr else k // 2, act=act, has_residual=resid, transposed=tr)
        plan.set_pitch(ip, op)
        if x_il or y_il or r_il:
            assert plan.il_caps() == (1 if cin % 4 == 0 else 16) | (6 if cout % 4 == 0 else 0)      # 16: interleaved input with padded channels
            plan.set_layouts(x_il, y_il, r_il)
        lay = lambda a, f: to_il(a, 4) if f else a
        xin = backend.dev(lay(pitched(x, ip), x_il))
        rin = backend.dev(lay(pitched(res, op), r_il)) if resid else None
        y = backend.empty((batch, cout // 4, ho, op, 4) if y_il else (batch, cout, ho, op))
        plan.enqueue(xin, y, rin, batch)
        out = backend.host(y).copy()
        out = from_il(out) if y_il else out
        plan.destroy()
        if op > wo:
            assert np.isnan(out[..., wo:]).all(), "padding columns were written"
        outs.append(out[..., :wo])
    for o in outs:
        assert np.abs(o - ref).max() <= 4e-7 * np.sqrt(cin * k * k) + 2e-6, np.abs(o - ref).max()
    assert np.array_equal(outs[1], outs[2])                       # layouts change addressing only


@pytest.mark.parametrize("ksplit", ["2", "4"])
@pytest.mark.parametrize("cin,cout,h,w,k,stride,tr,resid,batch", [
    (64, 64, 10, 37, 3, 1, False, True, 2),        # 4 chunks: every group one (KS = 4) or two (KS = 2) of them
    (80, 32, 9, 33, 3, 1, False, False, 1),        # 5 chunks: groups with 2 / 1 / 1 / 1 chunks
    (32, 64, 11, 37, 3, 2, False, False, 1),       # stride 2: at most two groups (a request for 4 is clamped)
    (128, 64, 5, 9, 3, 2, True, True, 1),          # transposed: 2 x 2 phase windows with tap masks, 8 chunks
    (16, 40, 6, 35, 1, 1, False, False, 2),        # 1 x 1 window, ONE chunk: groups 1 .. KS-1 contribute zeros
])
def test_split_k_forced(backend, monkeypatch, cin, cout, h, w, k, stride, tr, resid, batch, ksplit):
    """RT_S3_KSPLIT forces the number of contraction groups of conv_s3_kernel (the launch clamps it to what the instantiation's LDS and
    register budget allow): uneven chunk counts, fewer chunks than groups, stride 2, transposed phases -- against the fp64 reference and
    within rounding of the unsplit launch"""
    x, b = rnd(batch, cin, h, w), rnd(cout)
    wt = rnd(*((cin, cout, k, k) if tr else (cout, cin, k, k))) * np.float32(1 / np.sqrt(cin * k * k / (stride * stride if tr else 1)))
    shape = ref64_g(x, wt, b, None, capi.RT_ACT_NONE, stride, tr).shape
    res = rnd(*shape) if resid else None
    ref = ref64_g(x, wt, b, res, capi.RT_ACT_ELU, stride, tr)
    ho, wo = shape[-2:]
    ip, op = (w + 31) // 32 * 32, (wo + 31) // 32 * 32
    outs = []
    for ks in (ksplit, "0"):
        monkeypatch.setenv("RT_S3_KSPLIT", ks)
        plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, k, stride, 1 if tr else k // 2, act=capi.RT_ACT_ELU, has_residual=resid, transposed=tr)
        plan.set_pitch(ip, op)
        plan.set_layouts(1, 1, 1 if resid else 0)
        y = backend.empty((batch, cout // 4, ho, op, 4))
        plan.enqueue(backend.dev(to_il(pitched(x, ip), 4)), y, backend.dev(to_il(pitched(res, op), 4)) if resid else None, batch)
        out = from_il(backend.host(y).copy())
        plan.destroy()
        assert np.isnan(out[..., wo:]).all() if op > wo else True, "padding columns were written"
        outs.append(out[..., :wo])
    tol = 4e-7 * np.sqrt(cin * k * k) + 2e-6
    assert np.abs(outs[0] - ref).max() <= tol, np.abs(outs[0] - ref).max()
    assert np.abs(outs[0] - outs[1]).max() <= tol


@pytest.mark.parametrize("cin,cout,h,w,batch,pitch,y_il", [
    (3, 32, 21, 77, 2, 64, 1),        # ResNet-18 2D / NVSmall first layer
    (3, 32, 9, 129, 1, 96, 0),
    (3, 8, 13, 41, 2, 0, 1),          # NVTiny: 8 feature channels
    (1, 32, 5, 7, 1, 0, 0),           # grey image, smaller than a tile
    (3, 40, 8, 66, 1, 32, 1),         # two 32-channel blocks, even sizes
])
def test_split_first_layer(backend, cin, cout, h, w, batch, pitch, y_il):
    """conv_s3_first_kernel: 5x5 stride 2 on <= 3 channels, window row = contraction index, against fp64"""
    x, wt, b = rnd(batch, cin, h, w), rnd(cout, cin, 5, 5) * np.float32(1 / np.sqrt(cin * 25)), rnd(cout)
    ref = O.elu(O.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), 2, 2)).numpy()
    ho, wo = ref.shape[-2:]
    P = (wo + pitch - 1) // pitch * pitch if pitch else wo
    outs = []
    for il in ((0, 1) if y_il else (0,)):
        plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, 5, 2, 2, act=capi.RT_ACT_ELU)
        if pitch:
            plan.set_pitch(0, P)
        assert plan.il_caps() == (2 if cout % 4 == 0 else 0)
        if il:
            plan.set_layouts(0, 1)
        y = backend.empty((batch, cout // 4, ho, P, 4) if il else (batch, cout, ho, P))
        plan.enqueue(backend.dev(x), y, None, batch)
        out = backend.host(y).copy()
        out = from_il(out) if il else out
        plan.destroy()
        if P > wo:
            assert np.isnan(out[..., wo:]).all(), "padding columns were written"
        outs.append(out[..., :wo])
    assert np.abs(outs[0] - ref).max() <= 3e-6, np.abs(outs[0] - ref).max()
    if y_il:
        assert np.array_equal(outs[0], outs[1])


# ---- fused residual block (conv_s3rb_kernel) ------------------------------------------------------------------------------
RB_CASES = [
    # c, cmid, h, w, batch, pitch, (x_il, y_il)
    (32, 32, 19, 70, 2, 96, (1, 1)),      # the tower block: 3 x 3 tiles per image
    (32, 32, 8, 32, 1, 0, (0, 0)),        # exactly one tile, planar, dense
    (32, 32, 9, 33, 1, 64, (1, 0)),       # one row / pixel past a tile
    (16, 24, 11, 37, 2, 64, (0, 1)),      # Cmid != C, one chunk of input channels
    (20, 30, 7, 35, 1, 0, (1, 1)),        # channel tails (multiples of 4 only where interleaved)
    (13, 9, 5, 9, 3, 0, (0, 0)),          # odd channel counts, image smaller than a tile
    (8, 8, 3, 5, 1, 32, (1, 1)),
    (4, 4, 1, 1, 2, 0, (0, 0)),           # single pixel
    # 32 -> 32 -> 32 on interleaved tensors: the streaming kernel (conv_rbs.hip.h), strips of 30 columns x segments of 16 rows
    (32, 32, 16, 30, 1, 32, (1, 1)),      # exactly one strip and one segment
    (32, 32, 17, 31, 1, 0, (1, 1)),       # one row / one column past them, dense odd rows
    (32, 32, 37, 61, 2, 64, (1, 1)),      # three segments (16 + 16 + 5), three strips (30 + 30 + 1), two images
    (32, 32, 5, 100, 1, 128, (1, 1)),     # shorter than a segment
    (32, 32, 1, 1, 1, 0, (1, 1)),         # single pixel
    (32, 32, 33, 7, 1, 32, (1, 1)),       # narrower than a strip
]


@pytest.mark.parametrize("c,cmid,h,w,batch,pitch,il", RB_CASES)
def test_split_resblock(backend, c, cmid, h, w, batch, pitch, il):
    """whole residual block in one launch against an fp64 evaluation of the two layers, and against the same block run
    layer by layer through the general split kernel"""
    if not (c == 32 and cmid == 32 and il == (1, 1)) and not backend.klib.has_experimental():
        pytest.skip("the per-tile form (conv_s3rb_kernel) is compiled with RT_EXPERIMENTAL only (the emulator build of the CPU tier)")
    x = rnd(batch, c, h, w)
    w1, b1 = rnd(cmid, c, 3, 3) * np.float32(1 / np.sqrt(c * 9)), rnd(cmid)
    w2, b2 = rnd(c, cmid, 3, 3) * np.float32(1 / np.sqrt(cmid * 9)), rnd(c)
    X = torch.from_numpy(x).double()
    t = O.elu(O.conv2d(X, torch.from_numpy(w1).double(), torch.from_numpy(b1).double(), 1, 1))
    ref = O.elu(O.conv2d(t, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), 1, 1) + X).numpy()
    P = pitch or w
    x_il, y_il = il
    lay = lambda a, f: to_il(a, 4) if f else a
    plan = backend.klib.resblock_plan(w1, b1, w2, b2, c, cmid, h, w)
    if pitch:
        plan.set_pitch(pitch, pitch)
    assert plan.il_caps() == (5 if c % 4 == 0 else 0) | (2 if c % 4 == 0 else 0)
    if x_il or y_il:
        plan.set_layouts(x_il, y_il, x_il)
    xin = backend.dev(lay(pitched(x, P), x_il))                        # NaN in the padding columns
    y = backend.empty((batch, c // 4, h, P, 4) if y_il else (batch, c, h, P))
    plan.enqueue(xin, y, xin, batch)
    out = backend.host(y).copy()
    out = from_il(out) if y_il else out
    plan.destroy()
    if P > w:
        assert np.isnan(out[..., w:]).all(), "padding columns were written"
    out = out[..., :w]
    assert np.abs(out - ref).max() <= 6e-6, np.abs(out - ref).max()
    # layer by layer (conv_s3_kernel twice): the fused form keeps the intermediate in LDS as the same hi / lo pair the second
    # layer would have split it into; what differs is the order of the fp32 accumulation (taps x chunks)
    t2 = run(backend, x, w1, b1, None, capi.RT_ACT_ELU, batch)
    two = run(backend, t2, w2, b2, x, capi.RT_ACT_ELU, batch)
    assert np.abs(out - two).max() <= 6e-6, np.abs(out - two).max()


# ---- the tower block on PRE-SPLIT tensors (conv_rbd.hip.h) ---------------------------------------------------------------------
def to_split(a):
    """(N, C, H, P) fp32 -> the pre-split tensor (N, C/8, H, P, [8 hi | 8 lo]) as an fp32-typed array (N, C/8, H, P, 8):
    hi = fp16(v), lo = fp16((v - hi) * 2^11) (include/rt_stereo.h: rt_resblock_plan_set_split)"""
    n, c, h, p = a.shape
    g = a.reshape(n, c // 8, 8, h, p).transpose(0, 1, 3, 4, 2).astype(np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        hi = g.astype(np.float16)
        lo = ((g - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return np.ascontiguousarray(np.concatenate([hi, lo], axis=-1)).view(np.float32)


def from_split(a):
    """the values a pre-split tensor holds, hi + lo * 2^-11, as (N, C, H, P) fp32"""
    h16 = np.ascontiguousarray(a).view(np.float16)
    n, g, h, p, _ = h16.shape
    v = h16[..., :8].astype(np.float32) + h16[..., 8:].astype(np.float32) * np.float32(1 / 2048)
    return v.transpose(0, 1, 4, 2, 3).reshape(n, g * 8, h, p)


RBD_CASES = [
    # h, w, batch, pitch
    (16, 30, 1, 32),       # exactly one strip and one segment
    (17, 31, 1, 0),        # one row / one column past them, dense odd rows
    (37, 61, 2, 64),       # three segments (16 + 16 + 5), three strips (30 + 30 + 1), two images
    (5, 100, 1, 128),      # shorter than a segment
    (1, 1, 1, 0),          # single pixel
    (33, 7, 1, 32),        # narrower than a strip
    (70, 35, 1, 64),       # two 32-row segments and a rest: the x ring wraps more than once, the t ring's mirror rows are used
]


@pytest.mark.parametrize("x_split,y_split", [(1, 1), (1, 0), (0, 1)])
@pytest.mark.parametrize("h,w,batch,pitch", RBD_CASES)
def test_split_resblock_presplit_tensors(backend, h, w, batch, pitch, x_split, y_split):
    """the tower block reading and / or writing pre-split tensors -- conv_s3rbd_kernel (LDS-DMA fed) when it reads one,
    conv_s3rbs_kernel's split-writing epilogue when it only writes one -- against an fp64 evaluation of the two layers on the values
    the input tensor holds, and against the fp32-tensor form of the same plan.  The stored pair must be the split of the value it stands
    for (hi = fp16(v)): the next block's matrix instructions consume it as is."""
    c = 32
    x = rnd(batch, c, h, w)
    w1, b1 = rnd(c, c, 3, 3) * np.float32(1 / np.sqrt(c * 9)), rnd(c)
    w2, b2 = rnd(c, c, 3, 3) * np.float32(1 / np.sqrt(c * 9)), rnd(c)
    P = pitch or w
    xs = to_split(pitched(x, P))
    xv = from_split(xs)[..., :w] if x_split else x             # what the block computes on: 22 bits of x
    assert np.abs(xv - x).max() <= 2.0 ** -21 * np.abs(x).max()
    X = torch.from_numpy(xv).double()
    t = O.elu(O.conv2d(X, torch.from_numpy(w1).double(), torch.from_numpy(b1).double(), 1, 1))
    ref = O.elu(O.conv2d(t, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), 1, 1) + X).numpy()
    plan = backend.klib.resblock_plan(w1, b1, w2, b2, c, c, h, w)
    if pitch:
        plan.set_pitch(pitch, pitch)
    assert not plan.supports_split()                            # planar tensors: no
    plan.set_layouts(1, 1, 1)
    assert plan.supports_split()
    outs = []
    for xsp, ysp in ((x_split, y_split), (0, 0)):
        plan.set_split(xsp, ysp)
        xin = backend.dev(xs if xsp else to_il(pitched(x, P), 4))
        y = backend.empty((batch, c // 8, h, P, 8) if ysp else (batch, c // 4, h, P, 4))
        plan.enqueue(xin, y, xin, batch)
        raw = backend.host(y).copy()
        if ysp:
            h16 = np.ascontiguousarray(raw).view(np.float16)
            if P > w:
                assert np.isnan(raw[:, :, :, w:, :]).all(), "padding columns were written"     # (still the fp32 NaNs of backend.empty)
            out = from_split(raw)[..., :w]
            hi = h16[:, :, :, :w, :8].transpose(0, 1, 4, 2, 3).reshape(batch, c, h, w)
            # hi is a nearest fp16 of the value the pair stands for (at an exact tie the rounding of lo may put it on the other side)
            assert (np.abs(hi.astype(np.float32) - out) <= np.spacing(np.abs(hi)).astype(np.float32) * np.float32(0.5 + 2.0 ** -10)).all()
        else:
            out = from_il(raw)
            if P > w:
                assert np.isnan(out[..., w:]).all(), "padding columns were written"
            out = out[..., :w]
        outs.append(out)
    plan.destroy()
    # (fp32 accumulation of 2 x 288 products: relative to the largest value, which reaches 5 .. 10 on this data)
    assert np.abs(outs[0] - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), np.abs(outs[0] - ref).max()
    # the fp32-tensor form of the same block: same arithmetic up to the order of summation and the 22-bit input / output
    assert np.abs(outs[0] - outs[1]).max() <= 3e-6 * max(1.0, np.abs(ref).max()), np.abs(outs[0] - outs[1]).max()


def test_split_resblock_refuses_presplit_tensors_where_it_has_no_kernel(backend):
    w1, b1 = rnd(16, 16, 3, 3), rnd(16)
    plan = backend.klib.resblock_plan(w1, b1, w1, b1, 16, 16, 8, 32) if backend.klib.has_experimental() else None
    if plan is None:
        pytest.skip("the per-tile form (conv_s3rb_kernel) is compiled with RT_EXPERIMENTAL only")
    plan.set_layouts(1, 1, 1)
    assert not plan.supports_split()
    with pytest.raises(capi.RtError):
        plan.set_split(1, 1)
    plan.destroy()


# ---- correlation + soft-argmax on the matrix cores (corr_mfma.hip.h) ------------------------------------------------------
@pytest.mark.parametrize("shape,D,is_min,pitch", [((2, 32, 9, 140), 48, False, 160), ((1, 32, 5, 129), 64, False, 0),
                                                  ((1, 16, 7, 37), 6, True, 64), ((2, 8, 3, 33), 13, False, 0),
                                                  ((1, 4, 6, 6), 2, False, 32), ((1, 32, 4, 70), 33, True, 96)])
def test_corr_softargmax_mfma(backend, shape, D, is_min, pitch):
    """fused correlation + soft-argmax on channel-interleaved feature maps against the oracle (reference
    test_data_generator.py:242-259, 300-315) evaluated in fp64, incl. the zero entries left of the image (x < d)"""
    n, c, h, w = shape
    l, r = rnd(*shape) * np.float32(0.5), rnd(*shape) * np.float32(0.5)
    ref = O.softargmax(O.corr_cost_volume(torch.from_numpy(l).double(), torch.from_numpy(r).double(), D), is_min).numpy()
    P = pitch or w
    out = backend.empty((n, 1, h, P))
    backend.klib.corr_softargmax_il(backend.dev(to_il(pitched(l, P), 4)), backend.dev(to_il(pitched(r, P), 4)), out, n, c, h, w, D,
                                    is_min, P, P)
    got = backend.host(out)
    assert np.abs(got[..., :w] - ref).max() <= 2e-4 * max(1, D / 16), np.abs(got[..., :w] - ref).max()
    if P > w:
        assert np.isnan(got[..., w:]).all(), "padding columns were written"


def test_corr_softargmax_writes_the_ninth_group_and_conv_reads_33_interleaved_channels(backend):
    """the concatenation in front of conv2D_1 kept interleaved (resnet18_2D_513x257_net.cpp:601-615): one buffer of 9 groups of 4
    channels per sample -- 8 groups of the left feature map and a ninth whose lane 0 is the soft-argmax map (rt_corr_softargmax_il_slot,
    zeros in lanes 1..3) -- read by a 33 -> 32 convolution as a PADDED interleaved input (rt_conv_plan_supports_il8 bit 4).  Same bits
    as the planar concatenation."""
    n, c, h, w, D, P = 2, 32, 7, 45, 12, 64
    l, r = rnd(n, c, h, w) * np.float32(0.5), rnd(n, c, h, w) * np.float32(0.5)
    feat = rnd(n, 32, h, w)                                            # left_conv1_act's stand-in
    wt, b = rnd(32, 33, 3, 3) * np.float32(1 / np.sqrt(33 * 9)), rnd(32)
    lil, ril = backend.dev(to_il(pitched(l, P), 4)), backend.dev(to_il(pitched(r, P), 4))
    # planar reference of the same kernels: the map as a plane, the 33 channels planar
    disp = backend.empty((n, 1, h, P))
    backend.klib.corr_softargmax_il(lil, ril, disp, n, c, h, w, D, False, P, P)
    dmap = backend.host(disp).copy()
    cat = np.concatenate([pitched(feat, P), np.nan_to_num(dmap)], axis=1)           # (n, 33, h, P)
    plan = backend.klib.conv2d_plan(wt, b, 33, 32, h, w, 3, 1, 1, act=capi.RT_ACT_ELU)
    plan.set_pitch(P, P)
    caps = plan.il_caps()
    assert caps & 16 and not caps & 1, caps
    y = backend.empty((n, 32, h, P))
    plan.enqueue(backend.dev(cat), y, None, n)
    planar = backend.host(y).copy()
    # interleaved: (n, 9, h, P, 4), the ninth group written by the correlation kernel
    buf = np.full((n, 9, h, P, 4), np.nan, np.float32)
    buf[:, :8] = to_il(pitched(feat, P), 4)
    buf[:, 8] = 7.0                                                    # stale values: lanes 1..3 must be overwritten with zeros
    dbuf = backend.dev(buf)
    if backend.name == "gpu":
        slot = dbuf[:, 8]
        backend.klib.corr_softargmax_il(lil, ril, slot, n, c, h, w, D, False, P, P, out_bstride=9 * h * P * 4, out_slot=4)
    else:
        backend.klib.corr_softargmax_il(lil, ril, dbuf[:, 8], n, c, h, w, D, False, P, P, out_bstride=9 * h * P * 4, out_slot=4)
    got = backend.host(dbuf)
    assert np.array_equal(got[:, 8, :, :w, 0], dmap[:, 0, :, :w])
    assert (got[:, 8, :, :w, 1:] == 0).all() and (got[:, 8, :, w:] == 7.0).all()      # zeros beside the map, padding columns untouched
    plan.set_layouts(1, 0, 0)
    plan.set_batch_strides(9 * h * P * 4, 0, 0)
    y = backend.empty((n, 32, h, P))
    plan.enqueue(dbuf, y, None, n)
    assert np.array_equal(backend.host(y)[..., :w], planar[..., :w])
    plan.destroy()


# ---- default cost volume folded into the first Conv3D's gather (never materialised) ----------------------------------------
@pytest.mark.parametrize("f,k,h,w,D,batch", [(8, 16, 9, 37, 6, 2), (32, 32, 5, 40, 12, 1), (4, 8, 7, 33, 5, 1)])
def test_conv3d_on_folded_cost_volume(backend, f, k, h, w, D, batch):
    """rtConv3dDesc::cv_fold: Conv3D reads cv[d, 0:F] = L, cv[d, F:2F, y, x] = R[:, y, x - d] (0 for x < d) straight from
    the two feature maps (reference lib/kernels.cu:72-97 builds the (D, 2F, H, W) volume first); against the oracle's
    cost_volume + conv3d_tf in fp64"""
    l, r = rnd(batch, f, h, w), rnd(batch, f, h, w)
    wt, b = rnd(k, 3, 2 * f, 3, 3) * np.float32(1 / np.sqrt(27 * 2 * f)), rnd(k)
    cv = O.cost_volume(torch.from_numpy(l).double(), torch.from_numpy(r).double(), D)            # N D 2F H W
    ref = O.conv3d_tf(cv, torch.from_numpy(wt).double(), torch.from_numpy(b).double(), (1, 1, 1), (1, 1, 1), (1, 1, 1))
    ref = O.elu(O.transform(ref)).numpy()                                                         # (N, Do, K, Ho, Wo)
    plan = backend.klib.conv3d_plan(wt, b, 2 * f, k, (D, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU,
                                    out_dchw=True, cv_fold=f)
    x = np.concatenate([l, r], axis=1)                                                            # (N, 2F, H, W): [left | right]
    y = backend.empty((batch,) + plan.out_dims)
    plan.enqueue(backend.dev(x), y, None, batch)
    out = backend.host(y)
    plan.destroy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 4e-7 * np.sqrt(27 * 2 * f) + 2e-6, np.abs(out - ref).max()


# ---- fp16 storage of the 3-D tensors (half2 mode of the 3-D models) ------------------------------------------------------------
def _h16(a):
    return np.ascontiguousarray(a.astype(np.float16))


def _dev16(backend, a):
    return torch.from_numpy(_h16(a)).cuda() if backend.name == "gpu" else _h16(a)


def _host32(backend, t):
    if backend.name == "gpu":
        torch.cuda.synchronize()
        return t.cpu().numpy().astype(np.float32)
    return np.asarray(t).astype(np.float32)


def _empty16(backend, shape):
    if backend.name == "gpu":
        return torch.full(tuple(shape), float("nan"), dtype=torch.float16, device="cuda")
    return np.full(shape, np.nan, np.float16)


@pytest.mark.parametrize("x16", [True, False])
@pytest.mark.parametrize("c,k,d,h,w,stride,resid", [(8, 32, 5, 9, 35, 1, False), (16, 24, 6, 8, 37, 2, False), (32, 16, 4, 7, 33, 1, True)])
def test_conv3d_fp16_storage(backend, c, k, d, h, w, stride, resid, x16):
    """Conv3D with (D,C,H,W) tensors stored as fp16 (x16: input too -- inner layers; otherwise fp32 in, fp16 out -- the
    first layer after the fp32 feature towers), fp16 weight values, fp32 accumulation: against the oracle on the same
    (fp16-rounded) operands, one rounding of the output"""
    q16 = lambda a: a.astype(np.float16).astype(np.float32)
    x = q16(rnd(2, d, c, h, w)) if x16 else rnd(2, d, c, h, w)
    wt, b = q16(rnd(k, 3, c, 3, 3) * np.float32(1 / np.sqrt(27 * c))), rnd(k)
    if stride == 2:
        pads = (0, 1, 1) if d % 2 == 0 else (1, 1, 1)
        xin = O.pad_d(torch.from_numpy(x).double(), 1) if d % 2 == 0 else torch.from_numpy(x).double()
        ref = O.conv3d_tf(xin, torch.from_numpy(wt).double(), torch.from_numpy(b).double(), (2, 2, 2), pads, pads)
    else:
        pads = (1, 1, 1)
        ref = O.conv3d_tf(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), (1, 1, 1), pads, pads)
    ref = O.transform(ref)                                                                        # (N, Do, K, Ho, Wo)
    res = q16(rnd(*ref.shape)) if resid else None
    if resid:
        ref = ref + torch.from_numpy(res).double()
    ref = O.elu(ref).numpy()
    plan = backend.klib.conv3d_plan(_h16(wt), b.astype(np.float16), c, k, (d + (1 if stride == 2 and d % 2 == 0 else 0), h, w), (3, 3, 3), (stride,) * 3,
                                    pads, pads, act=capi.RT_ACT_ELU, out_dchw=True, has_residual=resid, dtype=capi.RT_F16,
                                    in_pad_end=1 if stride == 2 and d % 2 == 0 else 0)
    plan.set_io_types(capi.RT_F16 if x16 else capi.RT_F32, capi.RT_F16)
    y = _empty16(backend, ref.shape)
    plan.enqueue(_dev16(backend, x) if x16 else backend.dev(x), y, _dev16(backend, res) if resid else None, 2)
    out = _host32(backend, y)
    plan.destroy()
    bq = b.astype(np.float16).astype(np.float32)                      # the bias travels in the fp16 weight file too
    tol = 2e-3 * max(1.0, float(np.abs(ref).max())) + np.abs(b - bq).max()
    assert np.abs(out - ref).max() <= tol, (np.abs(out - ref).max(), tol)
