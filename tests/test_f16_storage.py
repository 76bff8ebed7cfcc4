"""fp16 storage of activations (TensorRT half2 mode): the convolution kernels, the fused correlation and the whole
ResNet-18 2D network with fp16 tensors between layers and fp32 arithmetic.  Tolerances follow the reference's own
fp16 tests (1e-2, tests_main.cpp:320,1025); the conv-level checks are tighter (one rounding of the output)."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi
from test_ops_parity import T, rnd
from test_pitch_parity import pitched


def h16(a):
    return np.ascontiguousarray(a.astype(np.float16))


def dev16(backend, a):
    return torch.from_numpy(h16(a)).cuda() if backend.name == "gpu" else h16(a)


def empty16(backend, shape):
    if backend.name == "gpu":
        return torch.full(tuple(shape), float("nan"), dtype=torch.float16, device="cuda")
    return np.full(shape, np.nan, np.float16)


def host(backend, t):
    if backend.name == "gpu":
        torch.cuda.synchronize()
        return t.cpu().numpy().astype(np.float32)
    return np.asarray(t).astype(np.float32)


CASES = [
    # cin, cout, h, w, k, stride, transposed, act, resid, x16, y16
    (32, 32, 9, 37, 3, 1, False, capi.RT_ACT_ELU, True, True, True),       # Winograd
    (3, 32, 11, 29, 5, 2, False, capi.RT_ACT_ELU, False, False, True),     # first layer: fp32 image in, fp16 out
    (32, 64, 9, 35, 3, 2, False, capi.RT_ACT_ELU, False, True, True),      # stride 2
    (64, 32, 5, 9, 3, 2, True, capi.RT_ACT_ELU, True, True, True),         # transposed, merged phases
    (32, 1, 6, 13, 3, 2, True, capi.RT_ACT_SIGMOID, False, True, False),   # last layer: fp16 in, fp32 out
    (40, 20, 9, 37, 3, 1, False, capi.RT_ACT_ELU, True, True, True),       # ragged channel counts (chunks of 16, tail of 4)
    (16, 72, 8, 70, 3, 1, False, capi.RT_ACT_NONE, False, True, True),     # three column blocks, three tiles across
    (24, 32, 7, 33, 3, 2, True, capi.RT_ACT_NONE, False, True, True),      # transposed, odd sizes
    (32, 33, 6, 41, 3, 1, False, capi.RT_ACT_ELU, True, True, True),       # odd channel count: 2-byte output accesses
    (16, 32, 10, 66, 3, 2, False, capi.RT_ACT_ELU, True, True, True),      # stride 2 with residual, odd output width (33)
]


@pytest.mark.parametrize("cin,cout,h,w,k,stride,tr,act,resid,x16,y16", CASES)
def test_conv2d_f16_storage(backend, cin, cout, h, w, k, stride, tr, act, resid, x16, y16):
    batch, pad = 2, (k // 2 if not tr else 1)
    x, b = rnd(batch, cin, h, w), rnd(cout)
    xq = x.astype(np.float16).astype(np.float32) if x16 else x
    # weights representable in fp16 (what trt_weights_fp16.bin holds): with both tensors fp16 the kernel multiplies in
    # fp16 on the matrix cores (conv_f16.hip.h), which is then exact in its operands
    q16 = lambda a: a.astype(np.float16).astype(np.float32)
    if tr:
        wt = q16(rnd(cin, cout, k, k) * np.float32(1 / np.sqrt(cin * k * k)))
        ref = O.deconv2d(T(xq), T(wt), T(b), stride, pad)
    else:
        wt = q16(rnd(cout, cin, k, k) * np.float32(1 / np.sqrt(cin * k * k)))
        ref = O.conv2d(T(xq), T(wt), T(b), stride, pad)
    res = rnd(*ref.shape) if resid else None
    if resid:
        ref = ref + T(res.astype(np.float16).astype(np.float32) if y16 else res)
    ref = O.elu(ref) if act == capi.RT_ACT_ELU else (torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref)
    ref = ref.numpy()
    wo = ref.shape[-1]
    ip, op = (w + 63) // 64 * 64, (wo + 63) // 64 * 64
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, k, stride, pad, act=act, has_residual=resid, transposed=tr)
    plan.set_pitch(ip, op)
    plan.set_io_types(capi.RT_F16 if x16 else capi.RT_F32, capi.RT_F16 if y16 else capi.RT_F32)
    xin = dev16(backend, pitched(x, ip)) if x16 else backend.dev(pitched(x, ip))      # NaN in the padding columns
    rin = None
    if resid:
        rin = dev16(backend, pitched(res, op, 0.0)) if y16 else backend.dev(pitched(res, op))
    y = empty16(backend, ref.shape[:-1] + (op,)) if y16 else backend.empty(ref.shape[:-1] + (op,))
    plan.enqueue(xin, y, rin, batch)
    out = host(backend, y) if y16 else backend.host(y)
    tol = 2e-3 * max(1.0, float(np.abs(ref).max())) if y16 else 3e-5
    assert np.abs(out[..., :wo] - ref).max() <= tol
    assert np.isnan(out[..., wo:]).all(), "padding columns were written"
    plan.destroy()


def test_corr_softargmax_f16_storage(backend):
    n, c, h, w, d = 2, 16, 7, 45, 12
    l, r = rnd(n, c, h, w) * np.float32(0.5), rnd(n, c, h, w) * np.float32(0.5)
    lq, rq = (a.astype(np.float16).astype(np.float32) for a in (l, r))
    ref = O.softargmax(O.corr_cost_volume(T(lq), T(rq), d), False).numpy()
    out = empty16(backend, (n, 1, h, 64))
    backend.klib.corr_softargmax_pitched(dev16(backend, pitched(l, 64, 0.0)), dev16(backend, pitched(r, 64, 0.0)), out, n, c,
                                         h, w, d, False, 64, 64, dtype=capi.RT_F16)
    res = host(backend, out)
    assert np.abs(res[..., :w] - ref).max() <= 8e-3          # half ulp of values below 16
    assert np.isnan(res[..., w:]).all()


def to_il8(a):
    """(N, C, H, P) planar -> (N, C/8, H, P, 8) channel-interleaved"""
    n, c, h, p = a.shape
    return np.ascontiguousarray(a.reshape(n, c // 8, 8, h, p).transpose(0, 1, 3, 4, 2))


def from_il8(a):
    n, g, h, p, _ = a.shape
    return a.transpose(0, 1, 4, 2, 3).reshape(n, g * 8, h, p)


@pytest.mark.parametrize("shape,D,is_min,pitch", [((2, 32, 9, 140), 48, False, 192), ((1, 32, 5, 129), 64, False, 192), ((1, 16, 7, 37), 6, True, 64),
                                                  ((2, 8, 3, 33), 13, False, 64), ((1, 32, 4, 70), 33, True, 128)])
def test_corr_softargmax_mfma_f16(backend, shape, D, is_min, pitch):
    """half2 mode's correlation + soft-argmax on the matrix cores (corr_softargmax_mfma_f16_kernel): fp16 channel-interleaved maps
    (C/8, H, pitch, 8), fp16 map out, against the oracle on the fp16-rounded maps evaluated in fp64 (reference
    test_data_generator.py:242-259, 300-315; fp16 tolerance tests_main.cpp:1025) and against the planar fp16 kernel it replaces in the
    engine -- the same exact products, another order of the fp32 sums: one fp16 ulp of the map at most."""
    n, c, h, w = shape
    l, r = rnd(*shape) * np.float32(0.5), rnd(*shape) * np.float32(0.5)
    lq, rq = (a.astype(np.float16).astype(np.float32) for a in (l, r))
    ref = O.softargmax(O.corr_cost_volume(torch.from_numpy(lq).double(), torch.from_numpy(rq).double(), D), is_min).numpy()
    out = empty16(backend, (n, 1, h, pitch))
    backend.klib.corr_softargmax_il8_f16(dev16(backend, to_il8(pitched(l, pitch))), dev16(backend, to_il8(pitched(r, pitch))), out, n, c, h, w, D,
                                         is_min, pitch, pitch)
    got = host(backend, out)
    assert np.isnan(got[..., w:]).all(), "padding columns were written"
    ulp = np.maximum(np.abs(ref), 1.0) * 2.0 ** -10                   # fp16 spacing at the value
    assert (np.abs(got[..., :w] - ref) <= 0.6 * ulp + 2e-4 * max(1, D / 16)).all(), np.abs(got[..., :w] - ref).max()
    planar = empty16(backend, (n, 1, h, pitch))
    backend.klib.corr_softargmax_pitched(dev16(backend, pitched(l, pitch, 0.0)), dev16(backend, pitched(r, pitch, 0.0)), planar, n, c, h, w, D, is_min,
                                         pitch, pitch, dtype=capi.RT_F16)
    assert (np.abs(got[..., :w] - host(backend, planar)[..., :w]) <= ulp).all()


def test_corr_f16_writes_the_fifth_group_and_conv_reads_33_interleaved_fp16_channels(backend):
    """half2 mode's form of the interleaved concatenation in front of conv2D_1 (resnet18_2D_513x257_net.cpp:601-615): one buffer of 5 groups
    of 8 fp16 channels per sample -- 4 groups of the left feature map and a fifth whose lane 0 is the soft-argmax map
    (rt_corr_softargmax_il8_f16, out_slot = 8: zeros in lanes 1 .. 7) -- read by a 33 -> 32 convolution on fp16 operands as a PADDED
    interleaved input (rt_conv_plan_supports_il8 bit 4).  Same bits as the planar concatenation through the same kernels."""
    n, c, h, w, D, P = 2, 32, 7, 45, 12, 64
    q16 = lambda a: a.astype(np.float16).astype(np.float32)
    l, r = q16(rnd(n, c, h, w) * np.float32(0.5)), q16(rnd(n, c, h, w) * np.float32(0.5))
    feat = q16(rnd(n, 32, h, w))
    wt, b = q16(rnd(32, 33, 3, 3) * np.float32(1 / np.sqrt(33 * 9))), rnd(32)
    lil, ril = dev16(backend, to_il8(pitched(l, P))), dev16(backend, to_il8(pitched(r, P)))
    # planar reference of the same kernels: the map as a plane, the 33 channels planar
    disp = empty16(backend, (n, 1, h, P))
    backend.klib.corr_softargmax_il8_f16(lil, ril, disp, n, c, h, w, D, False, P, P)
    dmap = host(backend, disp)[..., :w]
    x33 = np.concatenate([feat, dmap], axis=1)
    plan = backend.klib.conv2d_plan(wt, b, 33, 32, h, w, 3, 1, 1, act=capi.RT_ACT_ELU)
    plan.set_pitch(P, P)
    plan.set_io_types(capi.RT_F16, capi.RT_F16)
    assert plan.il_caps() & 16 and not plan.il_caps() & 1
    yp = empty16(backend, (n, 32, h, P))
    plan.enqueue(dev16(backend, pitched(x33, P, 0.0)), yp, None, n)
    planar = host(backend, yp)[..., :w]
    # interleaved: [4 groups of feat | fifth group: lane 0 = map], 40 channels allocated per sample
    buf = np.full((n, 5, h, P, 8), np.nan, np.float32)
    buf[:, :4] = to_il8(pitched(feat, P))
    bdev = dev16(backend, buf)
    elems = 5 * h * P * 8
    if backend.name == "gpu":
        slot = bdev.view(-1)[4 * h * P * 8:]
    else:
        slot = bdev.reshape(-1)[4 * h * P * 8:]
    backend.klib.corr_softargmax_il8_f16(lil, ril, slot, n, c, h, w, D, False, P, P, out_bstride=elems, out_slot=8)
    got5 = host(backend, bdev)
    assert np.array_equal(got5[:, 4, :, :w, 0], dmap[:, 0]) and (got5[:, 4, :, :w, 1:] == 0).all()
    plan.set_layouts(1, 0, 0)
    plan.set_batch_strides(elems, 0, 0)
    yi = empty16(backend, (n, 32, h, P))
    plan.enqueue(bdev, yi, None, n)
    assert np.array_equal(host(backend, yi)[..., :w], planar)
    plan.destroy()


@pytest.mark.parametrize("x_il8,y_il8,r_il8", [(1, 1, 1), (1, 1, 0), (0, 1, 0), (1, 0, 1), (0, 0, 1), (1, 0, 0)])
@pytest.mark.parametrize("cin,cout,h,w,resid", [(32, 32, 9, 37, True), (16, 72, 6, 70, True), (40, 24, 7, 33, False)])
def test_conv2d_f16_interleaved(backend, cin, cout, h, w, resid, x_il8, y_il8, r_il8):
    """the fp16-arithmetic kernel on channel-interleaved tensors, every combination with planar ones"""
    if r_il8 and not resid:
        pytest.skip("no residual")
    batch, act = 2, capi.RT_ACT_ELU
    q16 = lambda a: a.astype(np.float16).astype(np.float32)
    x, b = q16(rnd(batch, cin, h, w)), rnd(cout)
    wt = q16(rnd(cout, cin, 3, 3) * np.float32(1 / np.sqrt(cin * 9)))
    ref = O.conv2d(T(x), T(wt), T(b), 1, 1)
    res = q16(rnd(*ref.shape)) if resid else None
    if resid:
        ref = ref + T(res)
    ref = O.elu(ref).numpy()
    pitch = (w + 63) // 64 * 64
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, 3, 1, 1, act=act, has_residual=resid)
    plan.set_pitch(pitch, pitch)
    plan.set_io_types(capi.RT_F16, capi.RT_F16)
    assert plan.supports_il8()
    plan.set_layouts(x_il8, y_il8, r_il8)
    lay = lambda a, il: to_il8(a) if il else a
    xin = dev16(backend, lay(pitched(x, pitch), x_il8))                       # NaN in the padding columns
    rin = dev16(backend, lay(pitched(res, pitch), r_il8)) if resid else None
    yshape = (batch, cout // 8, h, pitch, 8) if y_il8 else (batch, cout, h, pitch)
    y = empty16(backend, yshape)
    plan.enqueue(xin, y, rin, batch)
    out = host(backend, y)
    out = from_il8(out) if y_il8 else out
    assert np.abs(out[..., :w] - ref).max() <= 2e-3 * max(1.0, float(np.abs(ref).max()))
    assert np.isnan(out[..., w:]).all(), "padding columns were written"
    plan.destroy()


@pytest.mark.parametrize("h,w,batch,pitch", [(16, 30, 1, 64), (17, 31, 1, 64), (37, 61, 2, 64), (5, 100, 1, 128), (1, 1, 1, 64), (33, 7, 1, 64),
                                             (70, 35, 1, 64)])
def test_resblock_f16_fused_is_bit_identical_to_its_two_layers(backend, h, w, batch, pitch):
    """the tower block of half2 mode in ONE launch (conv_f16rbd_kernel: fp16 tensors by LDS-DMA, fp16 operands, the intermediate rounded
    to fp16 in LDS) against the two launches of conv_f16mma_kernel it replaces -- same roundings, same order of summation: the same bits
    -- and against the oracle on the fp16-rounded operands.  Strips of 30 columns x segments of 16 / 32 rows, image edges, batch."""
    c, act = 32, capi.RT_ACT_ELU
    q16 = lambda a: a.astype(np.float16).astype(np.float32)
    x = q16(rnd(batch, c, h, w))
    w1, b1 = q16(rnd(c, c, 3, 3) * np.float32(1 / np.sqrt(c * 9))), rnd(c)
    w2, b2 = q16(rnd(c, c, 3, 3) * np.float32(1 / np.sqrt(c * 9))), rnd(c)
    xin = dev16(backend, to_il8(pitched(x, pitch)))

    def layer(wt, b, src, resid):
        plan = backend.klib.conv2d_plan(wt, b, c, c, h, w, 3, 1, 1, act=act, has_residual=resid is not None)
        plan.set_pitch(pitch, pitch)
        plan.set_io_types(capi.RT_F16, capi.RT_F16)
        plan.set_layouts(1, 1, 1 if resid is not None else 0)
        y = empty16(backend, (batch, c // 8, h, pitch, 8))
        plan.enqueue(src, y, resid, batch)
        plan.destroy()
        return y

    t = layer(w1, b1, xin, None)
    two = host(backend, layer(w2, b2, t, xin))
    plan = backend.klib.resblock_plan(w1, b1, w2, b2, c, c, h, w)
    plan.set_pitch(pitch, pitch)
    plan.set_io_types(capi.RT_F16, capi.RT_F16)
    plan.set_layouts(1, 1, 1)
    assert not plan.supports_split()                    # pre-split tensors are the fp32 engines'
    y = empty16(backend, (batch, c // 8, h, pitch, 8))
    plan.enqueue(xin, y, xin, batch)
    one = host(backend, y)
    plan.destroy()
    assert np.isnan(one[:, :, :, w:, :]).all(), "padding columns were written"
    assert np.array_equal(one[:, :, :, :w, :], two[:, :, :, :w, :]), np.abs(one[:, :, :, :w, :] - two[:, :, :, :w, :]).max()
    tt = q16(O.elu(O.conv2d(T(x), T(w1), T(b1), 1, 1)).numpy())
    ref = O.elu(O.conv2d(T(tt), T(w2), T(b2), 1, 1) + T(x)).numpy()
    out = from_il8(one)[..., :w]
    assert np.abs(out - ref).max() <= 4e-3 * max(1.0, float(np.abs(ref).max()))     # (an fp16 rounding of t that falls the other way, and of y)


def test_interleaved_layout_is_refused_where_it_does_not_exist(backend):
    plan = backend.klib.conv2d_plan(rnd(32, 32, 3, 3), rnd(32), 32, 32, 9, 20, 3, 2, 1)       # stride 2
    plan.set_pitch(64, 64)
    plan.set_io_types(capi.RT_F16, capi.RT_F16)
    assert not plan.supports_il8()
    with pytest.raises(capi.RtError):
        plan.set_layouts(1, 1)
    plan.destroy()


@pytest.mark.parametrize("cin,cout,h,w,y_il8", [(3, 32, 11, 29, 0), (3, 32, 11, 29, 1), (3, 64, 21, 135, 1), (1, 40, 9, 66, 0), (2, 24, 8, 70, 1)])
def test_first_layer_f16(backend, cin, cout, h, w, y_il8):
    """5x5 stride-2 first layer in half2 mode: fp32 image in, fp16 operands (image and weights rounded to fp16), fp16
    tensor out, planar or channel-interleaved (conv_f16_first.hip.h)"""
    batch, act = 2, capi.RT_ACT_ELU
    q16 = lambda a: a.astype(np.float16).astype(np.float32)
    x, b = rnd(batch, cin, h, w), rnd(cout)
    wt = q16(rnd(cout, cin, 5, 5) * np.float32(1 / np.sqrt(cin * 25)))
    ref = O.elu(O.conv2d(T(q16(x)), T(wt), T(b), 2, 2)).numpy()
    ho, wo = ref.shape[-2:]
    op = (wo + 63) // 64 * 64
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, 5, 2, 2, act=act)
    plan.set_pitch(0, op)
    plan.set_io_types(capi.RT_F32, capi.RT_F16)
    if y_il8:
        assert plan.supports_il8()
        plan.set_layouts(0, 1)
        with pytest.raises(capi.RtError):
            plan.set_layouts(1, 1)                                   # the image is planar fp32
        plan.set_layouts(0, 1)
    y = empty16(backend, (batch, cout // 8, ho, op, 8) if y_il8 else (batch, cout, ho, op))
    plan.enqueue(backend.dev(x), y, None, batch)
    out = host(backend, y)
    out = from_il8(out) if y_il8 else out
    assert np.abs(out[..., :wo] - ref).max() <= 2e-3 * max(1.0, float(np.abs(ref).max()))
    assert np.isnan(out[..., wo:]).all(), "padding columns were written"
    plan.destroy()


def _fuzz_cases(n=24, seed=20260924):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        cin, cout = int(rng.choice([8, 16, 24, 32, 40, 64])), int(rng.choice([8, 16, 32, 40, 64, 72]))
        h, w = int(rng.integers(1, 14)), int(rng.integers(1, 80))
        resid = bool(rng.integers(0, 2))
        x_il8, y_il8 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        r_il8 = int(rng.integers(0, 2)) if resid else 0
        act = int(rng.choice([capi.RT_ACT_NONE, capi.RT_ACT_ELU, capi.RT_ACT_SIGMOID]))
        cases.append((cin, cout, h, w, resid, x_il8, y_il8, r_il8, act, int(rng.integers(1, 4))))
    return cases


@pytest.mark.parametrize("cin,cout,h,w,resid,x_il8,y_il8,r_il8,act,batch", _fuzz_cases())
def test_conv2d_f16_fuzz(backend, cin, cout, h, w, resid, x_il8, y_il8, r_il8, act, batch):
    """seeded random shapes / layouts / epilogues for the fp16-arithmetic 3x3 kernel (1-pixel images, single rows,
    widths below one tile, channel counts that are not multiples of 32, every layout mix)"""
    q16 = lambda a: a.astype(np.float16).astype(np.float32)
    x, b = q16(rnd(batch, cin, h, w)), rnd(cout)
    wt = q16(rnd(cout, cin, 3, 3) * np.float32(1 / np.sqrt(cin * 9)))
    ref = O.conv2d(T(x), T(wt), T(b), 1, 1)
    res = q16(rnd(*ref.shape)) if resid else None
    if resid:
        ref = ref + T(res)
    ref = (O.elu(ref) if act == capi.RT_ACT_ELU else torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref).numpy()
    pitch = (w + 63) // 64 * 64
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, 3, 1, 1, act=act, has_residual=resid)
    plan.set_pitch(pitch, pitch)
    plan.set_io_types(capi.RT_F16, capi.RT_F16)
    plan.set_layouts(x_il8, y_il8, r_il8)
    lay = lambda a, il: to_il8(a) if il else a
    xin = dev16(backend, lay(pitched(x, pitch), x_il8))
    rin = dev16(backend, lay(pitched(res, pitch), r_il8)) if resid else None
    y = empty16(backend, (batch, cout // 8, h, pitch, 8) if y_il8 else (batch, cout, h, pitch))
    plan.enqueue(xin, y, rin, batch)
    out = host(backend, y)
    out = from_il8(out) if y_il8 else out
    assert np.abs(out[..., :w] - ref).max() <= 2e-3 * max(1.0, float(np.abs(ref).max()))
    assert np.isnan(out[..., w:]).all(), "padding columns were written"
    plan.destroy()


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("cin,cout,h,w,k,stride,tr", [(32, 32, 9, 37, 3, 1, False), (3, 32, 11, 29, 5, 2, False), (64, 32, 5, 9, 3, 2, True)])
def test_io_types_round_trip_restores_fp32(backend, cin, cout, h, w, k, stride, tr, exact):
    """set_io_types(F16, ..) switches a plan to the fp16-operand kernels (fp16 weight slabs, fp16 tiling);
    set_io_types(F32, F32) afterwards -- what the executor does when half2 mode has to fall back -- must restore the
    fp32 form completely: same bits as a plan that never left fp32 (round 1 returned NaN with a success code).  exact: a plan created
    with RT_CONV_EXACT_FP32 comes back as an exact-fp32 plan (the re-planning ran outside the option's scope and silently
    returned a split-fp16 plan with its |x| < 65504 domain)."""
    flags = capi.RT_CONV_EXACT_FP32 if exact else 0
    batch, pad = 2, (k // 2 if not tr else 1)
    x, b = rnd(batch, cin, h, w), rnd(cout)
    wt = rnd(*((cin, cout, k, k) if tr else (cout, cin, k, k))) * np.float32(1 / np.sqrt(cin * k * k))
    ref = (O.deconv2d if tr else O.conv2d)(T(x), T(wt), T(b), stride, pad).numpy()
    wo = ref.shape[-1]
    ip, op = (w + 63) // 64 * 64, (wo + 63) // 64 * 64
    outs = []
    for switch in (False, True):
        plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, k, stride, pad, transposed=tr, flags=flags)
        plan.set_pitch(ip, op)
        limit = plan.input_limit()
        assert limit == (float("inf") if exact else 65504.0)
        if switch:
            plan.set_io_types(capi.RT_F32 if cin == 3 else capi.RT_F16, capi.RT_F16)
            plan.set_io_types(capi.RT_F32, capi.RT_F32)
            assert plan.input_limit() == limit
        y = backend.empty(ref.shape[:-1] + (op,))
        plan.enqueue(backend.dev(pitched(x, ip)), y, None, batch)
        outs.append(np.array(backend.host(y))[..., :wo])
        plan.destroy()
    assert np.abs(outs[0] - ref).max() <= 3e-5
    assert np.array_equal(outs[0], outs[1])
