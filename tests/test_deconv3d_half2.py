"""half2 mode of the 3-D decoder at the operator level (BASELINE config C5, NVSmall fp16): Conv3DTranspose on fp16 operands with a
channel-interleaved input (K/8, Dy, Hy, Wy, 8) -- conv_f16mma_kernel over the stride-2 phases -- writing interleaved (D, C/8, H, W, 8) or,
with the fused Transform, (C/8, D, H, W, 8) tensors; Conv3D writing the channel-major interleaved tensor such a layer reads; and the
last layer (one or two output channels) on the matrix cores (deconv3d_s2_il_kernel).  Reference: the oracle on the same fp16-rounded
operands (lib/conv3d_transpose_plugin.cpp:205-243 semantics: cuDNN backward-data of the forward convolution), fp32 accumulation, one
rounding of the output.  The planar fp16 forms of the same plans must agree with the interleaved ones to that rounding as well."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi
from test_ops_parity import T, rnd


def q16(a):
    return a.astype(np.float16).astype(np.float32)


def h16(a):
    return np.ascontiguousarray(a.astype(np.float16))


def dev16(backend, a):
    return torch.from_numpy(h16(a)).cuda() if backend.name == "gpu" else h16(a)


def empty(backend, shape, f16):
    if backend.name == "gpu":
        return torch.full(tuple(shape), float("nan"), dtype=torch.float16 if f16 else torch.float32, device="cuda")
    return np.full(shape, np.nan, np.float16 if f16 else np.float32)


def host(backend, t):
    if backend.name == "gpu":
        torch.cuda.synchronize()
        return t.cpu().numpy().astype(np.float32)
    return np.asarray(t).astype(np.float32)


def il_cm(a):
    """channel-major (N, K, D, H, W) -> (N, K/8, D, H, W, 8)"""
    n, k, d, h, w = a.shape
    return np.ascontiguousarray(a.reshape(n, k // 8, 8, d, h, w).transpose(0, 1, 3, 4, 5, 2))


def un_il_cm(a):
    n, g, d, h, w, _ = a.shape
    return a.transpose(0, 1, 5, 2, 3, 4).reshape(n, g * 8, d, h, w)


def il_dm(a):
    """depth-major (N, D, C, H, W) -> (N, D, C/8, H, W, 8)"""
    n, d, c, h, w = a.shape
    return np.ascontiguousarray(a.reshape(n, d, c // 8, 8, h, w).transpose(0, 1, 2, 4, 5, 3))


def un_il_dm(a):
    n, d, g, h, w, _ = a.shape
    return a.transpose(0, 1, 2, 5, 3, 4).reshape(n, d, g * 8, h, w)


DECONV_CASES = [
    # K, C, (Dy, Hy, Wy), out depth computed (full), kept, pad_d, skip, cdhw
    (16, 8, (3, 4, 19), 7, 6, 0, True, True),       # NVSmall's decoder pattern: depth pad 0, surplus slice dropped, fused Transform
    (16, 8, (3, 4, 19), 7, 6, 0, True, False),      # ... without the Transform: (D, C/8, H, W, 8)
    (8, 16, (3, 5, 9), 5, 5, 1, True, True),        # depth pad 1 (odd depth), K = 8: one half-empty chunk of 16 gathered channels
    (24, 40, (2, 3, 17), 3, 3, 1, False, True),     # no skip tensor, channel counts that are not multiples of 16 / 32
    (32, 32, (2, 6, 33), 5, 4, 0, True, True),      # two 32-pixel tiles across, 2 row tiles
]


@pytest.mark.parametrize("p4", [1, 0])
@pytest.mark.parametrize("r_il", [1, 0])
@pytest.mark.parametrize("K,C,ydims,dfull,dkeep,pad_d,skip,cdhw", DECONV_CASES)
def test_conv3d_transpose_f16_interleaved(backend, monkeypatch, K, C, ydims, dfull, dkeep, pad_d, skip, cdhw, r_il, p4):
    """p4 = 1: interleaved in AND out runs on deconv_f16p_kernel (all four output phases per workgroup); 0: the ZSlice form, one phase per
    workgroup (what interleaved-in / planar-out always uses)"""
    if r_il and not skip:
        pytest.skip("no residual")
    monkeypatch.setenv("RT_NO_DECONV_P4", "0" if p4 else "1")
    n = 2
    dy, hy, wy = ydims
    hx, wx = 2 * hy - 1, 2 * wy - 1
    y = q16(rnd(n, K, dy, hy, wy))
    w = q16(rnd(K, 3, C, 3, 3) * np.float32(1 / np.sqrt(27 * K / 8)))
    b = q16(rnd(C))
    sk = q16(rnd(n, dkeep, C, hx, wx)) if skip else None
    ps = (pad_d, 1, 1)
    ref = O.conv3d_transpose_tf(T(y).double(), T(w).double(), T(b).double(), (dfull, C, hx, wx), (2, 2, 2), ps, ps)[:, :dkeep]
    if skip:
        ref = ref + T(sk).double()
    ref = O.elu(ref)
    if cdhw:
        ref = O.transform(ref)
    ref = ref.numpy()
    tol = 2e-3 * max(1.0, float(np.abs(ref).max()))

    def make():
        p = backend.klib.conv3d_plan(h16(w), h16(b), C, K, (dfull, hx, wx), (3, 3, 3), (2, 2, 2), ps, ps, act=capi.RT_ACT_ELU, out_dchw=cdhw,
                                     has_residual=skip, dtype=capi.RT_F16, transposed_in_dims=ydims, out_depth=dkeep)
        p.set_io_types(capi.RT_F16, capi.RT_F16)
        return p
    # planar fp16 tensors (the split kernel), the skip tensor interleaved or not
    plan = make()
    caps = plan.il_caps()
    assert caps & 1 and caps & 2 and caps & 8 and bool(caps & 4) == skip, caps
    if r_il:
        plan.set_layouts(0, 0, 1)
    out = empty(backend, ref.shape, True)
    plan.enqueue(dev16(backend, y), out, dev16(backend, il_dm(sk) if r_il else sk) if skip else None, n)
    planar = host(backend, out)
    assert np.abs(planar - ref).max() <= tol
    with pytest.raises(capi.RtError):
        plan.set_layouts(0, 1, r_il)                  # an interleaved output needs the interleaved input (fp16 operands)
    # interleaved input: fp16 operands, interleaved output
    plan.set_layouts(1, 1, r_il)
    oshape = (n, C // 8, dkeep, hx, wx, 8) if cdhw else (n, dkeep, C // 8, hx, wx, 8)
    out = empty(backend, oshape, True)
    plan.enqueue(dev16(backend, il_cm(y)), out, dev16(backend, il_dm(sk) if r_il else sk) if skip else None, n)
    got = host(backend, out)
    assert not np.isnan(got).any()
    got = un_il_cm(got) if cdhw else un_il_dm(got)
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()
    assert np.abs(got - planar).max() <= tol
    # interleaved input, planar output
    plan.set_layouts(1, 0, r_il)
    out = empty(backend, ref.shape, True)
    plan.enqueue(dev16(backend, il_cm(y)), out, dev16(backend, il_dm(sk) if r_il else sk) if skip else None, n)
    assert np.abs(host(backend, out) - ref).max() <= tol
    # and back: the plan returns to the split kernel on planar tensors with the same bits as before
    plan.set_layouts(0, 0, r_il)
    out = empty(backend, ref.shape, True)
    plan.enqueue(dev16(backend, y), out, dev16(backend, il_dm(sk) if r_il else sk) if skip else None, n)
    assert np.array_equal(host(backend, out), planar)
    plan.destroy()


@pytest.mark.parametrize("c,k,d,h,w,stride", [(16, 32, 5, 9, 35, 1), (16, 24, 6, 8, 37, 2), (32, 64, 3, 5, 33, 1)])
def test_conv3d_writes_channel_major_interleaved(backend, c, k, d, h, w, stride):
    """the last Conv3D of the encoder keeps (K, D, H, W) (no Transform, nvsmall_1025x321_net.cpp:287-300): with an interleaved input it
    writes (K/8, D, H, W, 8) -- the tensor the first Conv3DTranspose reads"""
    n = 2
    x = q16(rnd(n, d, c, h, w))
    wt, b = q16(rnd(k, 3, c, 3, 3) * np.float32(1 / np.sqrt(27 * c))), q16(rnd(k))
    even = stride == 2 and d % 2 == 0
    pads = (0, 1, 1) if even else (1, 1, 1)
    xin = O.pad_d(T(x).double(), 1) if even else T(x).double()
    ref = O.elu(O.conv3d_tf(xin, T(wt).double(), T(b).double(), (stride,) * 3, pads, pads)).numpy()      # (N, K, Do, Ho, Wo)
    plan = backend.klib.conv3d_plan(h16(wt), h16(b), c, k, (d + (1 if even else 0), h, w), (3, 3, 3), (stride,) * 3, pads, pads,
                                    act=capi.RT_ACT_ELU, out_dchw=False, dtype=capi.RT_F16, in_pad_end=1 if even else 0)
    plan.set_io_types(capi.RT_F16, capi.RT_F16)
    assert plan.il_caps() & 3 == 3
    plan.set_layouts(1, 1, 0)
    out = empty(backend, (n, k // 8) + ref.shape[2:] + (8,), True)
    plan.enqueue(dev16(backend, il_dm(x)), out, None, n)
    got = un_il_cm(host(backend, out))
    assert np.abs(got - ref).max() <= 2e-3 * max(1.0, float(np.abs(ref).max()))
    plan.destroy()


WALK_CASES = [
    # K, C, (Dy, Hy, Wy), out depth computed (full), kept, pad_d, skip, cdhw, segments
    (16, 8, (3, 4, 19), 7, 6, 0, True, True, 0),        # NVSmall's decoder pattern; segments chosen by the plan
    (32, 32, (5, 6, 33), 11, 10, 0, True, True, 2),     # five slices per class in two segments (3 + 2), two column tiles, two row tiles
    (32, 64, (4, 5, 35), 9, 8, 0, True, False, 1),      # two blocks of 32 output channels, one segment, a ragged column tile, no Transform
    (24, 40, (2, 3, 17), 3, 3, 1, False, True, 3),      # no skip tensor, ragged channel counts, more segments asked for than slices
    (48, 16, (6, 9, 40), 13, 12, 0, True, True, 4),     # three chunks per slice (even class), six (odd): the ring position at a slice's end varies
]


@pytest.mark.parametrize("K,C,ydims,dfull,dkeep,pad_d,skip,cdhw,nseg", WALK_CASES)
def test_conv3d_transpose_f16_walks_down_the_depths(backend, monkeypatch, K, C, ydims, dfull, dkeep, pad_d, skip, cdhw, nseg):
    """deconv_f16pw_kernel: a workgroup keeps its tile of the input plane and walks a segment of its class's output depths (chunks by LDS-DMA
    two ahead, the next slice's loads under the epilogue), both classes in one launch or one each -- against the oracle, and bit for bit against
    deconv_f16p_kernel (one workgroup per tile and depth; RT_F16P_WALK=0), whatever the segmentation (RT_F16P_WALK=<n>)."""
    n = 2
    dy, hy, wy = ydims
    hx, wx = 2 * hy - 1, 2 * wy - 1
    y = q16(rnd(n, K, dy, hy, wy))
    w = q16(rnd(K, 3, C, 3, 3) * np.float32(1 / np.sqrt(27 * K / 8)))
    b = q16(rnd(C))
    sk = q16(rnd(n, dkeep, C, hx, wx)) if skip else None
    ps = (pad_d, 1, 1)
    ref = O.conv3d_transpose_tf(T(y).double(), T(w).double(), T(b).double(), (dfull, C, hx, wx), (2, 2, 2), ps, ps)[:, :dkeep]
    if skip:
        ref = ref + T(sk).double()
    ref = O.elu(ref)
    if cdhw:
        ref = O.transform(ref)
    ref = ref.numpy()
    tol = 2e-3 * max(1.0, float(np.abs(ref).max()))
    oshape = (n, C // 8, dkeep, hx, wx, 8) if cdhw else (n, dkeep, C // 8, hx, wx, 8)

    def run(walk, classes="1"):
        monkeypatch.setenv("RT_F16P_WALK", walk)
        monkeypatch.setenv("RT_F16P_CLASSES", classes)
        p = backend.klib.conv3d_plan(h16(w), h16(b), C, K, (dfull, hx, wx), (3, 3, 3), (2, 2, 2), ps, ps, act=capi.RT_ACT_ELU, out_dchw=cdhw,
                                     has_residual=skip, dtype=capi.RT_F16, transposed_in_dims=ydims, out_depth=dkeep)
        p.set_io_types(capi.RT_F16, capi.RT_F16)
        p.set_layouts(1, 1, 1 if skip else 0)
        out = empty(backend, oshape, True)
        p.enqueue(dev16(backend, il_cm(y)), out, dev16(backend, il_dm(sk)) if skip else None, n)
        res = host(backend, out)
        p.destroy()
        return res
    per_depth = run("0")
    walk = run(str(nseg) if nseg else "-1")
    assert not np.isnan(walk).any()
    got = un_il_cm(walk) if cdhw else un_il_dm(walk)
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()
    assert np.array_equal(walk, per_depth)
    # (the default takes both depth classes -- even and odd output depths -- in one launch, slice m of the one, then slice m of the other;
    #  RT_F16P_CLASSES=0: a launch per class)
    assert np.array_equal(run(str(nseg) if nseg else "-1", "0"), per_depth)


SMALL_CASES = [
    # K, C, (Dy, Hy, Wy), out depth computed, kept, pad_d, act
    (32, 1, (3, 4, 19), 7, 6, 0, capi.RT_ACT_NONE),        # NVSmall / ResNet-18 3D: 32 channels -> 1, depth pad 0, the surplus slice dropped
    (32, 1, (2, 5, 40), 3, 3, 1, capi.RT_ACT_ELU),         # depth pad 1; 40 blocks across: three 16-block groups, the last one ragged
    (64, 2, (2, 3, 9), 5, 4, 0, capi.RT_ACT_NONE),         # two output channels (rows 8..15 of the MFMA), two blocks of 32 input channels
    (32, 2, (1, 1, 1), 1, 1, 1, capi.RT_ACT_SIGMOID),      # a single voxel
    (32, 1, (17, 3, 21), 35, 34, 0, capi.RT_ACT_NONE),     # 18 depth blocks: the depth walk (deconv3d_s2_ilw_kernel) takes them in two segments
]


@pytest.mark.parametrize("K,C,ydims,dfull,dkeep,pad_d,act", SMALL_CASES)
def test_last_deconv3d_on_interleaved_input(backend, monkeypatch, K, C, ydims, dfull, dkeep, pad_d, act):
    """(K = 32: deconv3d_s2_ilw_kernel, the wave walks down the depth blocks with the shared slice in registers -- it must give the bits of
    deconv3d_s2_il_kernel, one depth block per workgroup, which RT_SMALL_IL_WALK=0 selects)"""
    n = 2
    dy, hy, wy = ydims
    hx, wx = 2 * hy - 1, 2 * wy - 1
    y = q16(rnd(n, K, dy, hy, wy))
    w = q16(rnd(K, 3, C, 3, 3) * np.float32(1 / np.sqrt(27 * K / 8)))
    b = q16(rnd(C))
    ps = (pad_d, 1, 1)
    ref = O.conv3d_transpose_tf(T(y).double(), T(w).double(), T(b).double(), (dfull, C, hx, wx), (2, 2, 2), ps, ps)[:, :dkeep]
    ref = (O.elu(ref) if act == capi.RT_ACT_ELU else torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref).numpy()
    plan = backend.klib.conv3d_plan(h16(w), h16(b), C, K, (dfull, hx, wx), (3, 3, 3), (2, 2, 2), ps, ps, act=act, dtype=capi.RT_F16,
                                    transposed_in_dims=ydims, out_depth=dkeep)
    plan.set_io_types(capi.RT_F16, capi.RT_F32)
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev16(backend, y), out, None, n)
    planar = host(backend, out)
    tol = 2e-4 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(planar - ref).max() <= tol
    assert plan.il_caps() == 1
    plan.set_layouts(1, 0, 0)
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev16(backend, il_cm(y)), out, None, n)
    got = host(backend, out)
    assert not np.isnan(got).any()
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()
    plan.set_layouts(0, 0, 0)
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev16(backend, y), out, None, n)
    assert np.array_equal(host(backend, out), planar)
    plan.destroy()
    monkeypatch.setenv("RT_SMALL_IL_WALK", "0")
    plan = backend.klib.conv3d_plan(h16(w), h16(b), C, K, (dfull, hx, wx), (3, 3, 3), (2, 2, 2), ps, ps, act=act, dtype=capi.RT_F16,
                                    transposed_in_dims=ydims, out_depth=dkeep)
    plan.set_io_types(capi.RT_F16, capi.RT_F32)
    plan.set_layouts(1, 0, 0)
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev16(backend, il_cm(y)), out, None, n)
    assert np.array_equal(host(backend, out), got)
    plan.destroy()


@pytest.mark.parametrize("ydims,dfull,dkeep,pad_d,act,mode,scale", [
    ((3, 4, 19), 7, 6, 0, capi.RT_ACT_NONE, 2, 1.0),       # NVSmall / ResNet-18 3D: soft-argmin over the kept slices (the surplus one must not weigh in)
    ((17, 3, 21), 35, 34, 0, capi.RT_ACT_NONE, 2, 8.0),    # 18 depth blocks, a sharply peaked volume
    ((2, 5, 40), 3, 3, 1, capi.RT_ACT_ELU, 1, 1.0),        # soft-argmax, odd kept depth (the odd parity sees one slice fewer), an activation
    ((9, 2, 3), 19, 18, 0, capi.RT_ACT_NONE, 2, 60.0),     # values far apart: all the weight on one slice, no overflow
])
def test_last_deconv3d_ends_in_softargmax(backend, ydims, dfull, dkeep, pad_d, act, mode, scale):
    """rt_conv_plan_set_softarg: the depth walk of the last Conv3DTranspose keeps an online softmax and writes the (1, H, W) map of
    disp_softargmax (lib/softargmax_plugin.cpp:167-205) -- against the oracle's soft-argmax of the oracle's volume, and against
    rt_softargmax on the volume the unfused plan writes."""
    n, K, C = 2, 32, 1
    dy, hy, wy = ydims
    hx, wx = 2 * hy - 1, 2 * wy - 1
    y = q16(rnd(n, K, dy, hy, wy))
    w = q16(rnd(K, 3, C, 3, 3) * np.float32(scale / np.sqrt(27 * K / 8)))
    b = q16(rnd(C))
    ps = (pad_d, 1, 1)
    vol = O.conv3d_transpose_tf(T(y).double(), T(w).double(), T(b).double(), (dfull, C, hx, wx), (2, 2, 2), ps, ps)[:, :dkeep]
    vol = O.elu(vol) if act == capi.RT_ACT_ELU else vol
    ref = O.softargmax(vol, mode == 2).numpy()                           # (n, 1, hx, wx)
    plan = backend.klib.conv3d_plan(h16(w), h16(b), C, K, (dfull, hx, wx), (3, 3, 3), (2, 2, 2), ps, ps, act=act, dtype=capi.RT_F16,
                                    transposed_in_dims=ydims, out_depth=dkeep)
    plan.set_io_types(capi.RT_F16, capi.RT_F32)
    with pytest.raises(capi.RtError):
        plan.set_softarg(mode)                                           # planar input: no fused form, the plan stays as it is
    plan.set_layouts(1, 0, 0)
    v = empty(backend, vol.shape, False)
    plan.enqueue(dev16(backend, il_cm(y)), v, None, n)
    two = empty(backend, ref.shape, False)
    backend.klib.softargmax(v, two, n, dkeep, hx, wx, mode == 2)
    two = host(backend, two)
    plan.set_softarg(mode)
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev16(backend, il_cm(y)), out, None, n)
    got = host(backend, out)
    assert not np.isnan(got).any()
    # fp16-rounded operands, fp32 accumulation: the volume differs from the float64 one by ~1e-4 relative; the map is an average of indices
    tol = 2e-4 * dkeep * max(1.0, float(np.abs(vol.numpy()).max()))
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()
    assert np.abs(got - two).max() <= 2e-5 * dkeep, np.abs(got - two).max()
    with pytest.raises(capi.RtError):
        plan.set_layouts(1, 0, 1)                                        # refused (no skip tensor): the plan stays as it was, reduction included
    again = empty(backend, ref.shape, False)
    plan.enqueue(dev16(backend, il_cm(y)), again, None, n)
    assert np.array_equal(host(backend, again), got)
    plan.set_softarg(0)                                                  # ... and off again: the volume, bit for bit
    v2 = empty(backend, vol.shape, False)
    plan.enqueue(dev16(backend, il_cm(y)), v2, None, n)
    assert np.array_equal(host(backend, v2), host(backend, v))
    plan.destroy()


def il_2d(a):
    """(N, C, H, W) -> (N, C/8, H, W, 8)"""
    n, c, h, w = a.shape
    return np.ascontiguousarray(a.reshape(n, c // 8, 8, h, w).transpose(0, 1, 3, 4, 2))


R4_CASES = [
    # c, k, d, h, w, channel-major output, residual (0 none, 1 interleaved, 2 planar)
    (16, 32, 3, 20, 37, False, 0),      # two 16-row tiles (the second with 4 rows), two column tiles
    (32, 64, 2, 17, 70, False, 1),      # two blocks of 32 output channels, skip tensor interleaved
    (16, 24, 2, 33, 33, True, 0),       # Cout % 32 != 0, channel-major (K/8, D, H, W, 8) output, one pixel past a column tile
    (8, 32, 4, 5, 9, False, 2),         # image smaller than a tile, one half-empty chunk, planar skip tensor
]


@pytest.mark.parametrize("c,k,d,h,w,cm,resid", R4_CASES)
def test_conv3d_four_rows_per_wave(backend, monkeypatch, c, k, d, h, w, cm, resid):
    """conv_f16r4_kernel (16 x 32 tiles, four output rows per wave, operands reused from registers) against the oracle and against the
    4 x 32-tile kernel it replaces on the Conv3D layers between interleaved fp16 tensors"""
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
    for r4 in ("1", "0"):
        monkeypatch.setenv("RT_F16_R4", r4)
        plan = backend.klib.conv3d_plan(h16(wt), h16(b), c, k, (d, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU,
                                        out_dchw=not cm, has_residual=bool(resid), dtype=capi.RT_F16)
        plan.set_io_types(capi.RT_F16, capi.RT_F16)
        plan.set_layouts(1, 1, 0)
        oshape = (n, k // 8, d, h, w, 8) if cm else (n, d, k // 8, h, w, 8)
        out = empty(backend, oshape, True)
        rin = None
        if resid:
            rin = dev16(backend, (il_cm(res) if cm else il_dm(res)) if resid == 1 else res)
            if resid == 1:
                plan.set_layouts(1, 1, 1) if plan.il_caps() & 4 else pytest.skip("no interleaved residual for this plan")
        plan.enqueue(dev16(backend, il_dm(x)), out, rin, n)
        got = host(backend, out)
        assert not np.isnan(got).any()
        outs.append(un_il_cm(got) if cm else un_il_dm(got))
        plan.destroy()
    tol = 2e-3 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[0] - ref).max() <= tol, np.abs(outs[0] - ref).max()
    assert np.abs(outs[1] - ref).max() <= tol
    assert np.abs(outs[0] - outs[1]).max() <= tol


@pytest.mark.parametrize("f,k,h,w,D", [(8, 16, 18, 37, 6), (32, 32, 13, 40, 12)])
def test_conv3d_four_rows_per_wave_on_folded_cost_volume(backend, monkeypatch, f, k, h, w, D):
    """the first Conv3D of the 3-D models in half2 mode: the two fp16 feature maps, channel-interleaved (2F/8, H, W, 8), the right one read
    shifted by the slice's disparity (rtConv3dDesc::cv_fold), on the four-rows-per-wave kernel"""
    n = 2
    l, r = q16(rnd(n, f, h, w)), q16(rnd(n, f, h, w))
    wt, b = q16(rnd(k, 3, 2 * f, 3, 3) * np.float32(1 / np.sqrt(27 * 2 * f))), q16(rnd(k))
    cv = O.cost_volume(T(l).double(), T(r).double(), D)
    ref = O.elu(O.transform(O.conv3d_tf(cv, T(wt).double(), T(b).double(), (1, 1, 1), (1, 1, 1), (1, 1, 1)))).numpy()
    outs = []
    for r4 in ("1", "0"):
        monkeypatch.setenv("RT_F16_R4", r4)
        plan = backend.klib.conv3d_plan(h16(wt), h16(b), 2 * f, k, (D, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU,
                                        out_dchw=True, cv_fold=f, dtype=capi.RT_F16)
        plan.set_io_types(capi.RT_F16, capi.RT_F16)
        plan.set_layouts(1, 1, 0)
        out = empty(backend, (n, D, k // 8, h, w, 8), True)
        plan.enqueue(dev16(backend, il_2d(np.concatenate([l, r], axis=1))), out, None, n)
        outs.append(un_il_dm(host(backend, out)))
        plan.destroy()
    tol = 2e-3 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[0] - ref).max() <= tol, np.abs(outs[0] - ref).max()
    assert np.abs(outs[1] - ref).max() <= tol


# ---- fp32 engines (config C4): the 3-D tensors between Conv3D launches in groups of FOUR channels, (D, C/4, H, W, 4) --------------------
def il4_dm(a):
    n, d, c, h, w = a.shape
    return np.ascontiguousarray(a.reshape(n, d, c // 4, 4, h, w).transpose(0, 1, 2, 4, 5, 3))


def un_il4_dm(a):
    n, d, g, h, w, _ = a.shape
    return a.transpose(0, 1, 2, 5, 3, 4).reshape(n, d, g * 4, h, w)


def dev32(backend, a):
    a = np.ascontiguousarray(a.astype(np.float32))
    return torch.from_numpy(a).cuda() if backend.name == "gpu" else a


@pytest.mark.parametrize("c,k,d,h,w,stride,resid", [(16, 32, 3, 9, 35, 1, True), (8, 12, 5, 6, 37, 1, False), (32, 64, 4, 7, 33, 2, False),
                                                       (4, 32, 2, 5, 70, 1, True)])
def test_conv3d_fp32_on_interleaved_tensors(backend, c, k, d, h, w, stride, resid):
    """Conv3D of an fp32 engine (split-fp16 arithmetic) with any of input / output / skip tensor as (D, C/4, H, W, 4): the same arithmetic
    in the same order as on planar tensors, so the results must be the SAME BITS; and near the oracle as the planar form is"""
    n = 2
    x = rnd(n, d, c, h, w)
    wt, b = rnd(k, 3, c, 3, 3) * np.float32(1 / np.sqrt(27 * c)), rnd(k)
    even = stride == 2 and d % 2 == 0
    pads = (0, 1, 1) if even else (1, 1, 1)
    xin = O.pad_d(T(x).double(), 1) if even else T(x).double()
    ref = O.transform(O.conv3d_tf(xin, T(wt).double(), T(b).double(), (stride,) * 3, pads, pads))           # (N, Do, K, Ho, Wo)
    res = rnd(*ref.shape) if resid else None
    if resid:
        ref = ref + T(res).double()
    ref = O.elu(ref).numpy()
    plan = backend.klib.conv3d_plan(wt, b, c, k, (d + (1 if even else 0), h, w), (3, 3, 3), (stride,) * 3, pads, pads, act=capi.RT_ACT_ELU,
                                    out_dchw=True, has_residual=resid, in_pad_end=1 if even else 0)
    caps = plan.il_caps()
    assert caps & 1 and caps & 2 and bool(caps & 4) == resid, caps
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev32(backend, x), out, dev32(backend, res) if resid else None, n)
    planar = host(backend, out)
    assert np.abs(planar - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    for xi, yi, ri in [(1, 0, 0), (0, 1, 0), (1, 1, 0)] + ([(0, 1, 1), (1, 1, 1)] if resid else []):
        plan.set_layouts(xi, yi, ri)
        out = empty(backend, il4_dm(ref).shape if yi else ref.shape, False)
        plan.enqueue(dev32(backend, il4_dm(x) if xi else x), out, dev32(backend, il4_dm(res) if ri else res) if resid else None, n)
        got = host(backend, out)
        assert not np.isnan(got).any(), (xi, yi, ri)
        assert np.array_equal(un_il4_dm(got) if yi else got, planar), (xi, yi, ri)
    plan.destroy()


@pytest.mark.parametrize("f,k,h,w,D", [(8, 16, 9, 37, 6), (32, 32, 5, 40, 4), (4, 8, 7, 33, 5)])
def test_conv3d_fp32_folded_cost_volume_on_interleaved_feature_maps(backend, monkeypatch, f, k, h, w, D):
    """the GATHER form of the first Conv3D (the fallback of the factored form, RT_NO_FOLD_FACTOR=1 here) reads the two feature maps
    [left | right] as (2F/4, H, W, 4): the right half shifted by the slice's disparity in whole 16-byte slots.  Same bits as on planar maps."""
    monkeypatch.setenv("RT_NO_FOLD_FACTOR", "1")
    n = 2
    l, r = rnd(n, f, h, w), rnd(n, f, h, w)
    wt, b = rnd(k, 3, 2 * f, 3, 3) * np.float32(1 / np.sqrt(27 * 2 * f)), rnd(k)
    cv = O.cost_volume(T(l).double(), T(r).double(), D)
    ref = O.elu(O.transform(O.conv3d_tf(cv, T(wt).double(), T(b).double(), (1, 1, 1), (1, 1, 1), (1, 1, 1)))).numpy()
    plan = backend.klib.conv3d_plan(wt, b, 2 * f, k, (D, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU,
                                    out_dchw=True, cv_fold=f)
    assert plan.il_caps() & 3 == 3
    x = np.concatenate([l, r], axis=1)                                   # (N, 2F, H, W)
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev32(backend, x), out, None, n)
    planar = host(backend, out)
    assert np.abs(planar - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    xil = np.ascontiguousarray(x.reshape(n, 2 * f // 4, 4, h, w).transpose(0, 1, 3, 4, 2))
    for yi in (0, 1):
        plan.set_layouts(1, yi, 0)
        out = empty(backend, il4_dm(ref).shape if yi else ref.shape, False)
        plan.enqueue(dev32(backend, xil), out, None, n)
        got = host(backend, out)
        assert np.array_equal(un_il4_dm(got) if yi else got, planar), yi
    plan.destroy()


FOLD_CASES = [
    # F, K, H, W, D
    (8, 16, 9, 37, 6),
    (32, 32, 5, 40, 12),
    (4, 8, 7, 33, 5),
    (8, 8, 4, 5, 9),          # more disparities than columns: most of the right half is masked (x < d)
    (16, 24, 3, 70, 3),       # the minimum depth: first / middle / last slice and nothing else
]


@pytest.mark.parametrize("f,k,h,w,D", FOLD_CASES)
def test_conv3d_on_folded_cost_volume_factored(backend, monkeypatch, f, k, h, w, D):
    """fold_factor.hip.h: the first Conv3D over the default cost volume as two 2-D convolutions of the feature maps + one combining pass
    (a convolution commutes with the shift that builds the volume; the volume's right edge does not: the edge term) -- against the oracle's
    cost_volume + conv3d_tf in fp64, against the gather form, and in all four output forms"""
    n = 2
    l, r = rnd(n, f, h, w), rnd(n, f, h, w)
    wt, b = rnd(k, 3, 2 * f, 3, 3) * np.float32(1 / np.sqrt(27 * 2 * f)), rnd(k)
    cv = O.cost_volume(T(l).double(), T(r).double(), D)
    ref = O.elu(O.transform(O.conv3d_tf(cv, T(wt).double(), T(b).double(), (1, 1, 1), (1, 1, 1), (1, 1, 1)))).numpy()      # (N, D, K, H, W)
    x = np.concatenate([l, r], axis=1)
    tol = 2e-5 * max(1.0, float(np.abs(ref).max()))

    def make():
        return backend.klib.conv3d_plan(wt, b, 2 * f, k, (D, h, w), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), act=capi.RT_ACT_ELU, out_dchw=True, cv_fold=f)
    plan = make()
    caps = plan.il_caps()
    assert caps & 32 and caps & 2 and not caps & 1, caps                  # the factored form: planar fp32 maps in, any output
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev32(backend, x), out, None, n)
    fact = host(backend, out)
    assert not np.isnan(fact).any()
    assert np.abs(fact - ref).max() <= tol, np.abs(fact - ref).max()
    plan.set_layouts(0, 1, 0)                                               # (D, K/4, H, W, 4)
    out = empty(backend, il4_dm(ref).shape, False)
    plan.enqueue(dev32(backend, x), out, None, n)
    assert np.array_equal(un_il4_dm(host(backend, out)), fact)
    plan.set_layouts(0, 0, 0)
    plan.set_io_types(capi.RT_F32, capi.RT_F16)                             # half2 mode: fp32 maps in, fp16 volume out
    out = empty(backend, ref.shape, True)
    plan.enqueue(dev32(backend, x), out, None, n)
    f16 = host(backend, out)
    assert np.array_equal(f16, fact.astype(np.float16).astype(np.float32))  # the same sums, one rounding
    plan.set_layouts(0, 1, 0)                                               # (D, K/8, H, W, 8)
    out = empty(backend, il_dm(ref).shape, True)
    plan.enqueue(dev32(backend, x), out, None, n)
    assert np.array_equal(un_il_dm(host(backend, out)), f16)
    plan.destroy()
    monkeypatch.setenv("RT_NO_FOLD_FACTOR", "1")                            # the gather form: every product of the volume
    plan = make()
    assert not plan.il_caps() & 32
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev32(backend, x), out, None, n)
    gather = host(backend, out)
    plan.destroy()
    assert np.abs(gather - ref).max() <= tol
    assert np.abs(gather - fact).max() <= tol


@pytest.mark.parametrize("K,C,ydims,dfull,dkeep,pad_d", [(16, 8, (3, 4, 19), 7, 6, 0), (8, 16, (3, 5, 9), 5, 5, 1)])
def test_conv3d_transpose_fp32_reads_an_interleaved_skip_tensor(backend, K, C, ydims, dfull, dkeep, pad_d):
    n = 2
    dy, hy, wy = ydims
    hx, wx = 2 * hy - 1, 2 * wy - 1
    y, w, b = rnd(n, K, dy, hy, wy), rnd(K, 3, C, 3, 3) * np.float32(1 / np.sqrt(27 * K / 8)), rnd(C)
    sk = rnd(n, dkeep, C, hx, wx)
    ps = (pad_d, 1, 1)
    plan = backend.klib.conv3d_plan(w, b, C, K, (dfull, hx, wx), (3, 3, 3), (2, 2, 2), ps, ps, act=capi.RT_ACT_ELU, out_dchw=False,
                                    has_residual=True, transposed_in_dims=ydims, out_depth=dkeep)
    assert plan.il_caps() == 6                            # the skip tensor read interleaved, the output written interleaved (four-phase form)
    out = empty(backend, (n, dkeep, C, hx, wx), False)
    plan.enqueue(dev32(backend, y), out, dev32(backend, sk), n)
    planar = host(backend, out)
    ref = O.elu(O.conv3d_transpose_tf(T(y).double(), T(w).double(), T(b).double(), (dfull, C, hx, wx), (2, 2, 2), ps, ps)[:, :dkeep] + T(sk).double()).numpy()
    assert np.abs(planar - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    plan.set_layouts(0, 0, 1)
    out = empty(backend, (n, dkeep, C, hx, wx), False)
    plan.enqueue(dev32(backend, y), out, dev32(backend, il4_dm(sk)), n)
    assert np.array_equal(host(backend, out), planar)
    plan.destroy()


P4F32_CASES = [
    # K, C, (Dy, Hy, Wy), out depth computed (full), kept, pad_d, skip (0 none, 1 planar, 2 interleaved), cdhw
    (16, 8, (3, 4, 19), 7, 6, 0, 2, True),        # NVSmall's decoder pattern: depth pad 0, surplus slice dropped, fused Transform
    (16, 8, (3, 4, 19), 7, 6, 0, 1, False),       # planar skip tensor, no Transform
    (8, 16, (3, 5, 9), 5, 5, 1, 2, True),         # depth pad 1 (odd depth), K = 8: a half-empty chunk
    (24, 40, (2, 3, 17), 3, 3, 1, 0, True),       # no skip tensor, two blocks of output channels (the second with 8), K not a multiple of 16
    (32, 32, (2, 6, 33), 5, 4, 0, 2, False),      # two 32-pixel tiles across (the second holds one column: the unpaired last output pixel), 2 row tiles
    (5, 3, (1, 1, 1), 1, 1, 1, 1, False),         # a single voxel, odd channel counts
]


@pytest.mark.parametrize("K,C,ydims,dfull,dkeep,pad_d,skip,cdhw", P4F32_CASES)
def test_conv3d_transpose_fp32_four_phases_per_workgroup(backend, monkeypatch, K, C, ydims, dfull, dkeep, pad_d, skip, cdhw):
    """deconv_s3p_kernel (fp32 tensors, 3-term fp16 split, all four output phases of a 4 x 32 input tile per workgroup, 8-byte stores of
    pixel pairs at 4-byte aligned addresses) against the oracle and -- bit for bit -- against the one-launch-per-phase form it replaces"""
    if skip == 2 and C % 4:
        pytest.skip("interleaved skip tensor needs whole groups")
    n = 2
    dy, hy, wy = ydims
    hx, wx = 2 * hy - 1, 2 * wy - 1
    y, w, b = rnd(n, K, dy, hy, wy), rnd(K, 3, C, 3, 3) * np.float32(1 / np.sqrt(27 * K / 8)), rnd(C)
    sk = rnd(n, dkeep, C, hx, wx) if skip else None
    ps = (pad_d, 1, 1)
    ref = O.conv3d_transpose_tf(T(y).double(), T(w).double(), T(b).double(), (dfull, C, hx, wx), (2, 2, 2), ps, ps)[:, :dkeep]
    if skip:
        ref = ref + T(sk).double()
    ref = O.elu(ref)
    if cdhw:
        ref = O.transform(ref)
    ref = ref.numpy()
    outs = []
    monkeypatch.setenv("RT_S3_KSPLIT", "0")         # (the phase form would split the contraction of so small a launch over wave groups: another order)
    for p4 in ("0", "1"):
        monkeypatch.setenv("RT_NO_DECONV_P4F32", p4)
        plan = backend.klib.conv3d_plan(w, b, C, K, (dfull, hx, wx), (3, 3, 3), (2, 2, 2), ps, ps, act=capi.RT_ACT_ELU, out_dchw=cdhw,
                                        has_residual=bool(skip), transposed_in_dims=ydims, out_depth=dkeep)
        if skip == 2:
            assert plan.il_caps() == (4 if p4 == "1" else 6)
            plan.set_layouts(0, 0, 1)
        out = empty(backend, ref.shape, False)
        plan.enqueue(dev32(backend, y), out, dev32(backend, il4_dm(sk) if skip == 2 else sk) if skip else None, n)
        got = host(backend, out)
        assert not np.isnan(got).any()
        outs.append(got)
        plan.destroy()
    tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[0] - ref).max() <= tol, np.abs(outs[0] - ref).max()
    assert np.abs(outs[1] - ref).max() <= tol
    assert np.array_equal(outs[0], outs[1])              # the same arithmetic, operation for operation


def il4_cm(a):
    """channel-major (N, K, D, H, W) -> (N, K/4, D, H, W, 4)"""
    n, k, d, h, w = a.shape
    return np.ascontiguousarray(a.reshape(n, k // 4, 4, d, h, w).transpose(0, 1, 3, 4, 5, 2))


def un_il4_cm(a):
    n, g, d, h, w, _ = a.shape
    return a.transpose(0, 1, 5, 2, 3, 4).reshape(n, g * 4, d, h, w)


@pytest.mark.parametrize("K,C,ydims,dfull,dkeep,pad_d,skip,cdhw", [(16, 8, (3, 4, 19), 7, 6, 0, True, True), (8, 16, (3, 5, 9), 5, 5, 1, True, False),
                                                                  (24, 40, (2, 3, 17), 3, 3, 1, False, True), (32, 32, (2, 6, 33), 5, 4, 0, True, True)])
def test_conv3d_transpose_fp32_writes_an_interleaved_tensor(backend, K, C, ydims, dfull, dkeep, pad_d, skip, cdhw):
    """deconv_s3p_kernel<true>: the same arithmetic, the output as (C/4, D, H, W, 4) (fused Transform) or (D, C/4, H, W, 4) -- the tensor the
    last layer of an fp32 3-D engine reads on the matrix cores.  Same bits as the planar output."""
    n = 2
    dy, hy, wy = ydims
    hx, wx = 2 * hy - 1, 2 * wy - 1
    y, w, b = rnd(n, K, dy, hy, wy), rnd(K, 3, C, 3, 3) * np.float32(1 / np.sqrt(27 * K / 8)), rnd(C)
    sk = rnd(n, dkeep, C, hx, wx) if skip else None
    ps = (pad_d, 1, 1)
    plan = backend.klib.conv3d_plan(w, b, C, K, (dfull, hx, wx), (3, 3, 3), (2, 2, 2), ps, ps, act=capi.RT_ACT_ELU, out_dchw=cdhw,
                                    has_residual=skip, transposed_in_dims=ydims, out_depth=dkeep)
    assert plan.il_caps() == (2 | (4 if skip else 0))
    shape = (n, C, dkeep, hx, wx) if cdhw else (n, dkeep, C, hx, wx)
    out = empty(backend, shape, False)
    plan.enqueue(dev32(backend, y), out, dev32(backend, sk) if skip else None, n)
    planar = host(backend, out)
    plan.set_layouts(0, 1, 1 if skip else 0)
    out = empty(backend, (il4_cm if cdhw else il4_dm)(planar).shape, False)
    plan.enqueue(dev32(backend, y), out, dev32(backend, il4_dm(sk)) if skip else None, n)
    got = host(backend, out)
    assert not np.isnan(got).any()
    assert np.array_equal((un_il4_cm if cdhw else un_il4_dm)(got), planar)
    with pytest.raises(capi.RtError):
        plan.set_layouts(1, 1, 0)                       # the input stays planar
    plan.destroy()


@pytest.mark.parametrize("K,C,ydims,dfull,dkeep,pad_d,act", SMALL_CASES)
def test_last_deconv3d_fp32_on_interleaved_input(backend, K, C, ydims, dfull, dkeep, pad_d, act):
    """deconv3d_s2_il4_kernel: the last layer of an fp32 3-D engine with its input as (K/4, Dy, Hy, Wy, 4), 3-term fp16 split on
    v_mfma_f32_16x16x32_f16 -- against the oracle in fp64 and against the vector-ALU fp32 kernel on the planar tensor"""
    n = 2
    dy, hy, wy = ydims
    hx, wx = 2 * hy - 1, 2 * wy - 1
    y, w, b = rnd(n, K, dy, hy, wy), rnd(K, 3, C, 3, 3) * np.float32(1 / np.sqrt(27 * K / 8)), rnd(C)
    ps = (pad_d, 1, 1)
    ref = O.conv3d_transpose_tf(T(y).double(), T(w).double(), T(b).double(), (dfull, C, hx, wx), (2, 2, 2), ps, ps)[:, :dkeep]
    ref = (O.elu(ref) if act == capi.RT_ACT_ELU else torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref).numpy()
    plan = backend.klib.conv3d_plan(w, b, C, K, (dfull, hx, wx), (3, 3, 3), (2, 2, 2), ps, ps, act=act, transposed_in_dims=ydims, out_depth=dkeep)
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev32(backend, y), out, None, n)
    planar = host(backend, out)
    tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(planar - ref).max() <= tol
    assert plan.il_caps() == 1
    plan.set_layouts(1, 0, 0)
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev32(backend, il4_cm(y)), out, None, n)
    got = host(backend, out)
    assert not np.isnan(got).any()
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()            # fp32-class accuracy, as the fp32 fma chain
    plan.set_layouts(0, 0, 0)
    out = empty(backend, ref.shape, False)
    plan.enqueue(dev32(backend, y), out, None, n)
    assert np.array_equal(host(backend, out), planar)
    plan.destroy()
