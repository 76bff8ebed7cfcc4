"""Parity of the MFMA implicit-GEMM convolutions (2-D conv/deconv, Conv3D, Conv3DTranspose) through the
C ABI: the reference's golden Conv3D / Conv3DTranspose tensors (tests_main.cpp:345-878) plus seeded
random cases against the oracle.  Runs on the SIMT emulator (CPU tier) and on the GPU (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi
from test_ops_parity import T, near, rnd, float_eq


# ---- 2-D convolution -----------------------------------------------------------------------------
CONV2D_CASES = [
    # cin, cout, h, w, k, stride, pad, act, resid, batch
    (5, 7, 6, 21, 3, 1, 1, capi.RT_ACT_NONE, False, 1),
    (32, 32, 9, 70, 3, 1, 1, capi.RT_ACT_ELU, True, 2),
    (33, 32, 5, 33, 3, 1, 1, capi.RT_ACT_ELU, False, 1),
    (8, 70, 7, 19, 3, 2, 1, capi.RT_ACT_ELU, False, 1),
    (3, 32, 17, 41, 5, 2, 2, capi.RT_ACT_ELU, False, 2),
    (16, 40, 4, 9, 3, 1, 1, capi.RT_ACT_SIGMOID, False, 1),
]


@pytest.mark.parametrize("cin,cout,h,w,k,stride,pad,act,resid,batch", CONV2D_CASES)
def test_conv2d(backend, cin, cout, h, w, k, stride, pad, act, resid, batch):
    x, wt, b = rnd(batch, cin, h, w), rnd(cout, cin, k, k) * np.float32(1 / np.sqrt(cin * k * k)), rnd(cout)
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, k, stride, pad, act=act, has_residual=resid)
    co, ho, wo, _ = plan.out_dims
    ref = O.conv2d(T(x), T(wt), T(b), stride, pad)
    assert (co, ho, wo) == tuple(ref.shape[1:])
    res = rnd(*ref.shape) if resid else None
    if resid:
        ref = ref + T(res)
    ref = O.elu(ref) if act == capi.RT_ACT_ELU else (torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref)
    y = backend.empty(ref.shape)
    plan.enqueue(backend.dev(x), y, backend.dev(res) if resid else None, batch)
    near(backend.host(y), ref.numpy(), 2e-5)
    plan.destroy()


DECONV2D_CASES = [
    (6, 5, 4, 9, 1, capi.RT_ACT_NONE, False, 1),
    (32, 1, 5, 35, 1, capi.RT_ACT_SIGMOID, False, 2),
    (16, 40, 6, 7, 1, capi.RT_ACT_ELU, True, 1),
    (4, 3, 3, 5, 0, capi.RT_ACT_NONE, False, 1),
]


@pytest.mark.parametrize("cin,cout,h,w,pad,act,resid,batch", DECONV2D_CASES)
def test_deconv2d(backend, cin, cout, h, w, pad, act, resid, batch):
    x, wt, b = rnd(batch, cin, h, w), rnd(cin, cout, 3, 3) * np.float32(1 / np.sqrt(cin * 9 / 4)), rnd(cout)
    plan = backend.klib.conv2d_plan(wt, b, cin, cout, h, w, 3, 2, pad, act=act, has_residual=resid, transposed=True)
    ref = O.deconv2d(T(x), T(wt), T(b), 2, pad)
    assert tuple(plan.out_dims[:3]) == tuple(ref.shape[1:])
    res = rnd(*ref.shape) if resid else None
    if resid:
        ref = ref + T(res)
    ref = O.elu(ref) if act == capi.RT_ACT_ELU else (torch.sigmoid(ref) if act == capi.RT_ACT_SIGMOID else ref)
    y = backend.empty(ref.shape)
    plan.enqueue(backend.dev(x), y, backend.dev(res) if resid else None, batch)
    near(backend.host(y), ref.numpy(), 2e-5)
    plan.destroy()


# ---- Conv3D plugin semantics against the reference's golden tensors ---------------------------------
def conv3d_run(backend, x, w, b, stride, pad_start, pad_end, out_dchw=False, act=0):
    n, d, c, h, ww = x.shape
    k, v = w.shape[0], w.shape[1]
    plan = backend.klib.conv3d_plan(w, b, c, k, (d, h, ww), (v, w.shape[3], w.shape[4]), stride, pad_start, pad_end,
                                    act=act, out_dchw=out_dchw)
    y = backend.empty((n,) + plan.out_dims)
    plan.enqueue(backend.dev(x), y, None, n)
    out = backend.host(y)
    plan.destroy()
    return out


def transform(backend, x):
    n, a, b, c, d = x.shape
    y = backend.empty((n, b, a, c, d))
    backend.klib.permute4d(backend.dev(x), y, n, (a, b, c, d), (1, 0, 2, 3))
    return backend.host(y)


def pad1(backend, x):
    n, d = x.shape[:2]
    inner = int(np.prod(x.shape[2:]))
    y = backend.empty((n, d + 1) + x.shape[2:])
    backend.klib.pad_d(backend.dev(x), y, n, d, inner, 1)
    return backend.host(y)


def test_conv3d_01_basic(backend, golden):                       # tests_main.cpp:362-389
    g = golden
    y = conv3d_run(backend, g["conv3d_01_x"], g["conv3d_01_w"], None, (1, 1, 1), (0, 0, 0), (0, 0, 0))
    float_eq(transform(backend, y), g["conv3d_01_y"], ulps=4)           # EXPECT_FLOAT_EQ (tests_main.cpp:388); measured: oracle 1 ULP from the golden value


def test_conv3d_02_hw_strides(backend, golden):                  # :391-420
    g = golden
    y = conv3d_run(backend, g["conv3d_02_x"], g["conv3d_02_w"], None, (1, 2, 2), (0, 1, 1), (0, 1, 1))
    near(transform(backend, y), g["conv3d_02_y"], 1e-5)


def test_conv3d_03_dhw(backend, golden):                         # :422-455
    g = golden
    y = conv3d_run(backend, pad1(backend, g["conv3d_03_x"]), g["conv3d_03_w"], None, (1, 2, 2), (0, 1, 1), (0, 1, 1))
    near(transform(backend, y), g["conv3d_03_y"], 1e-5)


def test_conv3d_04_unit_sym(backend, golden):                    # :457-486
    g = golden
    y = conv3d_run(backend, g["conv3d_04_x"], g["conv3d_04_w"], None, (1, 1, 1), (1, 1, 1), (1, 1, 1))
    near(transform(backend, y), g["conv3d_04_y"], 1e-4)


def test_conv3d_04_fused_dchw_output(backend, golden):
    """out_dchw writes (Do,K,Ho,Wo) directly -- the Transform plugin elided by the executor"""
    g = golden
    y = conv3d_run(backend, g["conv3d_04_x"], g["conv3d_04_w"], None, (1, 1, 1), (1, 1, 1), (1, 1, 1), out_dchw=True)
    near(y, g["conv3d_04_y"], 1e-4)


def test_conv3d_05_asym(backend, golden):                        # :488-521
    g = golden
    y = conv3d_run(backend, pad1(backend, g["conv3d_05_x"]), g["conv3d_05_w"], None, (2, 2, 2), (0, 1, 1), (1, 1, 1))
    near(transform(backend, y), g["conv3d_05_y"], 1e-4)


def test_conv3d_06_bias_elu(backend, golden):                    # :523-570
    g = golden
    y = conv3d_run(backend, pad1(backend, g["conv3d_06_x"]), g["conv3d_06_w"], g["conv3d_06_b"], (2, 2, 2), (0, 1, 1),
                   (1, 1, 1))
    t = transform(backend, y)
    out = backend.empty(t.shape)
    backend.klib.elu(backend.dev(t), out, t.size)
    near(backend.host(out), g["conv3d_06_y"], 1e-4)
    # fused variant: bias + ELU + DCHW output in the conv epilogue
    y2 = conv3d_run(backend, pad1(backend, g["conv3d_06_x"]), g["conv3d_06_w"], g["conv3d_06_b"], (2, 2, 2),
                    (0, 1, 1), (1, 1, 1), out_dchw=True, act=capi.RT_ACT_ELU)
    near(y2, g["conv3d_06_y"], 1e-4)


def test_conv3d_07_multiple(backend, golden):                    # :572-623
    g = golden
    y1 = transform(backend, conv3d_run(backend, g["conv3d_07_x"], g["conv3d_07_w"], None, (1, 1, 1), (1, 1, 1), (1, 1, 1)))
    y2 = conv3d_run(backend, pad1(backend, y1), g["conv3d_07_w"], None, (2, 2, 2), (0, 1, 1), (0, 1, 1))
    near(transform(backend, y2), g["conv3d_07_y"], 1e-4)


def test_conv3d_random_nvtiny_like(backend):
    """16 -> 16 channels, 3x3x3, like conv3D_1 of NVTiny (nvtiny_513x161_net.cpp:173-180), odd sizes"""
    x, w, b = rnd(1, 5, 16, 6, 37), rnd(16, 3, 16, 3, 3) * np.float32(1 / np.sqrt(27 * 16)), rnd(16)
    y = conv3d_run(backend, x, w, b, (1, 1, 1), (1, 1, 1), (1, 1, 1))
    near(y, O.conv3d_tf(T(x), T(w), T(b), (1, 1, 1), (1, 1, 1), (1, 1, 1)).numpy(), 2e-5)


# ---- Conv3DTranspose plugin semantics -----------------------------------------------------------------
def conv3d_tran_run(backend, y, w, b, out_dims, stride, pad_start, pad_end, act=0):
    n, k, dy, hy, wy = y.shape
    dx, c, hx, wx = out_dims
    plan = backend.klib.conv3d_plan(w, b, c, k, (dx, hx, wx), (w.shape[1], w.shape[3], w.shape[4]), stride, pad_start,
                                    pad_end, act=act, transposed_in_dims=(dy, hy, wy))
    assert plan.out_dims == tuple(out_dims)
    x = backend.empty((n,) + tuple(out_dims))
    plan.enqueue(backend.dev(y), x, None, n)
    out = backend.host(x)
    plan.destroy()
    return out


def slice_last(backend, x):
    n, d = x.shape[:2]
    inner = int(np.prod(x.shape[2:]))
    y = backend.empty((n, d - 1) + x.shape[2:])
    backend.klib.slice_d(backend.dev(x), y, n, d, inner, 0, d - 1)
    return backend.host(y)


def test_conv3d_tran_01_basic(backend, golden):                  # tests_main.cpp:653-683
    y, w, x = (golden["conv3d_tran_01_" + k] for k in "ywx")
    out = conv3d_tran_run(backend, y, w, None, x.shape[1:], (1, 1, 1), (0, 0, 0), (0, 0, 0))
    float_eq(transform(backend, out), x)


def test_conv3d_tran_02_hw(backend, golden):                     # :685-715
    y, w, x = (golden["conv3d_tran_02_" + k] for k in "ywx")
    out = conv3d_tran_run(backend, y, w, None, x.shape[1:], (1, 2, 2), (0, 1, 1), (0, 1, 1))
    near(transform(backend, out).reshape(x.shape), x, 1e-4)


def tran_sliced(backend, y, w, b, x_shape, act=0):
    od = (x_shape[1] + 1,) + tuple(x_shape[2:])
    return slice_last(backend, conv3d_tran_run(backend, y, w, b, od, (2, 2, 2), (0, 1, 1), (0, 1, 1), act=act))


def test_conv3d_tran_03_asym(backend, golden):                   # :717-761
    y, w, x = (golden["conv3d_tran_03_" + k] for k in "ywx")
    near(tran_sliced(backend, y, w, None, x.shape), x, 1e-4)


def test_conv3d_tran_04_bias_elu(backend, golden):               # :763-817
    y, w, x = (golden["conv3d_tran_04_" + k] for k in "ywx")
    near(tran_sliced(backend, y, w, golden["conv3d_tran_04_b"], x.shape, act=capi.RT_ACT_ELU), x, 1e-4)


def test_conv3d_tran_05_multiple(backend, golden):               # :819-878
    g = golden
    x1 = tran_sliced(backend, g["conv3d_tran_05_y"], g["conv3d_tran_05_w1"], None, (1, 8, 8, 9, 9))
    x1 = transform(backend, x1)
    x2 = tran_sliced(backend, x1, g["conv3d_tran_05_w2"], None, g["conv3d_tran_05_x"].shape)
    near(x2, g["conv3d_tran_05_x"], 1e-4)


def test_conv3d_tran_symmetric_pad(backend):
    """odd output depth -> symmetric (1,1,1) pads, no Slice (ResNet-18 deconv3D_1, resnet18_1025x321_net.cpp:888-891)"""
    y, w, b = rnd(1, 8, 3, 4, 6), rnd(8, 3, 5, 3, 3) * np.float32(1 / np.sqrt(27 * 8 / 8)), rnd(5)
    out = conv3d_tran_run(backend, y, w, b, (5, 5, 7, 11), (2, 2, 2), (1, 1, 1), (1, 1, 1))
    ref = O.conv3d_transpose_tf(T(y), T(w), T(b), (5, 5, 7, 11), (2, 2, 2), (1, 1, 1), (1, 1, 1)).numpy()
    near(out, ref, 2e-5)


SMALL_DECONV_CASES = [
    # K, C, (Dy,Hy,Wy), out (Dx,Hx,Wx), pad_start, pad_end
    (8, 1, (3, 4, 6), (7, 7, 11), (0, 1, 1), (0, 1, 1)),        # the models' last layer: asymmetric D (then Slice)
    (32, 1, (2, 5, 37), (3, 9, 73), (1, 1, 1), (1, 1, 1)),      # symmetric pads, odd output dims, 2 workgroups in x
    (6, 2, (3, 3, 5), (7, 5, 9), (0, 1, 1), (0, 1, 1)),         # two output channels
]


@pytest.mark.parametrize("k,c,ydims,xdims,ps,pe", SMALL_DECONV_CASES)
@pytest.mark.parametrize("batch", [1, 2])
def test_conv3d_tran_small_output(backend, monkeypatch, k, c, ydims, xdims, ps, pe, batch):
    """deconv3d_s2_small_kernel (<= 2 output channels, 2x2x2 output block per thread) vs the oracle and vs the
    generic per-phase path on the same inputs (RT_NO_DECONV3D_SMALL)"""
    y, w, b = rnd(batch, k, *ydims), rnd(k, 3, c, 3, 3) * np.float32(1 / np.sqrt(27 * k / 8)), rnd(c)
    od = (xdims[0], c, xdims[1], xdims[2])
    ref = O.elu(O.conv3d_transpose_tf(T(y), T(w), T(b), od, (2, 2, 2), ps, pe)).numpy()
    out = conv3d_tran_run(backend, y, w, b, od, (2, 2, 2), ps, pe, act=capi.RT_ACT_ELU)
    # with the reference's padding only 27 of the 64 (phase, neighbour) products carry a tap and the plan drops the rest
    # (rt::SmallTaps); the products it drops are exact zeros, so the dense form gives the same bits
    monkeypatch.setenv("RT_NO_SMALL_SPARSE", "1")
    dense = conv3d_tran_run(backend, y, w, b, od, (2, 2, 2), ps, pe, act=capi.RT_ACT_ELU)
    monkeypatch.delenv("RT_NO_SMALL_SPARSE")
    assert np.array_equal(out, dense)
    monkeypatch.setenv("RT_NO_DECONV3D_SMALL", "1")
    gen = conv3d_tran_run(backend, y, w, b, od, (2, 2, 2), ps, pe, act=capi.RT_ACT_ELU)
    near(out, ref, 2e-5)
    near(gen, ref, 2e-5)


@pytest.mark.parametrize("cdhw", [False, True])
@pytest.mark.parametrize("c", [1, 5, 32])
def test_conv3d_tran_fused_epilogue(backend, c, cdhw):
    """Conv3DTranspose + Slice[0,d) + skip add + ELU (+ Transform {1,0,2,3}) in one launch: the decoder pattern of
    the 3-D models (nvsmall_1025x321_net.cpp:331-398).  The skip tensor stays (D,C,H,W) whatever the output layout."""
    n, k, ydims, dfull = 2, 8, (3, 4, 19), 7
    dkeep = dfull - 1
    y, w, b = rnd(n, k, *ydims), rnd(k, 3, c, 3, 3) * np.float32(1 / np.sqrt(27 * k / 8)), rnd(c)
    od = (dfull, c, 7, 37)
    skip = rnd(n, dkeep, c, 7, 37)
    ref = O.conv3d_transpose_tf(T(y), T(w), T(b), od, (2, 2, 2), (0, 1, 1), (0, 1, 1))[:, :dkeep]
    ref = O.elu(ref + T(skip))
    if cdhw:
        ref = O.transform(ref)
    plan = backend.klib.conv3d_plan(w, b, c, k, (dfull, 7, 37), (3, 3, 3), (2, 2, 2), (0, 1, 1), (0, 1, 1),
                                    act=capi.RT_ACT_ELU, out_dchw=cdhw, has_residual=True, transposed_in_dims=ydims,
                                    out_depth=dkeep)
    assert plan.out_dims == ((c, dkeep, 7, 37) if cdhw else (dkeep, c, 7, 37))
    out = backend.empty(ref.shape)
    plan.enqueue(backend.dev(y), out, backend.dev(skip), n)
    near(backend.host(out), ref.numpy(), 2e-5)
    plan.destroy()


def test_conv_rejects_bad_descriptors(backend):
    w = rnd(4, 4, 7, 7)
    with pytest.raises(capi.RtError):
        backend.klib.conv2d_plan(w, None, 4, 4, 8, 8, 7, 1, 3)          # 7x7 window is not on the hot path
    with pytest.raises(capi.RtError):                                   # H/W pad must be symmetric (conv3d_plugin.cpp:43-46)
        backend.klib.conv3d_plan(rnd(2, 3, 2, 3, 3), None, 2, 2, (4, 5, 5), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 2, 1))
