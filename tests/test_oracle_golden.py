"""Pins oracle/stereo_oracle.py to the reference's own golden tensors.

One test per TEST() in /root/reference/stereoDNN/tests/tests_main.cpp (line numbers cited),
with the reference's own tolerance: EXPECT_FLOAT_EQ == 4 ULP, EXPECT_NEAR otherwise.
"""
import os

import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def assert_float_eq(actual, expected, ulps=4):
    a = np.asarray(actual, np.float32).ravel()
    e = np.asarray(expected, np.float32).ravel()
    assert a.shape == e.shape
    tol = ulps * np.spacing(np.maximum(np.abs(a), np.abs(e)).astype(np.float32))
    bad = np.abs(a - e) > tol
    assert not bad.any(), "max |diff| %g at %d" % (np.abs(a - e).max(), int(np.argmax(np.abs(a - e))))


def assert_near(actual, expected, tol):
    a = np.asarray(actual, np.float32)
    e = np.asarray(expected, np.float32)
    assert a.shape == e.shape, (a.shape, e.shape)
    assert np.abs(a - e).max() <= tol, np.abs(a - e).max()


# ---- ELU (tests_main.cpp:280-342) -------------------------------------------------------
@pytest.mark.parametrize("idx", ["01", "02"])
def test_elu(golden, idx):
    assert_float_eq(O.elu(T(golden["elu_i_" + idx])), golden["elu_o_" + idx])


# ---- Conv3D (tests_main.cpp:345-623); fixtures: x NDCHW, w KVCRS, y NDKHW ------------------
def run_conv3d(g, idx, stride, pad_start, pad_end, pad_input_d=False, bias=None):
    x, w = T(g["conv3d_%s_x" % idx]), T(g["conv3d_%s_w" % idx])
    if pad_input_d:                      # "manually pad input in D dimension" (tests_main.cpp:435-438)
        x = O.pad_d(x, 1)
    y = O.conv3d_tf(x, w, bias, stride, pad_start, pad_end)       # N K D H W
    return O.transform(y)                                            # Transform {1,0,2,3}: KDHW -> DKHW


def test_conv3d_01_basic(golden):                       # :362-389
    assert_float_eq(run_conv3d(golden, "01", (1, 1, 1), (0, 0, 0), (0, 0, 0)), golden["conv3d_01_y"], ulps=4)      # EXPECT_FLOAT_EQ = 4 ULP; measured 1 ULP (the golden value itself is 0.94 ULP from fp64)


def test_conv3d_02_hw_strides(golden):                  # :391-420
    assert_near(run_conv3d(golden, "02", (1, 2, 2), (0, 1, 1), (0, 1, 1)), golden["conv3d_02_y"], 1e-5)


def test_conv3d_03_dhw_strides(golden):                 # :422-455
    assert_near(run_conv3d(golden, "03", (1, 2, 2), (0, 1, 1), (0, 1, 1), pad_input_d=True),
                golden["conv3d_03_y"], 1e-5)


def test_conv3d_04_unit_sym(golden):                    # :457-486
    assert_near(run_conv3d(golden, "04", (1, 1, 1), (1, 1, 1), (1, 1, 1)), golden["conv3d_04_y"], 1e-4)


def test_conv3d_05_asym(golden):                        # :488-521
    assert_near(run_conv3d(golden, "05", (2, 2, 2), (0, 1, 1), (1, 1, 1), pad_input_d=True),
                golden["conv3d_05_y"], 1e-4)


def test_conv3d_06_bias_elu(golden):                    # :523-570
    y = run_conv3d(golden, "06", (2, 2, 2), (0, 1, 1), (1, 1, 1), pad_input_d=True,
                   bias=T(golden["conv3d_06_b"]))
    assert_near(O.elu(y), golden["conv3d_06_y"], 1e-4)


def test_conv3d_07_multiple(golden):                    # :572-623
    x, w = T(golden["conv3d_07_x"]), T(golden["conv3d_07_w"])
    y1 = O.transform(O.conv3d_tf(x, w, None, (1, 1, 1), (1, 1, 1), (1, 1, 1)))
    y2 = O.conv3d_tf(O.pad_d(y1, 1), w, None, (2, 2, 2), (0, 1, 1), (0, 1, 1))
    assert_near(O.transform(y2), golden["conv3d_07_y"], 1e-4)


# ---- Conv3DTranspose (tests_main.cpp:629-878); y NDKHW (01,02) / NKDHW (03+), x NDCHW ------
def test_conv3d_tran_01_basic(golden):                  # :653-683
    y, w, x = (golden["conv3d_tran_01_" + k] for k in "ywx")
    out = O.conv3d_transpose_tf(T(y), T(w), None, x.shape[1:], (1, 1, 1), (0, 0, 0), (0, 0, 0))
    assert_float_eq(O.transform(out), x)                # D == C == ... transform is a no-op shape-wise


def test_conv3d_tran_02_hw(golden):                     # :685-715
    y, w, x = (golden["conv3d_tran_02_" + k] for k in "ywx")
    out = O.conv3d_transpose_tf(T(y), T(w), None, x.shape[1:], (1, 2, 2), (0, 1, 1), (0, 1, 1))
    assert_near(O.transform(out).reshape(x.shape), x, 1e-4)


def tran_sliced(y, w, b, x_shape):
    od = (x_shape[1] + 1,) + tuple(x_shape[2:])         # "manually pad output by 1 in D" (:730-732)
    out = O.conv3d_transpose_tf(T(y), T(w), b, od, (2, 2, 2), (0, 1, 1), (0, 1, 1))
    return O.slice_d(out, 0, x_shape[1])


def test_conv3d_tran_03_asym(golden):                   # :717-761
    y, w, x = (golden["conv3d_tran_03_" + k] for k in "ywx")
    assert_near(tran_sliced(y, w, None, x.shape), x, 1e-4)


def test_conv3d_tran_04_bias_elu(golden):               # :763-817
    y, w, x = (golden["conv3d_tran_04_" + k] for k in "ywx")
    assert_near(O.elu(tran_sliced(y, w, T(golden["conv3d_tran_04_b"]), x.shape)), x, 1e-4)


def test_conv3d_tran_05_multiple(golden):               # :819-878
    g = golden
    x1 = tran_sliced(g["conv3d_tran_05_y"], g["conv3d_tran_05_w1"], None, (1, 8, 8, 9, 9))
    x1 = O.transform(x1)                                # DCHW -> KDHW for the second deconv
    x2 = tran_sliced(x1.numpy(), g["conv3d_tran_05_w2"], None, g["conv3d_tran_05_x"].shape)
    assert_near(x2, g["conv3d_tran_05_x"], 1e-4)


# ---- Cost volumes (tests_main.cpp:884-1026) ----------------------------------------------
@pytest.mark.parametrize("idx", ["01", "02"])
def test_cost_volume(golden, idx):                      # :884-934
    cv = golden["cost_vol_%s_cv" % idx]
    out = O.cost_volume(T(golden["cost_vol_%s_l" % idx]), T(golden["cost_vol_%s_r" % idx]), cv.shape[1])
    assert_float_eq(out, cv)


def test_corr_cost_volume(golden):                      # :961-986
    cv = golden["corr_cost_vol_01_cv"]                   # (1, D, 1, H, W)
    out = O.corr_cost_volume(T(golden["corr_cost_vol_01_l"]), T(golden["corr_cost_vol_01_r"]), cv.shape[1])
    assert_near(out[:, :, None], cv, 1e-6)


# ---- Softargmax (tests_main.cpp:1032-1099) -----------------------------------------------
def test_softargmin_basic(golden):                      # :1032-1054
    assert_near(O.softargmax(T(golden["softargmax_01_x"]), True), golden["softargmax_01_y"], 2e-6)


def test_softargmin_batch2(golden):                     # :1056-1077
    assert_near(O.softargmax(T(golden["softargmax_02_x"]), True), golden["softargmax_02_y"], 1e-5)


def test_softargmax_basic(golden):                      # :1079-1099
    assert_near(O.softargmax(T(golden["softargmax_03_x"]), False), golden["softargmax_03_y"], 2e-6)


def test_corr_cpu_c_restatement(golden):
    """oracle/corr_cpu.c -- the plain C loop of lib/kernels.cu:168-200 that tools/bench_ops.py times as the op-level CPU baseline --
    against the reference's golden cost volume (corr_cost_vol_01, tests_main.cpp EXPECT_NEAR 1e-5... here 1e-6) and the torch oracle"""
    import ctypes
    from redtail_amd import build
    lib = ctypes.CDLL(build.build_oracle_c())
    l, r, want = golden["corr_cost_vol_01_l"], golden["corr_cost_vol_01_r"], golden["corr_cost_vol_01_cv"]
    l, r = np.ascontiguousarray(l, np.float32), np.ascontiguousarray(r, np.float32)
    c, h, w = l.shape[-3:]
    n = int(np.prod(l.shape[:-3]))
    d = want.shape[1]                                       # (1, D, 1, H, W)
    out = np.empty((n, d, h, w), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.corr_cost_volume_cpu(l.ctypes.data_as(fp), r.ctypes.data_as(fp), n, c, h, w, d, out.ctypes.data_as(fp))
    assert np.abs(out[:, :, None] - want).max() <= 1e-6
    ref = O.corr_cost_volume(T(l).reshape(n, c, h, w), T(r).reshape(n, c, h, w), d).numpy()
    assert np.abs(out - ref).max() <= 1e-6


# ---- second witness for the TensorRT-native 2-D layers (the reference ships no fixture for them) -----------------------
# The oracle's conv2d / deconv2d are torch.nn.functional calls.  These are plain fp64 loops written from the TensorFlow definitions the
# converter assumes (scripts/tensorrt_model_builder.py:140-147 _compute_tf_padding, :149-228 Conv2D NHWC x RSCK -> addConvolution KCRS,
# :230-288 Conv2DBackpropInput -> addDeconvolution with the SAME padding of its *output*), sharing no code with torch.
def _tf_padding(in_dim, kern, stride):                   # tensorrt_model_builder.py:140-147, restated
    along = max(kern - stride, 0) if in_dim % stride == 0 else max(kern - in_dim % stride, 0)
    return along // 2, along - along // 2


def _tf_conv2d_same(x_nhwc, f_rsck, stride):
    """tf.nn.conv2d(padding='SAME'): out[b,i,j,k] = sum_{di,dj,q} x[b, s*i+di-pt, s*j+dj-pl, q] * f[di,dj,q,k], out = ceil(in / s)"""
    n, h, w, c = x_nhwc.shape
    kh, kw, _, k = f_rsck.shape
    (pt, _), (pl, _) = _tf_padding(h, kh, stride), _tf_padding(w, kw, stride)
    ho, wo = -(-h // stride), -(-w // stride)
    out = np.zeros((n, ho, wo, k), np.float64)
    for i in range(ho):
        for j in range(wo):
            for di in range(kh):
                for dj in range(kw):
                    y, xx = stride * i + di - pt, stride * j + dj - pl
                    if 0 <= y < h and 0 <= xx < w:
                        out[:, i, j, :] += x_nhwc[:, y, xx, :].astype(np.float64) @ f_rsck[di, dj].astype(np.float64)
    return out


def _tf_conv2d_transpose_same(y_nhwc, f_rsck, out_hw, stride):
    """tf.nn.conv2d_transpose = Conv2DBackpropInput: the gradient of conv2d(SAME) with respect to its input of size out_hw; the filter is
    (kh, kw, C of the OUTPUT of the transposed conv, K of its input):  dx[b, s*i+di-pt, s*j+dj-pl, c] += y[b,i,j,k] * f[di,dj,c,k]"""
    n, hy, wy, k = y_nhwc.shape
    kh, kw, c, _ = f_rsck.shape
    H, W = out_hw
    (pt, _), (pl, _) = _tf_padding(H, kh, stride), _tf_padding(W, kw, stride)
    assert hy == -(-H // stride) and wy == -(-W // stride)
    out = np.zeros((n, H, W, c), np.float64)
    for i in range(hy):
        for j in range(wy):
            for di in range(kh):
                for dj in range(kw):
                    yy, xx = stride * i + di - pt, stride * j + dj - pl
                    if 0 <= yy < H and 0 <= xx < W:
                        out[:, yy, xx, :] += y_nhwc[:, i, j, :].astype(np.float64) @ f_rsck[di, dj].astype(np.float64).T
    return out


def _rsck_to_kcrs(f):                                    # scripts/data_converters.py:21-30
    return np.ascontiguousarray(np.transpose(f, (3, 2, 0, 1)))


@pytest.mark.parametrize("h,w,cin,cout,k,stride", [
    (11, 17, 3, 8, 5, 2),      # the first layer's form (resnet18_2D_513x257_net.cpp:48-53): 5x5 stride 2 on the image, odd sizes
    (9, 13, 6, 5, 3, 2),       # conv2D_*ds: 3x3 stride 2
    (7, 9, 4, 4, 3, 1),        # tower / bottleneck layers: 3x3 stride 1
])
def test_conv2d_oracle_against_plain_tf_definition(h, w, cin, cout, k, stride):
    rng = np.random.default_rng(h * 100 + w)
    x = rng.standard_normal((2, h, w, cin)).astype(np.float32)
    f = rng.standard_normal((k, k, cin, cout)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    want = _tf_conv2d_same(x, f, stride) + b.astype(np.float64)
    pad_h, pad_w = _tf_padding(h, k, stride), _tf_padding(w, k, stride)
    assert pad_h[0] == pad_h[1] and pad_w[0] == pad_w[1]            # what the converter asserts before it uses setPadding (:196-199)
    got = O.conv2d(T(np.transpose(x, (0, 3, 1, 2))).double(), T(_rsck_to_kcrs(f)).double(), T(b).double(), stride, pad_h[0]).numpy()
    assert got.shape == (2, cout, -(-h // stride), -(-w // stride))
    assert np.abs(np.transpose(got, (0, 2, 3, 1)) - want).max() <= 1e-12


@pytest.mark.parametrize("hy,wy,H,W,k_in,c_out,k,stride", [
    (5, 7, 9, 13, 6, 4, 3, 2),     # deconv2D_*: 3x3 stride 2, SAME padding 1 of the (odd) output
    (4, 6, 7, 11, 8, 1, 3, 2),     # the last layer: one output channel
    (5, 6, 5, 6, 3, 5, 3, 1),      # stride 1
])
def test_deconv2d_oracle_against_plain_tf_definition(hy, wy, H, W, k_in, c_out, k, stride):
    rng = np.random.default_rng(H * 100 + W)
    y = rng.standard_normal((2, hy, wy, k_in)).astype(np.float32)
    f = rng.standard_normal((k, k, c_out, k_in)).astype(np.float32)       # TF: (kh, kw, output channels, input channels)
    b = rng.standard_normal(c_out).astype(np.float32)
    want = _tf_conv2d_transpose_same(y, f, (H, W), stride) + b.astype(np.float64)
    pad_h, pad_w = _tf_padding(H, k, stride), _tf_padding(W, k, stride)
    assert pad_h[0] == pad_h[1] and pad_w[0] == pad_w[1]            # :268-270
    # rsck_to_kcrs of that filter is (k_in, c_out, R, S): addDeconvolution's (Cin, Cout, R, S)
    got = O.deconv2d(T(np.transpose(y, (0, 3, 1, 2))).double(), T(_rsck_to_kcrs(f)).double(), T(b).double(), stride, pad_h[0]).numpy()
    assert got.shape == (2, c_out, H, W)
    assert np.abs(np.transpose(got, (0, 2, 3, 1)) - want).max() <= 1e-12


# ---- a second witness for the INTER_AREA restatement (readImgFile, sample_app/main.cpp:83-98; SURVEY 8f-3: no OpenCV here) -------------
def _area_integral(x, dsize):
    """shrinking by area averaging, stated as what it IS: destination pixel d is the mean of the piecewise-constant source signal over
    [d s, (d + 1) s), s = ssize / dsize, clipped to the image -- exact overlap integrals in fp64, no table, no thresholds"""
    ssize = x.shape[0]
    s = ssize / dsize
    edges = np.arange(ssize + 1, dtype=np.float64)
    out = np.empty((dsize,) + x.shape[1:], np.float64)
    for d in range(dsize):
        a, b = d * s, min((d + 1) * s, float(ssize))
        w = np.clip(np.minimum(edges[1:], b) - np.maximum(edges[:-1], a), 0.0, None)      # overlap of [a, b) with source pixel [i, i + 1)
        out[d] = np.tensordot(w, x, axes=(0, 0)) / (b - a)
    return out


@pytest.mark.parametrize("src,dst", [((375, 1242), (321, 1025)), ((50, 97), (9, 17)), ((40, 66), (20, 33)), ((370, 1226), (369, 1225))])
def test_inter_area_restatement_against_overlap_integrals(src, dst):
    """oracle.preprocess_bgr8 restates OpenCV's computeResizeAreaTab (taps with weights below 1e-3 are dropped there without
    renormalising); the area average it implements is pinned here by the overlap integrals themselves -- an independent statement of the
    same filter; the reference's own data pins it in test_preprocessing_reproduces_the_reference_sample_input below."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=src + (3,), dtype=np.uint8)
    got = O.preprocess_bgr8(img, *dst).astype(np.float64)                                    # (3, dh, dw), RGB, / 255
    x = img.astype(np.float64)
    ref = _area_integral(_area_integral(x, dst[0]).transpose(1, 0, 2), dst[1]).transpose(1, 0, 2)      # rows, then columns
    ref = ref[:, :, ::-1].transpose(2, 0, 1) / 255.0
    # a dropped tap carries < 1e-3 of a source pixel's weight out of a cell of >= 1 pixel, per axis
    # (measured: 7.9e-4 max / 8e-7 mean at 1242x375 -> 1025x321, 3e-8 where no tap falls under the threshold)
    assert np.abs(got - ref).max() <= 1.5e-3, np.abs(got - ref).max()
    assert np.abs(got - ref).mean() <= 1e-5


# ---- the pre-processing pinned by the reference's own data (VERDICT r04 missing #4) ---------------------------------------------------
def _sample_fixture():
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "redtail_sample_image.npz"))
    return f["img_left_png_rgb"], f["img_left_bin_rows"], tuple(int(v) for v in f["band"])


def test_preprocessing_reproduces_the_reference_sample_input():
    """sample_app/data/img_left.bin is the network input the reference ships for its sample pair: 3 x 321 x 1025 float32 CHW RGB in
    [0, 1] -- readImgFile (sample_app/main.cpp:83-98: area resize of the 1242 x 375 PNG, BGR -> RGB, / 255) applied to img_left.png.
    (1) The oracle's restatement of that function reproduces it to <= 1e-3 max / <= 2e-5 mean (measured 8.3e-4 / 1.0e-5).
    (2) The whole of that distance is explained: the file was produced by TensorFlow's resize_area -- oracle.resize_area_tf matches it to
        3e-7 -- which keeps every tap, while OpenCV's INTER_AREA table drops taps below 1e-3 source pixels: away from the two destination
        columns where that happens (222 and 802) the two agree to 1e-4 (float32 source coordinates in TF: 7e-5 at the right edge).
    Fixture: the decoded PNG and destination rows 160..287 of the .bin (tests/golden/make_golden.py)."""
    png_rgb, ref_rows, (r0, r1) = _sample_fixture()
    assert png_rgb.shape == (375, 1242, 3) and ref_rows.shape == (3, r1 - r0, 1025)
    got = O.preprocess_bgr8(np.ascontiguousarray(png_rgb[:, :, ::-1]), 321, 1025)[:, r0:r1]             # the function takes BGR, as cv::imread gives it
    err = np.abs(got - ref_rows)
    assert err.max() <= 1e-3 and err.mean() <= 2e-5, (err.max(), err.mean())
    swapped = O.preprocess_bgr8(np.ascontiguousarray(png_rgb), 321, 1025)[:, r0:r1]                    # the channel order matters: 0.9 the other way round
    assert np.abs(swapped - ref_rows).max() > 0.5
    cols = O.area_taps_below_threshold(1242, 1025)
    assert cols == [222, 802] and O.area_taps_below_threshold(375, 321) == []
    keep = np.ones(1025, bool)
    keep[cols] = False
    assert err[:, :, keep].max() <= 1e-4, err[:, :, keep].max()
    assert err[:, :, cols].max() > 5e-4                                                                # ... and that IS where the distance comes from
    tf = (O.resize_area_tf(png_rgb, 321, 1025)[r0:r1].transpose(2, 0, 1) / np.float32(255.0)).astype(np.float32)
    assert np.abs(tf - ref_rows).max() <= 4e-7, np.abs(tf - ref_rows).max()
    assert ref_rows.max() > 1.0 and tf.max() == ref_rows.max()                                         # 1.0000731: float32 cell weights that do not sum to 1


def test_preprocessing_reproduces_both_reference_sample_inputs_in_full():
    """the same on the complete files, both images, where the reference's data is staged (weights/_ref/sample: wherever build() ran with
    /root/reference present; the snapshot that travels to the GPU box carries it)"""
    from PIL import Image
    from redtail_amd import model_files
    if model_files.sample_bin("left") is None:
        pytest.skip("the reference's sample data is not staged here")
    for side in ("left", "right"):
        png_rgb = np.asarray(Image.open(model_files.sample_image(side)))
        ref = np.fromfile(model_files.sample_bin(side), dtype="<f4").reshape(3, 321, 1025)
        err = np.abs(O.preprocess_bgr8(np.ascontiguousarray(png_rgb[:, :, ::-1]), 321, 1025) - ref)
        assert err.max() <= 1e-3 and err.mean() <= 2e-5, (side, err.max(), err.mean())
        tf = (O.resize_area_tf(png_rgb, 321, 1025).transpose(2, 0, 1) / np.float32(255.0)).astype(np.float32)
        assert np.abs(tf - ref).max() <= 4e-7, (side, np.abs(tf - ref).max())
