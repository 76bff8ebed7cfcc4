"""C-ABI parity tests: every op of include/rt_stereo.h against the reference's golden tensors and the
oracle.  Each test runs twice: on the SIMT emulator (CPU tier, same kernel sources) and, with -m gpu,
on the MI355X through librt_stereo_hip.so.  Tolerances are the reference's own
(/root/reference/stereoDNN/tests/tests_main.cpp, lines cited per test)."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def near(actual, expected, tol):
    a, e = np.asarray(actual, np.float32), np.asarray(expected, np.float32)
    assert a.shape == e.shape, (a.shape, e.shape)
    assert not np.isnan(a).any(), "output has unwritten / NaN elements"
    err = np.abs(a - e).max() if a.size else 0.0
    assert err <= tol, "max |diff| = %g > %g" % (err, tol)


def float_eq(actual, expected, ulps=4):
    a, e = np.asarray(actual, np.float32).ravel(), np.asarray(expected, np.float32).ravel()
    assert a.shape == e.shape
    assert not np.isnan(a).any()
    tol = ulps * np.spacing(np.maximum(np.abs(a), np.abs(e)).astype(np.float32))
    assert (np.abs(a - e) <= tol).all(), "max |diff| %g" % np.abs(a - e).max()


RNG = np.random.default_rng(1)


def rnd(*shape):
    return RNG.standard_normal(shape).astype(np.float32)


# ---- ELU / element-wise (tests_main.cpp:280-342) --------------------------------------------------
@pytest.mark.parametrize("idx", ["01", "02"])
def test_elu_golden(backend, golden, idx):
    x = backend.dev(golden["elu_i_" + idx])
    y = backend.empty(x.shape)
    backend.klib.elu(x, y, int(np.prod(x.shape)))
    float_eq(backend.host(y), golden["elu_o_" + idx])


@pytest.mark.parametrize("n,off", [(1, 0), (1027, 0), (4096, 1), (70001, 3)])
def test_elementwise_sizes(backend, n, off):
    """ragged tails and 4-byte (not 16-byte) aligned pointers"""
    a, b = 3 * rnd(n + off), 3 * rnd(n + off)
    da, db, dy = backend.dev(a), backend.dev(b), backend.empty((n + off,))
    backend.klib.elu(da[off:], dy[off:], n)
    near(backend.host(dy)[off:], O.elu(T(a[off:])).numpy(), 1e-6)
    backend.klib.add_act(da[off:], db[off:], dy[off:], n, capi.RT_ACT_ELU)
    near(backend.host(dy)[off:], O.elu(T(a[off:] + b[off:])).numpy(), 1e-6)
    backend.klib.add_act(da[off:], db[off:], dy[off:], n, capi.RT_ACT_NONE)
    near(backend.host(dy)[off:], a[off:] + b[off:], 0)
    backend.klib.activation(da[off:], dy[off:], n, capi.RT_ACT_SIGMOID)
    near(backend.host(dy)[off:], torch.sigmoid(T(a[off:])).numpy(), 1e-6)


def test_elementwise_empty(backend):
    backend.klib.elu(None, None, 0)            # n == 0 is a no-op, like an empty TRT tensor


# ---- correlation cost volume (tests_main.cpp:961-986) --------------------------------------------
def run_corr(backend, l, r, D):
    n, c, h, w = l.shape
    cv = backend.empty((n, D, h, w))
    backend.klib.corr_cost_volume(backend.dev(l), backend.dev(r), cv, n, c, h, w, D)
    return backend.host(cv)


def test_corr_golden(backend, golden):
    cv = golden["corr_cost_vol_01_cv"]
    out = run_corr(backend, golden["corr_cost_vol_01_l"], golden["corr_cost_vol_01_r"], cv.shape[1])
    near(out[:, :, None], cv, 1e-6)


@pytest.mark.parametrize("shape,D", [((1, 5, 7, 37), 6), ((2, 32, 9, 140), 48), ((1, 3, 5, 130), 70),
                                     ((1, 8, 4, 129), 16)])
def test_corr_random(backend, shape, D):
    l, r = rnd(*shape), rnd(*shape)
    near(run_corr(backend, l, r, D), O.corr_cost_volume(T(l), T(r), D).numpy(), 2e-5)


@pytest.mark.parametrize("shape,D", [((2, 32, 5, 140), 48), ((1, 16, 3, 64), 64), ((1, 24, 4, 97), 33), ((1, 32, 2, 161), 20),
                                     ((3, 20, 2, 70), 34)])
def test_corr_planar_on_the_matrix_cores(backend, shape, D, monkeypatch):
    """maps of a network's size take corr_mfma_planar_kernel (Gram band, 3-term fp16 split, transposed through the wave's LDS):
    bounded against the fp64 volume like the split convolutions, every element written, and the fp32 fmaf kernel still there"""
    l, r = rnd(*shape), rnd(*shape)
    r[:, :, :, :3] *= np.float32(40.0)                     # large products at the left edge, where x - d < 0 must give exact zeros
    got = run_corr(backend, l, r, D)
    ref = O.corr_cost_volume(T(l).double(), T(r).double(), D).numpy()
    mag = O.corr_cost_volume(T(np.abs(l)).double(), T(np.abs(r)).double(), D).numpy()
    assert np.all(np.abs(got - ref) <= 2.0 ** -21 * mag + 1e-30), float(np.max(np.abs(got - ref) / np.maximum(mag, 1e-30)))
    for d in range(1, D):
        assert not got[:, d, :, :min(d, shape[3])].any()
    # RT_CONV_EXACT_FP32 (what CostVolumePlugin::enqueue passes in an exact-fp32 engine) keeps the fp32 fmaf kernel
    n, c, h, w = shape
    cv = backend.empty((n, D, h, w))
    backend.klib.corr_cost_volume(backend.dev(l), backend.dev(r), cv, n, c, h, w, D, flags=capi.RT_CONV_EXACT_FP32)
    exact = backend.host(cv)
    near(exact, ref, 2e-5 * float(np.abs(ref).max()))
    assert (exact != got).any()                              # (two kernels, two summation orders)
    monkeypatch.setenv("RT_NO_CORR_MFMA_PLANAR", "1")
    assert np.array_equal(run_corr(backend, l, r, D), exact)


@pytest.mark.parametrize("shape,D,is_min", [((2, 32, 5, 140), 48, False), ((1, 16, 3, 64), 64, True), ((1, 24, 4, 97), 33, False),
                                            ((3, 20, 2, 70), 34, True)])
def test_corr_softargmax_planar_on_the_matrix_cores(backend, shape, D, is_min, monkeypatch):
    """rt_corr_softargmax on a client's planar maps of a network's size: the Gram band on the matrix cores (3-term fp16 split), the same map
    as the fp32 chain to the accuracy of the split; rt_corr_softargmax_pitched -- what exact-fp32 engines call -- stays on the fp32 chain"""
    l, r = rnd(*shape) * np.float32(0.5), rnd(*shape) * np.float32(0.5)
    n, c, h, w = shape
    ref = O.softargmax(O.corr_cost_volume(T(l).double(), T(r).double(), D), is_min).numpy()
    out = backend.empty((n, 1, h, w))
    backend.klib.corr_softargmax(backend.dev(l), backend.dev(r), out, n, c, h, w, D, is_min)
    got = backend.host(out)
    near(got, ref, 2e-4)
    fp32 = backend.empty((n, 1, h, w))
    backend.klib.corr_softargmax_pitched(backend.dev(l), backend.dev(r), fp32, n, c, h, w, D, is_min, 0, 0)
    fp32 = backend.host(fp32)
    near(fp32, ref, 2e-4)
    assert (fp32 != got).any()                               # (two kernels)
    monkeypatch.setenv("RT_NO_CORR_MFMA_PLANAR", "1")
    again = backend.empty((n, 1, h, w))
    backend.klib.corr_softargmax(backend.dev(l), backend.dev(r), again, n, c, h, w, D, is_min)
    assert np.array_equal(backend.host(again), fp32)


@pytest.mark.parametrize("shape,D,is_min", [((1, 5, 7, 37), 6, False), ((2, 32, 9, 140), 48, False),
                                            ((1, 8, 5, 129), 13, True), ((1, 4, 6, 6), 2, False)])
def test_corr_softargmax_fused(backend, shape, D, is_min):
    l, r = rnd(*shape), rnd(*shape)
    n, c, h, w = shape
    out = backend.empty((n, 1, h, w))
    backend.klib.corr_softargmax(backend.dev(l), backend.dev(r), out, n, c, h, w, D, is_min)
    ref = O.softargmax(O.corr_cost_volume(T(l), T(r), D), is_min).numpy()
    near(backend.host(out), ref, 2e-4)


def run_corr_fp16(backend, l, r, D):
    """fp16 NC2HW2 tensors travel as 4-byte slots (numpy float32 views of the float16 pairs)"""
    n, c, h, w = l.shape
    lh, rh = O.to_nc2hw2(T(l)), O.to_nc2hw2(T(r))
    as_slots = lambda t: np.ascontiguousarray(t.numpy()).view(np.float32).reshape(t.shape[:-1])
    cv = backend.empty((n, (D + 1) // 2, h, w))
    backend.klib.corr_cost_volume(backend.dev(as_slots(lh)), backend.dev(as_slots(rh)), cv, n, c, h, w, D,
                                  dtype=capi.RT_F16, fmt=capi.RT_NC2HW2)
    out = torch.from_numpy(np.ascontiguousarray(backend.host(cv)).view(np.float16).reshape(n, (D + 1) // 2, h, w, 2))
    return O.from_nc2hw2(out, D).numpy(), O.from_nc2hw2(O.corr_cost_volume_fp16(lh, rh, c, D), D).numpy()


def test_corr_golden_fp16_nc2hw2(backend, golden):                 # tests_main.cpp:988-1026, tolerance 0.01
    cv = golden["corr_cost_vol_01_cv"]
    got, ref16 = run_corr_fp16(backend, golden["corr_cost_vol_01_l"], golden["corr_cost_vol_01_r"], cv.shape[1])
    near(got, cv.reshape(got.shape), 1e-2)
    near(got, ref16, 2e-3)                                         # one half ulp at |x| < 4 is 2e-3


@pytest.mark.parametrize("shape,D", [((2, 5, 7, 45), 7), ((1, 32, 9, 133), 48)])
def test_corr_random_fp16_nc2hw2(backend, shape, D):
    """odd channel and disparity counts (zero padded half of the last slot), several tiles"""
    l, r = rnd(*shape) * np.float32(0.5), rnd(*shape) * np.float32(0.5)
    got, ref16 = run_corr_fp16(backend, l, r, D)
    scale = max(1.0, float(np.abs(ref16).max()))
    near(got, ref16, 1e-3 * scale)
    near(got, O.corr_cost_volume(T(l), T(r), D).numpy(), 1e-2 * scale)


@pytest.mark.parametrize("is_min", [False, True])
def test_softargmax_fp16(backend, is_min):
    """kHALF volumes in NCHW (softargmax_plugin.cpp:51-54): fp16 in and out, fp32 arithmetic in between"""
    n, d, h, w = 2, 12, 9, 33
    vol = (rnd(n, d, h, w) * np.float32(3)).astype(np.float16)
    ref = O.softargmax(T(vol.astype(np.float32)), is_min).numpy()
    if backend.name == "gpu":
        src, dst = torch.from_numpy(vol).cuda(), torch.full((n, 1, h, w), float("nan"), dtype=torch.float16, device="cuda")
        backend.klib.softargmax(src, dst, n, d, h, w, is_min, dtype=capi.RT_F16)
        torch.cuda.synchronize()
        got = dst.cpu().numpy().astype(np.float32)
    else:
        src, dst = np.ascontiguousarray(vol), np.full((n, 1, h, w), np.nan, np.float16)
        backend.klib.softargmax(src, dst, n, d, h, w, is_min, dtype=capi.RT_F16)
        got = dst.astype(np.float32)
    near(got, ref, 1e-2)                                     # the reference's fp16 tolerance (tests_main.cpp:320)
    near(got, ref.astype(np.float16).astype(np.float32), 8e-3)   # at most one half ulp below 16


def test_corr_softargmax_into_concat_buffer(backend):
    """out_batch_stride places the result in channel 32 of a 33-channel buffer (resnet18_2D net :601-615)"""
    n, c, h, w, D = 2, 8, 6, 40, 12
    l, r = rnd(n, c, h, w), rnd(n, c, h, w)
    buf = backend.dev(np.zeros((n, 5, h, w), np.float32))
    backend.klib.corr_softargmax(backend.dev(l), backend.dev(r), buf[:, 4], n, c, h, w, D, False, out_bstride=5 * h * w)
    got = backend.host(buf)
    near(got[:, 4:5], O.softargmax(O.corr_cost_volume(T(l), T(r), D), False).numpy(), 2e-4)
    assert (got[:, :4] == 0).all()


# ---- default cost volume (tests_main.cpp:884-934) ------------------------------------------------
@pytest.mark.parametrize("idx", ["01", "02"])
def test_cost_volume_golden(backend, golden, idx):
    l, r, cv = (golden["cost_vol_%s_%s" % (idx, k)] for k in ("l", "r", "cv"))
    n, c, h, w = l.shape
    out = backend.empty(cv.shape)
    backend.klib.cost_volume(backend.dev(l), backend.dev(r), out, n, c, h, w, cv.shape[1])
    float_eq(backend.host(out), cv)


def test_cost_volume_random(backend):
    l, r = rnd(2, 3, 5, 300), rnd(2, 3, 5, 300)
    out = backend.empty((2, 7, 6, 5, 300))
    backend.klib.cost_volume(backend.dev(l), backend.dev(r), out, 2, 3, 5, 300, 7)
    near(backend.host(out), O.cost_volume(T(l), T(r), 7).numpy(), 0)


@pytest.mark.parametrize("n,c,h,w,D", [(2, 4, 5, 33, 7),      # plane 165 = 1 (mod 4): every channel another alignment, ragged first / last groups
                                       (1, 8, 3, 18, 24),     # more disparities than columns (x < d everywhere in the late slices), plane = 2 (mod 4)
                                       (2, 12, 2, 64, 5),     # aligned planes
                                       (1, 4, 1, 3, 2)])      # a plane smaller than one group of four
def test_cost_volume_16_byte_stores(backend, monkeypatch, n, c, h, w, D):
    """C a multiple of 4: cost_volume_f32x4_kernel (four plane elements per thread, stores aligned in the OUTPUT) -- against the oracle
    (lib/kernels.cu:50-97 semantics) and, bit for bit, against the element-per-thread kernel (RT_NO_CV_X4=1)."""
    l, r = rnd(n, c, h, w), rnd(n, c, h, w)
    ref = O.cost_volume(T(l), T(r), D).numpy()
    out = backend.empty(ref.shape)
    backend.klib.cost_volume(backend.dev(l), backend.dev(r), out, n, c, h, w, D)
    got = backend.host(out).copy()
    near(got, ref, 0)
    monkeypatch.setenv("RT_NO_CV_X4", "1")
    out = backend.empty(ref.shape)
    backend.klib.cost_volume(backend.dev(l), backend.dev(r), out, n, c, h, w, D)
    assert np.array_equal(backend.host(out), got)


# ---- soft-argmax (tests_main.cpp:1032-1099) ---------------------------------------------------------
@pytest.mark.parametrize("idx,is_min,tol", [("01", True, 2e-6), ("02", True, 1e-5), ("03", False, 2e-6)])
def test_softargmax_golden(backend, golden, idx, is_min, tol):
    x, y = golden["softargmax_%s_x" % idx], golden["softargmax_%s_y" % idx]
    n, d, _, h, w = x.shape
    out = backend.empty(y.shape)
    backend.klib.softargmax(backend.dev(x), out, n, d, h, w, is_min)
    near(backend.host(out), y, tol)


@pytest.mark.parametrize("D", [1, 7, 8, 9, 48, 136])
def test_softargmax_random(backend, D):
    x = 4 * rnd(2, D, 5, 67)
    out = backend.empty((2, 1, 5, 67))
    backend.klib.softargmax(backend.dev(x), out, 2, D, 5, 67, False)
    near(backend.host(out), O.softargmax(T(x), False).numpy(), 1e-5 * max(D, 8))


# ---- Transform / Padding / Slice / concat -------------------------------------------------------------
def test_permute_pad_slice_concat(backend):
    x = rnd(2, 3, 4, 5, 6)
    y = backend.empty((2, 4, 3, 5, 6))
    backend.klib.permute4d(backend.dev(x), y, 2, (3, 4, 5, 6), (1, 0, 2, 3))
    near(backend.host(y), O.transform(T(x)).numpy(), 0)
    y = backend.empty((2, 6, 3, 5, 4))
    backend.klib.permute4d(backend.dev(x), y, 2, (3, 4, 5, 6), (3, 0, 2, 1))
    near(backend.host(y), np.transpose(x, (0, 4, 1, 3, 2)), 0)
    # runs that stay in place (permute_runs_kernel): the innermost two dimensions / the innermost one; 16-byte units and ragged ones;
    # a run longer than one block's share (256 * 8 elements)
    for dims, order in [((3, 4, 5, 8), (1, 0, 2, 3)), ((3, 4, 5, 7), (1, 0, 2, 3)), ((3, 4, 5, 8), (2, 0, 1, 3)), ((4, 3, 5, 6), (2, 1, 0, 3)),
                        ((2, 3, 37, 61), (1, 0, 2, 3)), ((2, 3, 48, 64), (1, 0, 2, 3))]:
        xx = rnd(2, *dims)
        want = np.ascontiguousarray(np.transpose(xx, (0,) + tuple(1 + o for o in order)))
        yy = backend.empty(want.shape)
        backend.klib.permute4d(backend.dev(xx), yy, 2, dims, order)
        near(backend.host(yy), want, 0)
    y = backend.empty((2, 4, 4, 5, 6))
    backend.klib.pad_d(backend.dev(x), y, 2, 3, 4 * 5 * 6, 1)
    near(backend.host(y), O.pad_d(T(x), 1).numpy(), 0)
    y = backend.empty((2, 2, 4, 5, 6))
    backend.klib.slice_d(backend.dev(x), y, 2, 3, 4 * 5 * 6, 0, 2)
    near(backend.host(y), x[:, 0:2], 0)
    backend.klib.slice_d(backend.dev(x), y, 2, 3, 4 * 5 * 6, 1, 3)
    near(backend.host(y), x[:, 1:3], 0)
    buf = backend.dev(np.zeros((2, 5, 4, 5, 6), np.float32))
    backend.klib.concat_channels(backend.dev(x), buf, 2, 3, 5, 1, 4 * 5 * 6)
    got = backend.host(buf)
    near(got[:, 1:4], x, 0)
    assert (got[:, 0] == 0).all() and (got[:, 4] == 0).all()


def test_bad_arguments_fail_loudly(backend):
    x = backend.dev(rnd(8))
    with pytest.raises(capi.RtError):
        backend.klib.slice_d(x, x, 1, 3, 1, 2, 1)            # empty slice, slice_plugin.cpp:28-29 asserts
    with pytest.raises(capi.RtError):
        backend.klib.permute4d(x, x, 1, (1, 2, 2, 2), (0, 0, 1, 2))
    with pytest.raises(capi.RtError):
        backend.klib.corr_cost_volume(x, x, x, 1, 0, 1, 1, 1)
