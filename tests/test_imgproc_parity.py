"""GPU front-end / back-end of the sample application against the oracle restatement of
readImgFile / the 16-bit PNG encoding (sample_app/main.cpp:83-98, 324-330).  The area filter is pinned by the reference's own sample
input (sample_app/data/img_left.bin: test_preprocess_reproduces_the_reference_sample_input here, and in tests/test_oracle_golden.py for
the oracle); the integer-factor case is additionally checked against a plain box average."""
import os
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi


def run_pre(backend, img, dh, dw):
    n, sh, sw, _ = img.shape
    if backend.name == "gpu":
        src = torch.from_numpy(img).cuda()
        dst = torch.full((n, 3, dh, dw), float("nan"), device="cuda")
        backend.klib.preprocess_bgr8(src, sh, sw, dst, dh, dw, n)
        torch.cuda.synchronize()
        return dst.cpu().numpy()
    dst = np.full((n, 3, dh, dw), np.nan, np.float32)
    backend.klib.preprocess_bgr8(np.ascontiguousarray(img), sh, sw, dst, dh, dw, n)
    return dst


@pytest.mark.parametrize("src,dst", [((37, 59), (37, 59)), ((375, 1242), (321, 1025)), ((40, 66), (20, 33)), ((50, 97), (9, 17))])
def test_preprocess(backend, src, dst):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(2,) + src + (3,), dtype=np.uint8)
    out = run_pre(backend, img, *dst)
    ref = np.stack([O.preprocess_bgr8(img[i], *dst) for i in range(2)])
    assert np.abs(out - ref).max() <= 2e-6
    if src == (40, 66):      # integer factor 2: INTER_AREA is the 2x2 box average
        box = img.astype(np.float64).reshape(2, 20, 2, 33, 2, 3).mean((2, 4))[..., ::-1].transpose(0, 3, 1, 2) / 255
        assert np.abs(out - box).max() <= 2e-6


def test_preprocess_reproduces_the_reference_sample_input(backend):
    """rt_preprocess_bgr8 on the reference's sample_app/data/img_left.png against the network input the reference ships for it
    (img_left.bin, rows 160..287 in the fixture): <= 1e-3 max, <= 2e-5 mean -- the bounds tests/test_oracle_golden.py holds the oracle to"""
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "redtail_sample_image.npz"))
    png_rgb, ref_rows, (r0, r1) = f["img_left_png_rgb"], f["img_left_bin_rows"], tuple(int(v) for v in f["band"])
    out = run_pre(backend, np.ascontiguousarray(png_rgb[None, :, :, ::-1]), 321, 1025)[0, :, r0:r1]
    err = np.abs(out - ref_rows)
    assert err.max() <= 1e-3 and err.mean() <= 2e-5, (err.max(), err.mean())


def test_preprocess_rejects_upscaling(backend):
    img = np.zeros((1, 8, 8, 3), np.uint8)
    with pytest.raises(capi.RtError):
        backend.klib.preprocess_bgr8(img if backend.name == "emu" else torch.from_numpy(img).cuda(), 8, 8,
                                     np.zeros((1, 3, 9, 9), np.float32) if backend.name == "emu" else torch.zeros(1, 3, 9, 9, device="cuda"),
                                     9, 9, 1)


def test_disparity_to_u16(backend):
    disp = np.concatenate([np.float32([0.0, 0.5 / 256, 1.5 / 256, 2.5 / 256, -3.0, 300.0, 255.99]),
                           np.random.default_rng(5).uniform(0, 200, 1000).astype(np.float32)])
    n = disp.size
    if backend.name == "gpu":
        out = torch.zeros(n, dtype=torch.int16, device="cuda")
        backend.klib.disparity_to_u16(torch.from_numpy(disp).cuda(), out, n, 256.0)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint16)
    else:
        got = np.zeros(n, np.uint16)
        backend.klib.disparity_to_u16(disp, got, n, 256.0)
    ref = O.disparity_to_u16(disp, 256.0)
    assert np.array_equal(got, ref)
    assert list(got[:7]) == [0, 0, 2, 2, 0, 65535, 65533]      # ties to even, saturation at both ends
