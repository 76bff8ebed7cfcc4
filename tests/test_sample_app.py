"""The REFERENCE's sample application (stereoDNN/sample_app/main.cpp + its four generated network builders), compiled
untouched against our headers and linked to libnvstereo_inference.so (redtail_amd/build.py:build_sample_app ->
oracle/_ref/nvstereo_sample_app; OpenCV is replaced by the test-only subset in tests/shim/), run as a process on the
reference's own sample pair with the reference's trained weights, its .bin / .png outputs compared with the oracle.
"sample_app links unchanged" (BASELINE north_star) as an executable check."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi, kitti, model_files

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "oracle", "_ref", "nvstereo_sample_app")


def sample_pair(w, h):
    """the app's readImgFile pipeline (main.cpp:83-98) restated by the oracle on the decoded PNGs"""
    from PIL import Image
    def load(side):
        rgb = np.array(Image.open(model_files.sample_image(side)).convert("RGB"))
        return O.preprocess_bgr8(rgb[:, :, ::-1].copy(), h, w)
    return load("left"), load("right")


def run_app(tmp_path, model, w, h, weights, dtype=None):
    out = str(tmp_path / "disp.bin")
    cmd = [APP, model, str(w), str(h), weights, model_files.sample_image("left"), model_files.sample_image("right"), out]
    if dtype:
        cmd.append(dtype)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "Done" in res.stdout
    return np.fromfile(out, dtype=np.float32).reshape(h, w), out, res


@pytest.mark.gpu
def test_sample_app_resnet18_2d_fp32_and_plan(tmp_path):
    if not os.path.exists(APP):
        pytest.skip("oracle/_ref/nvstereo_sample_app not built (no /root/reference at build time)")
    w, h = 513, 257
    wfile = str(tmp_path / "trt_weights.bin")                       # the app writes <weights>.plan next to the file
    shutil.copyfile(model_files.weight_file("resnet18_2D"), wfile)
    out, path, res = run_app(tmp_path, "resnet18_2D", w, h, wfile)
    assert "Saving TensorRT plan" in res.stdout and os.path.exists(wfile + ".plan")
    l, r = sample_pair(w, h)
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], capi.read_weights(wfile)).numpy()[0, 0]
    err = np.abs(out - ref).max()
    print("sample_app resnet18_2D 513x257 fp32: max |disp - oracle| = %.3g" % err)
    assert err <= 1e-3, err
    # 16-bit PNG as KITTI stores disparities (main.cpp:324-330): disp * 256 * width
    png = kitti.read_disparity_png(path + ".png") if hasattr(kitti, "read_disparity_png") else None
    if png is not None:
        assert np.abs(png - out * w).max() <= 1.0 / 256 + 1e-6
    # second run: the plan is there -> IRuntime::deserializeCudaEngine + StereoDnnPluginFactory (main.cpp:198-220)
    out2, _, res2 = run_app(tmp_path, "resnet18_2D", w, h, wfile)
    assert "Loading TensorRT plan" in res2.stdout
    assert np.array_equal(out, out2)


@pytest.mark.gpu
def test_sample_app_resnet18_2d_fp16(tmp_path):
    """`fp16` on the command line: readWeights takes the fp16 file, setHalf2Mode(true) (main.cpp:126, 256-262)"""
    if not os.path.exists(APP):
        pytest.skip("oracle/_ref/nvstereo_sample_app not built")
    w, h = 513, 257
    wfile = str(tmp_path / "trt_weights_fp16.bin")
    shutil.copyfile(model_files.weight_file("resnet18_2D", True), wfile)
    out, _, _ = run_app(tmp_path, "resnet18_2D", w, h, wfile, "fp16")
    l, r = sample_pair(w, h)
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], capi.read_weights(wfile, fp16=True)).numpy()[0, 0]
    err = np.abs(out - ref).max()
    print("sample_app resnet18_2D 513x257 fp16: max |disp - oracle(fp16 weights)| = %.3g" % err)
    assert err <= 1e-2, err            # the reference's own fp16 tolerance (tests_main.cpp:320, 1025)


@pytest.mark.gpu
def test_sample_app_nvtiny(tmp_path):
    """`nvsmall 513 161` selects createNVTiny513x161Network (main.cpp:233-236): the 3-D plugin path"""
    if not os.path.exists(APP):
        pytest.skip("oracle/_ref/nvstereo_sample_app not built")
    w, h = 513, 161
    out, _, _ = run_app(tmp_path, "nvsmall", w, h, model_files.weight_file("nvtiny"))
    l, r = sample_pair(w, h)
    with torch.no_grad():
        ref = O.stereo3d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], capi.read_weights(model_files.weight_file("nvtiny")),
                         O.NVTINY_3D, 24).numpy()[0, 0]
    err = np.abs(out - ref).max()
    print("sample_app NVTiny 513x161: max |disp - oracle| = %.3g px" % err)
    assert err <= 1e-3, err


def test_sample_app_is_built_from_untouched_reference_sources():
    """CPU tier: the build recipe compiles main.cpp where it lies (nothing is copied into the repository) and the binary
    resolves its libraries"""
    if not os.path.isdir("/root/reference"):
        pytest.skip("needs /root/reference (CPU container only)")
    from redtail_amd import build
    app = build.build_sample_app()
    assert app and os.path.exists(app)
    deps = subprocess.run(["ldd", app], capture_output=True, text=True).stdout
    assert "libnvstereo_inference.so" in deps and "not found" not in deps
    for f in ("main.cpp", "networks.h"):
        assert not os.path.exists(os.path.join(ROOT, "apps", f)) and not os.path.exists(os.path.join(ROOT, f))
