"""The C-ABI libraries must load and export every function declared in include/*.h -- on any machine,
GPU or not (no compute calls here)."""
import ctypes
import os
import re

import pytest

from redtail_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("which", ["hip", "emu"])
def test_kernel_library_exports_rt_stereo_h(which):
    path = build.build_hip() if which == "hip" else build.build_emu()
    lib = ctypes.CDLL(path)
    names = declared("rt_stereo.h")
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert set(names) == set(capi.KERNEL_SYMBOLS), set(names) ^ set(capi.KERNEL_SYMBOLS)


@pytest.mark.parametrize("which", ["hip", "emu"])
def test_host_library_exports_rt_stereo_net_h(which):
    path = build.build_host() if which == "hip" else build.build_host_emu()
    lib = ctypes.CDLL(path)
    names = declared("rt_stereo_net.h")
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert set(names) == set(capi.NET_SYMBOLS)
    # the C++ API of redtail_tensorrt_plugins.h is exported too (mangled): spot-check the factory and a helper
    syms = os.popen("nm -D --defined-only %s | c++filt" % path).read()
    for s in ("redtail::tensorrt::IPluginContainer::create(nvinfer1::ILogger&)",
              "redtail::tensorrt::addCostVolume(", "redtail::tensorrt::addConv3DTranspose(",
              "redtail::tensorrt::StereoDnnPluginFactory::createPlugin(",
              "nvinfer1::createInferBuilder(nvinfer1::ILogger&)",
              "redtail::tensorrt::createResNet18_2D_513x257Network("):
        assert s in syms, s


def test_product_loader_refuses_to_run_without_a_gpu():
    """no silent CPU fallback: on a machine without a HIP device the product loader raises"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(capi.RtError):
        capi.KernelLib()
