"""The REFERENCE's own plugin test suite -- stereoDNN/tests/tests_main.cpp: 23 googletest cases that build a one-plugin
TensorRT network per case through the public C++ API (IPluginContainer::create*Plugin, addPlugin / addPluginExt,
addShuffle for 4-D inputs, buildCudaEngine -> createExecutionContext -> execute -> destroy) and compare with its
TensorFlow-generated golden tensors at its own tolerances -- compiled UNTOUCHED against our headers and the test-only
googletest / OpenCV subsets of tests/shim/, linked to our libraries (redtail_amd/build.py:build_reference_tests) and run
as a process.  Includes the two kHALF cases (ELU on fp16 NCHW, correlation on fp16 NC2HW2), for which the executor
converts the fp32 bindings around the plugin like TensorRT's reformat layers do."""
import os
import subprocess

import pytest

from redtail_amd import build, model_files

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(binary, flt=None):
    cmd = [binary, model_files.tests_data_dir()] + (["--gtest_filter=" + flt] if flt else [])
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    return res.returncode, res.stdout + res.stderr


def test_reference_plugin_tests_on_emulator():
    """CPU tier: every case except the 1 GB performance case, on the SIMT emulator build of the libraries"""
    if not os.path.isdir("/root/reference"):
        pytest.skip("needs /root/reference (CPU container only)")
    binary = build.build_reference_tests(emu=True)
    rc, out = run(binary, "-*PerfTests*")
    assert rc == 0, out[-4000:]
    assert "[==========] 22 tests ran." in out and "[  PASSED  ] 22 tests." in out, out[-2000:]


@pytest.mark.gpu
def test_reference_plugin_tests_on_gpu():
    """all 23 cases, incl. CostVolumePluginPerfTests.NVSmall (the 1 GB cost volume of NVSmall through the plugin)"""
    binary = os.path.join(ROOT, "oracle", "_ref", "nvstereo_tests")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/nvstereo_tests not built (no /root/reference at build time)")
    rc, out = run(binary)
    print(out[-3000:])
    assert rc == 0, out[-4000:]
    assert "[==========] 23 tests ran." in out and "[  PASSED  ] 23 tests." in out, out[-2000:]
