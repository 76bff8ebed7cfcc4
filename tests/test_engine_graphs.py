"""Executor passes (two-stream schedule, concatenation folding, siamese merge) on graphs the four Stereo DNN models do not contain:
tests/cpp/engine_graph_tests.cpp builds them through the public C++ API and compares the default engine with a layer-by-layer,
single-stream build of the same network (and, for the siamese merge, bit for bit with separate tower launches)."""
import os
import subprocess

import pytest

from redtail_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(binary, quick=False):
    env = dict(os.environ, RT_TEST_QUICK="1") if quick else None
    res = subprocess.run([binary], capture_output=True, text=True, timeout=1200, env=env)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-4000:]
    assert "PASSED 4 of 4 engine graph tests" in out, out[-2000:]


def test_engine_graphs_on_emulator():
    _run(build.build_engine_tests(emu=True), quick=True)       # (batch 2 of the siamese cases runs in the GPU tier: CPU-tier time)


@pytest.mark.gpu
def test_engine_graphs_on_gpu():
    binary = os.path.join(ROOT, "tools", "build", "engine_graph_tests")
    assert os.path.exists(binary), "tools/build/engine_graph_tests not built (__graft_entry__.build())"
    _run(binary)
