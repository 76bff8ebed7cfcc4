import os as _os
_os.environ.setdefault("RT_DEV_KNOBS", "1")      # the tests' A/B switches (RT_NO_FUSION, RT_RB, ...) are development knobs: opt in
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """The reference's 58 TF-generated tensors (tests/golden/make_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "redtail_fixtures.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


# ---------------------------------------------------------------------------------------------
# Backends for the C-ABI parity tests: the real HIP library on a GPU (-m gpu) and, for the CPU
# tier, the very same kernel sources compiled against the SIMT emulator in tests/emu.
# ---------------------------------------------------------------------------------------------
class _Backend:
    name = "?"

    def run(self):
        pass


class EmuBackend(_Backend):
    name = "emu"

    def __init__(self):
        from redtail_amd import build, capi
        self.klib = capi.KernelLib(build.build_emu())

    def dev(self, a):
        return np.ascontiguousarray(np.asarray(a, dtype=np.float32)).copy()

    def empty(self, shape):
        return np.full(shape, np.nan, dtype=np.float32)

    def host(self, t):
        return np.asarray(t)

    def host_ptr(self, a):
        return a


class GpuBackend(_Backend):
    name = "gpu"

    def __init__(self):
        import torch
        from redtail_amd import capi
        assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
        self.torch = torch
        self.klib = capi.KernelLib()          # raises if librt_stereo_hip.so is missing

    def dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).cuda()

    def empty(self, shape):
        return self.torch.full(tuple(shape), float("nan"), dtype=self.torch.float32, device="cuda")

    def host(self, t):
        self.torch.cuda.synchronize()
        return t.cpu().numpy()

    def host_ptr(self, a):
        return a


_backends = {}


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    if request.param not in _backends:
        _backends[request.param] = EmuBackend() if request.param == "emu" else GpuBackend()
    return _backends[request.param]
