"""Guards of the cure for the round-3 / round-4 nondeterminism (profiles/r04_race.txt): the product library is built
without the SLP vectoriser, and nothing else may put packed fp32 math back into a kernel that can share a CU with fp16-MFMA waves."""
import os
import sys

import numpy as np
import pytest

from redtail_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_library_has_no_packed_fp32_outside_the_allow_list():
    """disassembles librt_stereo_hip.so (llvm-objdump on its gfx950 code object): v_pk_{add,mul,fma}_f32 only in the kernels that write
    them in their source (Winograd input transform, stand-alone element-wise plugins)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import check_no_packed_f32 as guard
    finally:
        sys.path.pop(0)
    counts, bad, kernels = guard.check(build.build_hip())
    assert kernels > 200 and not bad, bad
    assert any("conv_wino_f32_kernel" in k for k in counts)          # the scan does see packed math where it is allowed


def test_device_build_without_the_flag_pair_is_refused(tmp_path):
    """rt_capi.hip carries the contract itself: a device compile without -fno-slp-vectorize -DRT_BUILT_NO_SLP stops at an #error"""
    import subprocess
    out = subprocess.run([build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fsyntax-only", os.path.join(build.CSRC, "rt_capi.hip")],
                         capture_output=True, text=True)
    assert out.returncode != 0 and "RT_BUILT_NO_SLP" in out.stderr


@pytest.mark.gpu
def test_interleaved_winograd_is_deterministic_beside_fp16_mfma_waves():
    """the two-kernel reproducer of profiles/r04_race.txt (tools/race_pair.py) as a test: the exact-fp32 Winograd plan on interleaved tensors
    runs 2000 times on one stream while split-fp16 convolutions (v_mfma_f32_32x32x16_f16) run on three others; with SLP-packed epilogue
    math 1662 of 4000 victim launches deviated, the product must give the same bits every time"""
    import torch
    k = capi.KernelLib()
    rng = np.random.default_rng(5)
    H, W, P = 185, 629, 640

    def conv(flags):
        wt = (rng.standard_normal(32 * 32 * 9) / np.sqrt(288)).astype(np.float32)
        plan = k.conv2d_plan(wt, rng.standard_normal(32).astype(np.float32), 32, 32, H, W, 3, 1, 1, act=capi.RT_ACT_ELU, has_residual=True, flags=flags)
        plan.set_pitch(P, P)
        plan.set_layouts(1, 1, 1)
        x, r = torch.randn(1, 32, H, P, device="cuda"), torch.randn(1, 32, H, P, device="cuda")
        y = torch.zeros(1, 32, H, P, device="cuda")
        return plan, x, y, r

    victim, vx, vy, vr = conv(capi.RT_CONV_EXACT_FP32)
    aggressors = [conv(0) for _ in range(3)]
    vs, streams = torch.cuda.Stream(), [torch.cuda.Stream() for _ in range(3)]
    victim.enqueue(vx, vy, vr, 1, stream=vs.cuda_stream)
    torch.cuda.synchronize()
    first = vy.clone()
    bad = 0
    for it in range(2000):
        for (plan, x, y, r), s in zip(aggressors, streams):
            plan.enqueue(x, y, r, 1, stream=s.cuda_stream)
        with torch.cuda.stream(vs):
            vy.zero_()
        victim.enqueue(vx, vy, vr, 1, stream=vs.cuda_stream)
        with torch.cuda.stream(vs):
            bad += int(not torch.equal(vy, first))          # (waits for the victim's stream only: the aggressors keep running)
    torch.cuda.synchronize()
    assert bad == 0, "%d of 2000 victim launches deviated" % bad
