"""GPU-only parity at BASELINE.json's full sizes (ResNet-18 2D @1257x369 -> half-res 629x185), against the
oracle on the same seeded inputs, plus size-independent properties (linearity of the correlation,
fused == unfused, soft-argmax of a one-hot volume)."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi

pytestmark = pytest.mark.gpu
C, H, W, D = 32, 185, 629, 48


@pytest.fixture(scope="module")
def klib():
    return capi.KernelLib()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rnd(seed, *shape):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def test_corr_full_size(klib):
    l, r = rnd(1, 2, C, H, W), rnd(2, 2, C, H, W)
    cv = torch.full((2, D, H, W), float("nan"), device="cuda")
    klib.corr_cost_volume(dev(l), dev(r), cv, 2, C, H, W, D)
    ref = O.corr_cost_volume(torch.from_numpy(l), torch.from_numpy(r), D)
    assert (cv.cpu() - ref).abs().max().item() <= 5e-5
    # linearity in the left operand: corr(a*L1 + L2, R) == a*corr(L1,R) + corr(L2,R)
    l2 = rnd(3, 2, C, H, W)
    cv2, cv3 = torch.empty_like(cv), torch.empty_like(cv)
    klib.corr_cost_volume(dev(l2), dev(r), cv2, 2, C, H, W, D)
    klib.corr_cost_volume(dev(0.5 * l + l2), dev(r), cv3, 2, C, H, W, D)
    assert (cv3 - (0.5 * cv + cv2)).abs().max().item() <= 1e-4


def test_fused_equals_unfused_full_size(klib):
    l, r = dev(rnd(4, 1, C, H, W)), dev(rnd(5, 1, C, H, W))
    cv = torch.empty(1, D, H, W, device="cuda")
    a, b = torch.empty(1, 1, H, W, device="cuda"), torch.empty(1, 1, H, W, device="cuda")
    klib.corr_cost_volume(l, r, cv, 1, C, H, W, D)
    klib.softargmax(cv, a, 1, D, H, W, False)
    klib.corr_softargmax(l, r, b, 1, C, H, W, D, False)
    ref = O.softargmax(cv.cpu(), False)
    assert (a.cpu() - ref).abs().max().item() <= 1e-4
    assert (a - b).abs().max().item() <= 2e-4


def test_softargmax_one_hot(klib):
    """a volume that is 1e4 at disparity d*(y,x) and 0 elsewhere must regress exactly d*"""
    idx = torch.randint(0, D, (1, 1, H, W), generator=torch.Generator().manual_seed(0))
    vol = torch.zeros(1, D, H, W).scatter_(1, idx, 1e4).cuda()
    out = torch.empty(1, 1, H, W, device="cuda")
    klib.softargmax(vol, out, 1, D, H, W, False)
    assert torch.equal(out.cpu(), idx.float())
    klib.softargmax(-vol, out, 1, D, H, W, True)
    assert torch.equal(out.cpu(), idx.float())


@pytest.mark.parametrize("cin,cout,h,w,k,stride,pad,batch,resid", [
    (32, 32, H, W, 3, 1, 1, 2, True),          # the 34 resblock convs
    (3, 32, 369, 1257, 5, 2, 2, 1, False),     # conv1
    (33, 32, H, W, 3, 1, 1, 1, False),         # conv2D_1 on the concat
    (32, 64, H, W, 3, 2, 1, 1, False),
    (64, 128, 93, 315, 3, 2, 1, 1, False),
    (128, 128, 47, 158, 3, 1, 1, 1, False),
])
def test_conv2d_full_size(klib, cin, cout, h, w, k, stride, pad, batch, resid):
    x = rnd(10, batch, cin, h, w)
    wt = rnd(11, cout, cin, k, k) * np.float32(1 / np.sqrt(cin * k * k))
    b = rnd(12, cout)
    plan = klib.conv2d_plan(wt, b, cin, cout, h, w, k, stride, pad, act=capi.RT_ACT_ELU, has_residual=resid)
    ref = O.conv2d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), stride, pad)
    res = rnd(13, *ref.shape) if resid else None
    if resid:
        ref = ref + torch.from_numpy(res)
    ref = O.elu(ref)
    y = torch.full(tuple(ref.shape), float("nan"), device="cuda")
    plan.enqueue(dev(x), y, dev(res) if resid else None, batch)
    assert (y.cpu() - ref).abs().max().item() <= 5e-5
    plan.destroy()


@pytest.mark.parametrize("cin,cout,h,w", [(128, 64, 47, 158), (64, 32, 93, 315), (32, 1, H, W)])
def test_deconv2d_full_size(klib, cin, cout, h, w):
    x = rnd(20, 1, cin, h, w)
    wt = rnd(21, cin, cout, 3, 3) * np.float32(1 / np.sqrt(cin * 9 / 4))
    b = rnd(22, cout)
    plan = klib.conv2d_plan(wt, b, cin, cout, h, w, 3, 2, 1, transposed=True)
    ref = O.deconv2d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), 2, 1)
    y = torch.full(tuple(ref.shape), float("nan"), device="cuda")
    plan.enqueue(dev(x), y, None, 1)
    assert (y.cpu() - ref).abs().max().item() <= 5e-5
    plan.destroy()


def test_conv3d_nvtiny_size(klib):
    """conv3D_1 of NVTiny: (24,16,81,257) -> (16,24,81,257) (nvtiny_513x161_net.cpp:173-180)"""
    x = rnd(30, 1, 24, 16, 81, 257)
    w = rnd(31, 16, 3, 16, 3, 3) * np.float32(1 / np.sqrt(27 * 16))
    b = rnd(32, 16)
    plan = klib.conv3d_plan(w, b, 16, 16, (24, 81, 257), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1))
    y = torch.full((1,) + plan.out_dims, float("nan"), device="cuda")
    plan.enqueue(dev(x), y, None, 1)
    ref = O.conv3d_tf(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), (1, 1, 1), (1, 1, 1), (1, 1, 1))
    assert (y.cpu() - ref).abs().max().item() <= 5e-5
    plan.destroy()
