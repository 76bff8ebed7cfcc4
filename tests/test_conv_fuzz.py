"""Seeded random sweep over convolution shapes through the C ABI: every case picks its own channel counts, image
size, stride, batch, epilogue and row pitches, so tile edges, channel tails, kernel selection (Winograd / direct MFMA /
small-output kernels, merged transposed-conv phases) and the buffer range checks are exercised far from the shapes of
the networks.  Runs on the SIMT emulator (CPU tier) and on the GPU (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import stereo_oracle as O
from redtail_amd import capi
from test_ops_parity import T, near
from test_pitch_parity import pitched


def case2d(seed):
    r = np.random.default_rng(1000 + seed)
    tr = bool(r.integers(0, 4) == 0)
    k, stride = (3, 2) if tr else [(3, 1), (3, 1), (3, 2), (5, 2)][r.integers(0, 4)]
    cin = int(r.choice([1, 3, 5, 8, 16, 31, 32, 33, 48, 64]))
    cout = int(r.choice([1, 2, 3, 8, 24, 31, 32, 33, 40, 64, 70]))
    h, w = int(r.integers(3, 20)), int(r.integers(3, 75))
    act = int(r.choice([capi.RT_ACT_NONE, capi.RT_ACT_ELU, capi.RT_ACT_SIGMOID]))
    return dict(tr=tr, k=k, stride=stride, cin=cin, cout=cout, h=h, w=w, act=act, resid=bool(r.integers(0, 2)),
                batch=int(r.integers(1, 3)), ip=bool(r.integers(0, 2)), op=bool(r.integers(0, 2)), seed=seed)


@pytest.mark.parametrize("seed", range(36))
def test_conv2d_random(backend, seed):
    c = case2d(seed)
    r = np.random.default_rng(seed)
    rnd = lambda *s: r.standard_normal(s).astype(np.float32)
    pad = c["k"] // 2 if not c["tr"] else 1
    x, b = rnd(c["batch"], c["cin"], c["h"], c["w"]), rnd(c["cout"])
    scale = np.float32(1 / np.sqrt(c["cin"] * c["k"] ** 2))
    if c["tr"]:
        wt = rnd(c["cin"], c["cout"], c["k"], c["k"]) * scale
        ref = O.deconv2d(T(x), T(wt), T(b), c["stride"], pad)
    else:
        wt = rnd(c["cout"], c["cin"], c["k"], c["k"]) * scale
        ref = O.conv2d(T(x), T(wt), T(b), c["stride"], pad)
    res = rnd(*ref.shape) if c["resid"] else None
    if c["resid"]:
        ref = ref + T(res)
    ref = O.elu(ref) if c["act"] == capi.RT_ACT_ELU else (torch.sigmoid(ref) if c["act"] == capi.RT_ACT_SIGMOID else ref)
    ref = ref.numpy()
    wo = ref.shape[-1]
    ip = (c["w"] + 31) // 32 * 32 if c["ip"] else 0
    op = (wo + 31) // 32 * 32 if c["op"] else 0
    plan = backend.klib.conv2d_plan(wt, b, c["cin"], c["cout"], c["h"], c["w"], c["k"], c["stride"], pad, act=c["act"],
                                    has_residual=c["resid"], transposed=c["tr"])
    plan.set_pitch(ip, op)
    y = backend.empty(ref.shape[:-1] + (op if op else wo,))
    plan.enqueue(backend.dev(pitched(x, ip) if ip else x), y,
                 backend.dev(pitched(res, op) if op else res) if c["resid"] else None, c["batch"])
    out = backend.host(y)
    near(out[..., :wo], ref, 3e-5)
    if op:
        assert np.isnan(out[..., wo:]).all(), (c, "padding columns were written")
    plan.destroy()


def case3d(seed):
    r = np.random.default_rng(2000 + seed)
    tr = bool(r.integers(0, 2))
    c_in, c_out = int(r.choice([1, 4, 8, 16])), int(r.choice([1, 2, 8, 24, 32]))
    return dict(tr=tr, cin=c_in, cout=c_out, d=int(r.integers(2, 6)), h=int(r.integers(3, 10)), w=int(r.integers(3, 40)),
                stride=int(r.choice([1, 2])), dchw=bool(r.integers(0, 2)), seed=seed)


@pytest.mark.parametrize("seed", range(16))
def test_conv3d_random(backend, seed):
    c = case3d(seed)
    r = np.random.default_rng(seed)
    rnd = lambda *s: r.standard_normal(s).astype(np.float32)
    s = c["stride"]
    if not c["tr"]:
        x = rnd(1, c["d"], c["cin"], c["h"], c["w"])
        w, b = rnd(c["cout"], 3, c["cin"], 3, 3) * np.float32(1 / np.sqrt(27 * c["cin"])), rnd(c["cout"])
        ref = O.conv3d_tf(T(x), T(w), T(b), (s, s, s), (1, 1, 1), (1, 1, 1))
        if c["dchw"]:
            ref = O.transform(ref)
        plan = backend.klib.conv3d_plan(w, b, c["cin"], c["cout"], (c["d"], c["h"], c["w"]), (3, 3, 3), (s, s, s), (1, 1, 1),
                                        (1, 1, 1), act=capi.RT_ACT_ELU, out_dchw=c["dchw"])
        src = x
    else:
        ydims = (c["d"], c["h"], c["w"])
        out_sp = tuple((n - 1) * 2 - 2 + 3 for n in ydims)                  # stride 2, symmetric (1,1,1) pads
        y = rnd(1, c["cin"], *ydims)                                        # K = cin input channels
        w, b = rnd(c["cin"], 3, c["cout"], 3, 3) * np.float32(1 / np.sqrt(27 * c["cin"] / 8)), rnd(c["cout"])
        od = (out_sp[0], c["cout"], out_sp[1], out_sp[2])
        ref = O.conv3d_transpose_tf(T(y), T(w), T(b), od, (2, 2, 2), (1, 1, 1), (1, 1, 1))
        if c["dchw"]:
            ref = O.transform(ref)
        plan = backend.klib.conv3d_plan(w, b, c["cout"], c["cin"], out_sp, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1),
                                        act=capi.RT_ACT_ELU, out_dchw=c["dchw"], transposed_in_dims=ydims)
        src = y
    ref = O.elu(ref).numpy()
    assert plan.out_dims == tuple(ref.shape[1:])
    out = backend.empty(ref.shape)
    plan.enqueue(backend.dev(src), out, None, 1)
    near(backend.host(out), ref, 3e-5)
    plan.destroy()
