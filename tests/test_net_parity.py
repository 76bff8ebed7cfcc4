"""End-to-end parity of the whole inference path (NvInfer.h shim -> plugins -> fused executor -> HIP
kernels, driven through include/rt_stereo_net.h) against the oracle's op-for-op restatement of the
reference's generated networks.  BASELINE tolerance: 1e-3 abs on the network's raw `disp` output.
CPU tier: tiny images on the SIMT emulator; GPU tier (-m gpu): the real sizes incl. 1257x369."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import stereo_oracle as O
from redtail_amd import build, capi, model_files

_nets = {}


def real_weights(model, fp16=False):
    """the reference's trained weight file (weights/_ref/, staged by build()); FileNotFoundError -- never a synthetic
    fall-back -- when it is not there"""
    return O.read_weights(model_files.weight_file(model, fp16), fp16=fp16)


def netlib(kind):
    if kind not in _nets:
        if kind == "emu":
            _nets[kind] = capi.NetLib(build.build_host_emu(), build.build_emu())
        else:
            _nets[kind] = capi.NetLib()
    return _nets[kind]


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def rt(request):
    class R:
        kind = request.param
        lib = netlib(request.param)

        def dev(self, a):
            a = np.ascontiguousarray(a, dtype=np.float32)
            return a.copy() if self.kind == "emu" else torch.from_numpy(a).cuda()

        def empty(self, *shape):
            return np.full(shape, np.nan, np.float32) if self.kind == "emu" else torch.full(shape, float("nan"), device="cuda")

        def host(self, t):
            if self.kind == "emu":
                return t
            torch.cuda.synchronize()
            return t.cpu().numpy()
    return R()


def pairs(n, h, w, seed=1234):
    ls, rs = zip(*(O.synth_pair(h, w, seed + i) for i in range(n)))
    return np.stack(ls), np.stack(rs)


def run_net(rt, model, weights, l, r, **kw):
    n, _, h, w = l.shape
    net = rt.lib.create(model, w, h, max_batch=n, weights=weights, **kw)
    out = rt.empty(n, 1, h, w)
    net.execute(rt.dev(l), rt.dev(r), out, n)
    res = np.array(rt.host(out))
    info = (net.num_layers, net.num_launches)
    net.destroy()
    return res, info


def test_resnet18_2d_tiny(rt):
    """all fusions on; 41x25 image, batch 2, D = 8"""
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(2, 25, 41)
    out, (layers, launches) = run_net(rt, "resnet18_2D", w, l, r, max_disp=8)
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), w, max_disp=8).numpy()
    assert not np.isnan(out).any()
    assert np.abs(out - ref).max() <= 1e-3, np.abs(out - ref).max()
    assert np.abs(out - ref).max() <= 2e-4            # fp32 path: far inside the budget
    assert layers > 100 and launches < layers / 2     # conv+add+ELU, corr+softargmax fused


def test_resnet18_2d_fused_residual_blocks(rt, monkeypatch):
    """RT_RB=1: every residual block of the two towers as ONE launch of the streaming kernel (round 4: also the left tower's first block,
    whose input -- a member of conv2D_1's concatenation -- now stays interleaved as the host of that concatenation); 73 x 41 image = two strips
    and two segments at half resolution (37 x 21) with 16-row segments, one with 32.  Same numbers as layer by layer."""
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(1 if rt.kind == "emu" else 2, 41, 73)          # (one pair on the emulator: CPU tier time)
    base, (_, launches0) = run_net(rt, "resnet18_2D", w, l, r, max_disp=8)
    assert launches0 == 48 - 18                            # siamese merge: 16 block convolutions + encoder2D_out + (round 6) the first layers of the two towers pair up
    monkeypatch.setenv("RT_RB", "1")
    for seg in ("16", "32"):
        monkeypatch.setenv("RT_RBS_SEG", seg)
        out, (layers, launches) = run_net(rt, "resnet18_2D", w, l, r, max_disp=8)
        # 16 blocks, two launches -> one; blocks 1-8 + encoder2D_out: two towers -> one
        assert launches == 48 - 16 - 10
        assert not np.isnan(out).any()
        assert np.abs(out - base).max() <= 2e-5, np.abs(out - base).max()
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), w, max_disp=8).numpy()
    assert np.abs(out - ref).max() <= 2e-4


def test_resnet18_2d_presplit_tensors_between_tower_blocks(rt, monkeypatch):
    """Round 6: the tensors that travel from one fused tower block to the next are stored pre-split (fp16 hi / lo operand pairs in the fp32
    tensor's bytes, rt_resblock_plan_set_split; conv_rbd.hip.h) -- seven of the eight merged block launches of a pair write or read one.  Same
    launches as with RT_NO_RBD=1 (fp32 tensors between the blocks), the stored bytes of exactly those tensors differ, the disparity agrees
    to fp32 rounding (the skip connections carry 22 bits) and with the oracle."""
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(1, 41, 73)
    monkeypatch.setenv("RT_RB", "1")
    monkeypatch.setenv("RT_RBS_SEG", "16")
    res = {}
    for no_rbd in ("0", "1"):
        if no_rbd == "1":
            monkeypatch.setenv("RT_NO_RBD", "1")
        net = rt.lib.create("resnet18_2D", 73, 41, weights=w, max_disp=8)
        net.set_launch_trace(True)
        out = rt.empty(1, 1, 41, 73)
        net.execute(rt.dev(l), rt.dev(r), out, 1)
        res[no_rbd] = (np.array(rt.host(out)), net.read_launch_trace(), [net.launch_name(i) for i in range(net.num_launches)])
        net.destroy()
    (o_split, h_split, names), (o_f32, h_f32, names2) = res["0"], res["1"]
    assert names == names2
    differ = [n for n, a, b in zip(names, h_split, h_f32) if a != b]
    # blocks 1 .. 7 of the merged towers write a pre-split tensor; block 8 (fp32 out) and everything behind it differ in rounding only
    blocks = [n for n in names if "resblock" in n]
    assert len(blocks) == 8 and all(n in differ for n in blocks[:7]), (blocks, differ)
    assert not any(a != b for n, a, b in zip(names, h_split, h_f32) if names.index(n) < names.index(blocks[0]))
    assert np.abs(o_split - o_f32).max() <= 5e-6, np.abs(o_split - o_f32).max()
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), w, max_disp=8).numpy()
    assert np.abs(o_split - ref).max() <= 2e-4


def test_resnet18_2d_siamese_merge_is_bit_identical(rt, monkeypatch):
    """The two feature towers share their weights (left_* == right_* in the reference's weight files): launches of twin layers whose
    tensors are plain internal buffers run as ONE launch over [left samples | right samples] (EngineImpl::mergeSiamese).  Same bits
    as the separate launches (RT_NO_SIAMESE=1), batch 1 and a batch below maxBatchSize, layer by layer and with fused blocks;
    towers whose weights differ are not merged."""
    w = O.synth_weights_resnet18_2d()
    for n, maxb, env in ((1, 1, {}), (2, 3, {}), (2, 2, {"RT_RB": "1", "RT_RBS_SEG": "16"})):
        if rt.kind == "emu" and (n, maxb) == (2, 3):
            continue                                              # (a batch below maxBatchSize: GPU tier and tests/cpp/engine_graph_tests.cpp; CPU-tier time)
        hh, ww = (41, 73) if (env or rt.kind != "emu") else (25, 41)      # the streaming block needs two strips x two segments; otherwise small on the emulator
        l, r = pairs(n, hh, ww)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs, counts = [], []
        for siamese in (True, False):
            if siamese:
                monkeypatch.delenv("RT_NO_SIAMESE", raising=False)
            else:
                monkeypatch.setenv("RT_NO_SIAMESE", "1")
            net = rt.lib.create("resnet18_2D", ww, hh, max_batch=maxb, weights=w, max_disp=8)
            out = rt.empty(n, 1, hh, ww)
            net.execute(rt.dev(l), rt.dev(r), out, n)
            outs.append(np.array(rt.host(out)))
            counts.append(net.num_launches)
            net.destroy()
        monkeypatch.delenv("RT_NO_SIAMESE", raising=False)
        assert counts[0] < counts[1], counts
        assert np.array_equal(outs[0], outs[1]), np.abs(outs[0] - outs[1]).max()
        for k in env:
            monkeypatch.delenv(k, raising=False)
    # different weights on the two sides: nothing may be merged
    w2 = dict(w)
    k = "right_resblock3_conv1_k"
    w2[k] = (np.asarray(w[k]) * 1.5).astype(np.float32)
    l, r = pairs(1, 25, 41)
    out, (_, launches) = run_net(rt, "resnet18_2D", w2, l, r, max_disp=8)
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), w2, max_disp=8).numpy()
    assert np.abs(out - ref).max() <= 2e-4
    assert launches == 48 - 5                              # only the first layers and blocks 1 and 2 (before the differing layer) pair up: twins need twin inputs


@pytest.mark.parametrize("ksplit", [None, "0"])
def test_resnet18_2d_one_stream_per_context(rt, monkeypatch, ksplit):
    """IExecutionContext::setExecutionStreams(1) (rt_net_set_streams): every launch on the caller's stream -- the throughput set-up of
    bench.py -- and back.  A one-stream context passes the throughput hint, under which small launches do not split their contraction
    over wave groups (another fp32 summation order: equal to 1e-5 px); with that split off (RT_S3_KSPLIT=0) the two schedules agree
    bit for bit."""
    if ksplit is not None:
        monkeypatch.setenv("RT_S3_KSPLIT", ksplit)
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(1, 25, 41)
    net = rt.lib.create("resnet18_2D", 41, 25, max_batch=1, weights=w, max_disp=8)
    outs = []
    for streams in (2, 1, 2):
        net.set_streams(streams)
        out = rt.empty(1, 1, 25, 41)
        net.execute(rt.dev(l), rt.dev(r), out, 1)
        outs.append(np.array(rt.host(out)))
    with pytest.raises(capi.RtError):
        net.set_streams(3)
    net.destroy()
    assert not np.isnan(outs[0]).any()
    assert np.array_equal(outs[0], outs[2])
    if ksplit == "0":
        assert np.array_equal(outs[0], outs[1])
    else:
        assert np.abs(outs[0] - outs[1]).max() <= 1e-5


@pytest.mark.parametrize("streams", [2, 1])
def test_resnet18_2d_graph_mode(rt, streams):
    """IExecutionContext::setGraphMode (rt_net_set_graph): the second execute with the same pointers is captured as a hipGraph -- both
    streams of the context inside one capture -- and later ones replay it; new pointers capture another graph; switching the mode off
    returns to direct launches.  Same bits throughout.  (The emulator has no graphs: the context logs that and launches directly.)"""
    if rt.kind == "emu" and streams == 1:
        pytest.skip("the emulator has no graphs; its fallback path is covered by the two-stream case")
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(1, 25, 41)
    l2, r2 = pairs(1, 25, 41, seed=7)
    net = rt.lib.create("resnet18_2D", 41, 25, max_batch=1, weights=w, max_disp=8)
    net.set_streams(streams)
    L, R, L2, R2 = rt.dev(l), rt.dev(r), rt.dev(l2), rt.dev(r2)

    def run(a, b, out=None):
        out = rt.empty(1, 1, 25, 41) if out is None else out
        net.execute(a, b, out, 1)
        return out, np.array(rt.host(out))

    _, ref1 = run(L, R)
    _, ref2 = run(L2, R2)
    assert not np.isnan(ref1).any() and not np.array_equal(ref1, ref2)
    net.set_graph(True)
    emu = rt.kind == "emu"                               # no graphs there: the context says so once and launches directly (fewer passes: CPU tier time)
    out = rt.empty(1, 1, 25, 41)
    for i in range(2 if emu else 4):                     # 1: direct, 2: capture + launch, 3, 4: replay
        _, got = run(L, R, out)
        assert np.array_equal(got, ref1), i
    out2 = rt.empty(1, 1, 25, 41)
    for i in range(1 if emu else 3):                     # other bindings: another graph; the first one stays valid
        _, got = run(L2, R2, out2)
        assert np.array_equal(got, ref2), i
        _, got = run(L, R, out)
        assert np.array_equal(got, ref1), i
    net.set_graph(False)
    _, got = run(L, R, out)
    assert np.array_equal(got, ref1)
    net.destroy()


def test_resnet18_2d_unfused_equals_fused(rt, monkeypatch):
    """RT_NO_FUSION runs every plugin through its own enqueue(), i.e. the reference's layer-by-layer order"""
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(1, 17, 33)
    fused, (_, n1) = run_net(rt, "resnet18_2D", w, l, r, max_disp=6)
    monkeypatch.setenv("RT_NO_FUSION", "1")
    unfused, (_, n2) = run_net(rt, "resnet18_2D", w, l, r, max_disp=6)
    assert n2 > n1
    assert np.abs(fused - unfused).max() <= 1e-5


def test_resnet18_2d_pitched_equals_dense(rt, monkeypatch):
    """internal activations with 128-byte aligned rows (default) vs dense rows (RT_NO_PITCH): same disparity"""
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(2, 25, 41)
    pitched, _ = run_net(rt, "resnet18_2D", w, l, r, max_disp=8)
    monkeypatch.setenv("RT_NO_PITCH", "1")
    dense, _ = run_net(rt, "resnet18_2D", w, l, r, max_disp=8)
    assert not np.isnan(pitched).any()
    # the convolutions only change their addressing; the fused correlation runs on the matrix cores when its feature maps
    # are channel-interleaved (which needs pitched rows) and on the vector ALU otherwise: fp32 roundoff apart
    assert np.abs(pitched - dense).max() <= 2e-6


def test_nvtiny_tiny(rt):
    """3-D path: default cost volume, Conv3D/Transform/Pad/ELU, Conv3DTranspose/Slice/add, softargmin"""
    w = O.synth_weights_3d(O.NVTINY_3D)
    l, r = pairs(1, 25, 33)
    out, (layers, launches) = run_net(rt, "nvtiny", w, l, r, max_disp=4)
    with torch.no_grad():
        ref = O.stereo3d(torch.from_numpy(l), torch.from_numpy(r), w, O.NVTINY_3D, 4).numpy()
    assert not np.isnan(out).any()
    assert np.abs(out - ref).max() <= 1e-3, np.abs(out - ref).max()


def test_nvtiny_unfused_equals_fused(rt, monkeypatch):
    """3-D decoder fusion (Conv3DTranspose + Slice + add + ELU + Transform in one launch) vs every plugin on its own"""
    w = O.synth_weights_3d(O.NVTINY_3D)
    l, r = pairs(1, 33, 65)
    fused, (_, n1) = run_net(rt, "nvtiny", w, l, r, max_disp=8)
    monkeypatch.setenv("RT_NO_FUSION", "1")
    unfused, (_, n2) = run_net(rt, "nvtiny", w, l, r, max_disp=8)
    assert n1 <= 26 < n2, (n1, n2)
    assert np.abs(fused - unfused).max() <= 1e-4


@pytest.mark.parametrize("model,cfg,disp", [("nvtiny", "NVTINY_3D", 8), ("nvsmall", "NVSMALL_3D", 8)])
def test_3d_fp32_interleaved_equals_planar(rt, monkeypatch, model, cfg, disp):
    """fp32 engines keep the 3-D tensors between Conv3D launches (and the decoder's skip tensors) as (D, C/4, H, W, 4): the same arithmetic
    in the same order as on planar tensors, hence the same bits"""
    if rt.kind == "emu" and model == "nvsmall":
        pytest.skip("CPU tier time: NVTiny covers the executor path on the emulator, NVSmall runs in the GPU tier")
    w = O.synth_weights_3d(getattr(O, cfg))
    l, r = pairs(1, 17, 33) if rt.kind == "emu" else pairs(2, 33, 65)
    il, _ = run_net(rt, model, w, l, r, max_disp=disp)
    # (the last layer reads its interleaved input on the matrix cores in split form -- other arithmetic than the vector-ALU fp32 kernel
    #  of the planar engine, the same accuracy; without it the two engines agree bit for bit)
    monkeypatch.setenv("RT_NO_SMALL_IL_F32", "1")
    il_valu, _ = run_net(rt, model, w, l, r, max_disp=disp)
    monkeypatch.setenv("RT_NO_IL8_3D_F32", "1")
    planar, _ = run_net(rt, model, w, l, r, max_disp=disp)
    assert not np.isnan(il).any()
    assert np.array_equal(il_valu, planar)
    assert np.abs(il - planar).max() <= 2e-5, np.abs(il - planar).max()


def test_nvsmall_half2_softargmin_inside_the_last_layer(rt, monkeypatch):
    """half2 NVSmall: disp_softargmax (nvsmall_1025x321_net.cpp:401-425; lib/softargmax_plugin.cpp:167-205) runs inside the depth walk of the
    last Conv3DTranspose (rt_conv_plan_set_softarg, fuseSoftargmax3D in engine.cpp): one launch fewer, no volume tensor, the same map to
    fp32 rounding of the online softmax; RT_NO_SOFTARG_FUSE=1 keeps the volume and the plugin's own launch."""
    w = O.synth_weights_3d(O.NVSMALL_3D)
    l, r = pairs(1, 17, 33) if rt.kind == "emu" else pairs(2, 33, 65)
    fused, (_, n_fused) = run_net(rt, "nvsmall", w, l, r, max_disp=8, fp16_weights=True)
    monkeypatch.setenv("RT_NO_SOFTARG_FUSE", "1")
    plain, (_, n_plain) = run_net(rt, "nvsmall", w, l, r, max_disp=8, fp16_weights=True)
    assert not np.isnan(fused).any()
    assert n_fused == n_plain - 1, (n_fused, n_plain)
    assert np.abs(fused - plain).max() <= 2e-5 * 8, np.abs(fused - plain).max()
    wq = {k: np.asarray(v).astype(np.float16).astype(np.float32) for k, v in w.items()}
    with torch.no_grad():
        ref = O.stereo3d(torch.from_numpy(l), torch.from_numpy(r), wq, O.NVSMALL_3D, 8).numpy()
    # (fp16 tensors on synthetic weights: the distance from the fp32-tensor oracle is the half2 engine's, with or without the fusion)
    err_f, err_p = np.abs(fused - ref).max(), np.abs(plain - ref).max()
    assert err_f <= err_p + 2e-4 and err_f <= 5e-2, (err_f, err_p)


def test_resnet18_2d_half2_mode(rt, monkeypatch):
    """fp16 weight file = half2 mode (sample_app/main.cpp:256-262): activations between the fused launches are stored
    as fp16, arithmetic stays fp32.  Error against the fp32 oracle with the same (fp16-rounded) weights must stay inside
    the reference's fp16 tolerance of 1e-2 (tests_main.cpp:320, 1025) -- measured ~1e-3 on the normalised disparity."""
    w = O.synth_weights_resnet18_2d()
    wq = {k: np.asarray(v).astype(np.float16).astype(np.float32) for k, v in w.items()}
    l, r = pairs(2, 25, 41)
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), wq, max_disp=8).numpy()
    half, _ = run_net(rt, "resnet18_2D", w, l, r, max_disp=8, fp16_weights=True)
    assert not np.isnan(half).any()
    err_half = np.abs(half - ref).max()
    assert err_half <= 1e-2, err_half
    monkeypatch.setenv("RT_NO_F16", "1")                       # same fp16 weights, fp32 activations
    full, _ = run_net(rt, "resnet18_2D", w, l, r, max_disp=8, fp16_weights=True)
    err_full = np.abs(full - ref).max()
    assert err_full <= 2e-4, err_full
    assert err_half > err_full                                 # i.e. the fp16 storage path really ran
    monkeypatch.delenv("RT_NO_F16")
    monkeypatch.setenv("RT_NO_IL8", "1")                       # planar fp16 tensors everywhere
    planar, _ = run_net(rt, "resnet18_2D", w, l, r, max_disp=8, fp16_weights=True)
    # the channel-interleaved tensors inside the towers change addressing only: same operands, same accumulation order -- except (round 6)
    # in the correlation, which runs on the matrix cores for interleaved fp16 maps: the same exact products in another order of summation,
    # so the fp16 soft-argmax map may differ by an ulp, and the disparity by what follows from that
    assert np.abs(planar - half).max() <= 2e-3, np.abs(planar - half).max()
    monkeypatch.delenv("RT_NO_IL8")
    monkeypatch.setenv("RT_NO_CORR_MFMA_F16", "1")             # interleaved towers, planar-style correlation: the bits of the planar engine
    same, _ = run_net(rt, "resnet18_2D", w, l, r, max_disp=8, fp16_weights=True)
    assert np.array_equal(planar, same)


def test_resnet18_2d_half2_fused_residual_blocks(rt, monkeypatch):
    """half2 mode with every tower block as ONE launch (RT_RB=1 forces the fusion on an image this small; at 1257 x 369 it is the default):
    conv_f16rbd_kernel rounds the intermediate to fp16 in LDS as the layer-by-layer path rounds it in HBM and sums in the same order --
    the disparity map has the same bits, from 8 launches fewer (8 merged blocks x 2 layers -> 8 launches).  sample_app/main.cpp:228-266."""
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(1 if rt.kind == "emu" else 2, 41, 73)
    base, (_, n0) = run_net(rt, "resnet18_2D", w, l, r, max_disp=8, fp16_weights=True)
    monkeypatch.setenv("RT_RB", "1")
    for seg in ("16", "32"):
        monkeypatch.setenv("RT_RBS_SEG", seg)
        out, (_, n1) = run_net(rt, "resnet18_2D", w, l, r, max_disp=8, fp16_weights=True)
        assert n1 == n0 - 8, (n0, n1)
        assert np.array_equal(out, base), np.abs(out - base).max()
    # the concatenation in front of conv2D_1 stays interleaved in half2 mode too (round 6: groups of 8 fp16 channels, the map in lane 0 of a
    # fifth group), which is what lets the left tower's first layer and first block join the right tower's launches: 22 launches as in the
    # fp32 engine, 25 without -- addressing only, the same bits
    assert n1 == 22, n1
    monkeypatch.setenv("RT_NO_IL_CONCAT_F16", "1")
    apart, (_, n2) = run_net(rt, "resnet18_2D", w, l, r, max_disp=8, fp16_weights=True)
    assert n2 == 25 and np.array_equal(apart, base), (n2, np.abs(apart - base).max())


def test_nvtiny_fp16_weight_file(rt, monkeypatch):
    """a 3-D model built from an fp16 weight file (NVSmall only ships trt_weights_fp16.bin): the executor tries half2
    mode, the cost-volume plugin cannot take fp16 tensors, every plan already switched must go back to fp32 --
    result = fp32 arithmetic on the fp16-rounded weights (round 1: NaN with a success code)."""
    w = O.synth_weights_3d(O.NVTINY_3D)
    wq = {k: np.asarray(v).astype(np.float16).astype(np.float32) for k, v in w.items()}
    l, r = pairs(1, 25, 33)
    with torch.no_grad():
        ref = O.stereo3d(torch.from_numpy(l), torch.from_numpy(r), wq, O.NVTINY_3D, 4).numpy()
    # half2 mode of a 3-D model: the 4-D tensors between the fused Conv3D / Conv3DTranspose launches are stored as fp16 and
    # multiplied as fp16 operands (fp32 accumulation); the 2-D towers, the last layer's volume and the soft-argmin stay fp32.
    # Reference tolerance for fp16: 1e-2 (tests_main.cpp:320, 1025)
    out, _ = run_net(rt, "nvtiny", w, l, r, max_disp=4, fp16_weights=True)
    assert not np.isnan(out).any()
    err16 = np.abs(out - ref).max()
    assert err16 <= 1e-2, err16
    monkeypatch.setenv("RT_NO_F16_3D", "1")          # same fp16 weights, fp32 tensors everywhere
    out32, _ = run_net(rt, "nvtiny", w, l, r, max_disp=4, fp16_weights=True)
    err32 = np.abs(out32 - ref).max()
    assert err32 <= 1e-3, err32
    assert err16 > err32                              # i.e. the fp16-storage path really ran


def test_plan_round_trip(rt):
    """ICudaEngine::serialize -> IRuntime::deserializeCudaEngine with StereoDnnPluginFactory (sample_app/main.cpp:198-220,
    269-275): the re-created engine must produce the same bits; the 3-D models have no plan, as in the reference
    (their Conv3D / Transform / Pad / Slice plugins are not serialisable)."""
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(2, 25, 41)
    net = rt.lib.create("resnet18_2D", 41, 25, max_batch=2, weights=w, max_disp=8)
    ref = rt.empty(2, 1, 25, 41)
    net.execute(rt.dev(l), rt.dev(r), ref, 2)
    ref = rt.host(ref).copy()
    plan = net.serialize()
    net.destroy()
    assert plan[:8] == b"RTSDPLN1" and len(plan) > 3_000_000          # carries the 3.6 MB of weights
    net2 = rt.lib.create_from_plan(plan, 41, 25)
    out = rt.empty(2, 1, 25, 41)
    net2.execute(rt.dev(l), rt.dev(r), out, 2)
    assert np.array_equal(rt.host(out), ref)
    assert net2.serialize() == plan                                     # and serialises to the same bytes again
    net2.destroy()
    with pytest.raises(capi.RtError):
        rt.lib.create_from_plan(plan[:len(plan) // 2], 41, 25)          # truncated
    with pytest.raises(capi.RtError):
        rt.lib.create_from_plan(b"not a plan" * 10, 41, 25)
    net3 = rt.lib.create("nvtiny", 65, 33, weights=O.synth_weights_3d(O.NVTINY_3D), max_disp=8)
    with pytest.raises(capi.RtError):
        net3.serialize()
    net3.destroy()


def test_bad_inputs_fail_loudly(rt):
    w = O.synth_weights_resnet18_2d()
    with pytest.raises(capi.RtError):
        rt.lib.create("resnet18_2D", 40, 25, weights=w)              # 40 is not 1 (mod 8): asymmetric pad
    bad = dict(w)
    del bad["conv2D_4_k"]
    with pytest.raises(capi.RtError):
        rt.lib.create("resnet18_2D", 41, 25, weights=bad)            # missing tensor
    with pytest.raises(capi.RtError):
        rt.lib.create("resnet18_2D", 41, 25, weights=b"garbage")


# ---- GPU only: real sizes -----------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("w,h,batch", [(513, 257, 1), (1257, 369, 1), (1257, 369, 2), (1241, 377, 1)])      # 1241x377: the other KITTI size
def test_resnet18_2d_full_size(w, h, batch):
    """BASELINE config C2 with the reference's trained weights (ResNet-18_2D/TensorRT/trt_weights.bin)"""
    lib = netlib("gpu")
    weights = real_weights("resnet18_2D")
    l, r = pairs(batch, h, w)
    net = lib.create("resnet18_2D", w, h, max_batch=batch, weights_path=model_files.weight_file("resnet18_2D"))
    out = torch.full((batch, 1, h, w), float("nan"), device="cuda")
    net.execute(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out, batch)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), weights)
    err = (out.cpu() - ref).abs().max().item()
    print("ResNet-18 2D %dx%d batch %d, real weights: max |disp - oracle| = %.3g, %d launches" % (w, h, batch, err, net.num_launches))
    assert err <= 1e-3, err
    # at 1257x369 the executor runs the 16 residual blocks as one (streaming) launch each (48 -> 32) and blocks 1-8 + encoder2D_out of
    # the two towers as one launch over both images (siamese merge: 32 -> 23; round 3: 25, the left tower's first block read a planar member
    # of conv2D_1's concatenation); at 513x257 no block is fused (too few strips) and the 16 block convolutions + encoder2D_out pair up (48 -> 31)
    # (round 6: the towers' first layers read the two bindings in one launch too: 22 / 30)
    assert net.num_launches == (22 if w >= 1241 else 30), net.num_launches
    net.destroy()


@pytest.mark.gpu
def test_resnet18_2d_half2_real_weights():
    """BASELINE config C3: half2 mode, 1257x369, batch 8, the reference's trt_weights_fp16.bin.  Error of the raw
    `disp` output (disparity / width) against (i) the fp32 oracle with the fp32 weight file and (ii) the oracle with the
    fp16 file's weights and fp32 arithmetic.  north_star's budget of 1e-3 is asserted on (ii); (i) adds the
    rounding of the weights themselves and gets the reference's own fp16 tolerance 1e-2 (tests_main.cpp:320, 1025)."""
    lib = netlib("gpu")
    w, h, batch = 1257, 369, 8
    w32, w16 = real_weights("resnet18_2D"), real_weights("resnet18_2D", fp16=True)
    l, r = pairs(batch, h, w)
    net = lib.create("resnet18_2D", w, h, max_batch=batch, weights_path=model_files.weight_file("resnet18_2D", True), fp16_weights=True)
    out = torch.full((batch, 1, h, w), float("nan"), device="cuda")
    net.execute(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out, batch)
    torch.cuda.synchronize()
    out = out.cpu()
    with torch.no_grad():
        ref32 = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), w32)
        ref16 = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), w16)
    e32, e16 = (out - ref32).abs(), (out - ref16).abs()
    print("half2 1257x369 batch 8, real fp16 weights: max err vs fp32 oracle %.3g (mean %.3g), vs fp16-weights oracle %.3g (mean %.3g); "
          "fp16-weights oracle vs fp32 oracle %.3g" % (e32.max(), e32.mean(), e16.max(), e16.mean(), (ref16 - ref32).abs().max()))
    assert not torch.isnan(out).any()
    assert e16.max().item() <= 1e-3, e16.max().item()
    assert e32.max().item() <= 1e-2, e32.max().item()
    net.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("half2", [False, True])
def test_resnet18_2d_full_size_layouts(monkeypatch, half2):
    """BASELINE size 1257x369, batch 2: half2 mode against the oracle with the same fp16-rounded weights (reference
    fp16 tolerance 1e-2, tests_main.cpp:320,1025), and -- fp32 and half2 -- channel-interleaved tensors against planar
    ones (RT_NO_IL8): addressing only, so bit for bit."""
    lib = netlib("gpu")
    w, h, batch = 1257, 369, 2
    weights = O.synth_weights_resnet18_2d()
    l, r = pairs(batch, h, w)
    outs = []
    for no_il in ("0", "1"):
        monkeypatch.setenv("RT_NO_IL8", no_il)
        net = lib.create("resnet18_2D", w, h, max_batch=batch, weights=weights, fp16_weights=half2)
        out = torch.full((batch, 1, h, w), float("nan"), device="cuda")
        net.execute(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out, batch)
        torch.cuda.synchronize()
        outs.append(out.cpu())
        net.destroy()
    if half2:
        assert torch.equal(outs[0], outs[1])
    else:       # fp32: the correlation of interleaved feature maps runs on the matrix cores, of planar ones on the vector ALU; and (round 6)
                # the tensors between the fused tower blocks of the interleaved engine are stored pre-split: 22 bits (conv_rbd.hip.h)
        assert (outs[0] - outs[1]).abs().max().item() <= 5e-6
    if half2:
        wq = {k: np.asarray(v).astype(np.float16).astype(np.float32) for k, v in weights.items()}
        with torch.no_grad():
            ref = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), wq)
        err = (outs[0] - ref).abs().max().item()
        assert err <= 1e-2, err


@pytest.mark.gpu
def test_resnet18_2d_interleaved_contexts():
    """Four engines on four streams, pairs issued round-robin with no synchronisation in between (what bench.py
    times): every context must reproduce the oracle for ITS pair, i.e. no buffers are shared between contexts."""
    lib = netlib("gpu")
    weights = O.synth_weights_resnet18_2d()
    w, h, nctx = 513, 257, 4
    nets = [lib.create("resnet18_2D", w, h, max_batch=1, weights=weights) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    ins = [pairs(1, h, w, seed=100 + i) for i in range(nctx)]
    dl = [torch.from_numpy(l).cuda() for l, _ in ins]
    dr = [torch.from_numpy(r).cuda() for _, r in ins]
    outs = [torch.full((1, 1, h, w), float("nan"), device="cuda") for _ in range(nctx)]
    torch.cuda.synchronize()
    for rep in range(3):
        for i in range(nctx):
            nets[i].execute(dl[i], dr[i], outs[i], 1, stream=streams[i].cuda_stream)
    torch.cuda.synchronize()
    with torch.no_grad():
        for i in range(nctx):
            ref = O.resnet18_2d(torch.from_numpy(ins[i][0]), torch.from_numpy(ins[i][1]), weights)
            err = (outs[i].cpu() - ref).abs().max().item()
            assert err <= 1e-3, (i, err)
    for n in nets:
        n.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("model,cfg,w,h,disp,tol", [("nvsmall", "NVSMALL_3D", 257, 129, 16, 1e-3), ("resnet18", "RESNET18_3D", 257, 129, 12, 1e-3),
                                                    ("nvsmall", "NVSMALL_3D", 1025, 321, 48, 1e-3),
                                                    # 30 layers deep, disparities up to 136 px: fp32 roundoff alone exceeds 1e-3 px -- the test evaluates
                                                    # the graph in fp64 and bounds the GPU's error by 1.5x the fp32 oracle's own (tol = None)
                                                    ("resnet18", "RESNET18_3D", 1025, 321, 68, None)])
def test_3d_models(model, cfg, w, h, disp, tol):
    """NVSmall / ResNet-18 3D (BASELINE configs C5 / C4 in fp32) at a quarter of their resolution and at full size
    (1025x321, D = 48 / 68 at half resolution): Winograd Conv3D, folded Pad, fused Conv3DTranspose decoder and the
    small-output last layer against the oracle graph; budget 1e-3 px on the soft-argmin disparity."""
    lib = netlib("gpu")
    weights = O.synth_weights_3d(getattr(O, cfg))
    l, r = pairs(1, h, w)
    net = lib.create(model, w, h, weights=weights, max_disp=disp)
    out = torch.full((1, 1, h, w), float("nan"), device="cuda")
    net.execute(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out, 1)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.stereo3d(torch.from_numpy(l), torch.from_numpy(r), weights, getattr(O, cfg), disp)
    err = (out.cpu() - ref).abs().max().item()
    net.destroy()
    if tol is not None:
        assert err <= tol, err
        return
    with torch.no_grad():                                     # fp64 evaluation of the same graph (about a minute of CPU)
        ref64 = O.stereo3d(torch.from_numpy(l).double(), torch.from_numpy(r).double(), weights, getattr(O, cfg), disp)
    e_gpu = (out.cpu().double() - ref64).abs().max().item()
    e_oracle = (ref.double() - ref64).abs().max().item()
    print("%s %dx%d: |GPU - fp64| = %.3g px, |fp32 oracle - fp64| = %.3g px, |GPU - fp32 oracle| = %.3g px" % (model, w, h, e_gpu, e_oracle, err))
    assert e_gpu <= max(1.5 * e_oracle, 1e-3), (e_gpu, e_oracle)
    assert err <= e_gpu + e_oracle + 1e-6                      # triangle inequality: nothing else hides in the fp32 comparison


@pytest.mark.gpu
def test_nvtiny_full_size():
    """BASELINE config C1 (NVTiny 513x161) with the reference's trained weights"""
    lib = netlib("gpu")
    weights = real_weights("nvtiny")
    l, r = pairs(1, 161, 513)
    net = lib.create("nvtiny", 513, 161, weights_path=model_files.weight_file("nvtiny"))
    out = torch.full((1, 1, 161, 513), float("nan"), device="cuda")
    net.execute(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out, 1)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.stereo3d(torch.from_numpy(l), torch.from_numpy(r), weights, O.NVTINY_3D, 24)
    err = (out.cpu() - ref).abs().max().item()
    print("NVTiny 513x161, real weights: max |disp - oracle| = %.3g px" % err)
    assert err <= 1e-3, err
    net.destroy()


@pytest.mark.gpu
def test_nvsmall_full_size_real_fp16_weights(monkeypatch):
    """BASELINE config C5's model with the only weight file the reference ships for it (NVSmall trt_weights_fp16.bin),
    1025x321, D = 48 at half resolution, against the oracle on the same (fp16-valued) weights."""
    lib = netlib("gpu")
    weights = real_weights("nvsmall", fp16=True)
    l, r = pairs(1, 321, 1025)
    net = lib.create("nvsmall", 1025, 321, weights_path=model_files.weight_file("nvsmall", True), fp16_weights=True)
    out = torch.full((1, 1, 321, 1025), float("nan"), device="cuda")
    net.execute(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out, 1)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.stereo3d(torch.from_numpy(l), torch.from_numpy(r), weights, O.NVSMALL_3D, 48)
    err = (out.cpu() - ref).abs()
    print("NVSmall 1025x321, real fp16 weights, half2 mode (fp16 3-D tensors): max |disp - oracle| = %.3g px, mean %.3g px, disparities up to %.1f px"
          % (err.max().item(), err.mean().item(), ref.max().item()))
    assert not torch.isnan(out).any()
    # fp16 storage of the two feature maps and of 11 stacked 3-D layers on disparities up to ~100 px: measured 0.19 px at worst / 3e-4 px
    # on average against the fp32-tensor oracle (the reference's accuracy metric D1 counts errors above 3 px), bounded below by the
    # error of the reference's own fp16 mode; with fp32 tensors (RT_NO_F16_3D) the same weights give < 1e-3 px, further down
    # ... and it is the error class of the REFERENCE's own fp16 mode: its Conv3D / Conv3DTranspose plugins convert to fp16 before cuDNN,
    # get an fp16 tensor back, add the bias to it and convert to fp32 (lib/conv3d_plugin.cpp:187-216, 247-274).  The oracle restates
    # that (plugin_fp16=True); our half2 output must be no further from it than twice its own distance from the fp32-tensor oracle.
    with torch.no_grad():
        ref16 = O.stereo3d(torch.from_numpy(l), torch.from_numpy(r), weights, O.NVSMALL_3D, 48, plugin_fp16=True)
    d_ref = (ref16 - ref).abs()
    d_hip = (out.cpu() - ref16).abs()
    print("reference-style fp16 plugins (oracle): max |disp - fp32-tensor oracle| = %.3g px, mean %.3g px; ours against that: max %.3g px, mean %.3g px"
          % (d_ref.max().item(), d_ref.mean().item(), d_hip.max().item(), d_hip.mean().item()))
    assert d_hip.max().item() <= 2 * d_ref.max().item() and d_hip.mean().item() <= 2 * d_ref.mean().item(), (d_hip.max().item(), d_ref.max().item())
    assert err.max().item() <= 2 * d_ref.max().item() and err.mean().item() <= 2 * d_ref.mean().item()      # no worse than the reference's mode
    assert err.max().item() <= 0.25 and err.mean().item() <= 4e-4, (err.max().item(), err.mean().item())    # 1.3x the measured figures
    net.destroy()
    monkeypatch.setenv("RT_NO_F16_3D", "1")
    net = lib.create("nvsmall", 1025, 321, weights_path=model_files.weight_file("nvsmall", True), fp16_weights=True)
    out32 = torch.full((1, 1, 321, 1025), float("nan"), device="cuda")
    net.execute(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out32, 1)
    torch.cuda.synchronize()
    err32 = (out32.cpu() - ref).abs().max().item()
    print("NVSmall 1025x321, real fp16 weights, fp32 tensors: max |disp - oracle| = %.3g px" % err32)
    assert err32 <= 1e-3, err32
    net.destroy()


@pytest.mark.gpu
def test_nvsmall_half2_batch8_full_size_every_pair():
    """BASELINE config C5 AS IT IS TIMED (bench.py --model nvsmall --half2 --batch 8): NVSmall 1025x321, the reference's fp16 weight file,
    half2 mode, one GPU's shard of eight DIFFERENT pairs.  Every pair of the batch must equal, bit for bit, what a batch-1 engine computes
    for that pair alone (so the batch -- other launch shapes, other depth segmentation of the depth-walking Conv3D -- adds nothing), and the
    first pair is bounded against both oracles exactly as test_nvsmall_full_size_real_fp16_weights bounds the batch-1 engine."""
    lib = netlib("gpu")
    weights = real_weights("nvsmall", fp16=True)
    b, h, w = 8, 321, 1025
    l, r = pairs(b, h, w)
    path = model_files.weight_file("nvsmall", True)
    net = lib.create("nvsmall", w, h, max_batch=b, weights_path=path, fp16_weights=True)
    out = torch.full((b, 1, h, w), float("nan"), device="cuda")
    net.execute(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out, b)
    torch.cuda.synchronize()
    out = out.cpu()
    net.destroy()
    assert not torch.isnan(out).any()
    net1 = lib.create("nvsmall", w, h, max_batch=1, weights_path=path, fp16_weights=True)
    for i in range(b):
        o1 = torch.full((1, 1, h, w), float("nan"), device="cuda")
        net1.execute(torch.from_numpy(l[i:i + 1]).cuda(), torch.from_numpy(r[i:i + 1]).cuda(), o1, 1)
        torch.cuda.synchronize()
        assert torch.equal(o1.cpu(), out[i:i + 1]), (i, (o1.cpu() - out[i:i + 1]).abs().max().item())
    net1.destroy()
    assert not torch.equal(out[0], out[1])                    # the pairs do differ
    with torch.no_grad():
        ref = O.stereo3d(torch.from_numpy(l[:1]), torch.from_numpy(r[:1]), weights, O.NVSMALL_3D, 48)
        ref16 = O.stereo3d(torch.from_numpy(l[:1]), torch.from_numpy(r[:1]), weights, O.NVSMALL_3D, 48, plugin_fp16=True)
    err, d_ref, d_hip = (out[:1] - ref).abs(), (ref16 - ref).abs(), (out[:1] - ref16).abs()
    print("NVSmall half2 batch 8, pair 0: max |disp - fp32-tensor oracle| = %.3g px (mean %.3g); reference-style fp16-plugin oracle: %.3g px from it, ours %.3g px from that"
          % (err.max().item(), err.mean().item(), d_ref.max().item(), d_hip.max().item()))
    assert d_hip.max().item() <= 2 * d_ref.max().item() and d_hip.mean().item() <= 2 * d_ref.mean().item()
    assert err.max().item() <= 2 * d_ref.max().item() and err.mean().item() <= 2 * d_ref.mean().item()
    assert err.max().item() <= 0.25 and err.mean().item() <= 4e-4, (err.max().item(), err.mean().item())


@pytest.mark.gpu
@pytest.mark.parametrize("model,cfg,w,h,disp,batch,tol", [("nvsmall", "NVSMALL_3D", 1025, 321, 48, 4, 1e-3), ("resnet18", "RESNET18_3D", 1025, 321, 68, 4, 5e-3),
                                                          ("nvsmall", "NVSMALL_3D", 257, 129, 16, 8, 1e-3), ("resnet18", "RESNET18_3D", 257, 129, 12, 8, 1e-3)])
def test_3d_models_batched(model, cfg, w, h, disp, batch, tol):
    """per-GPU shards of BASELINE configs C5 / C4 (64 / 32 pairs over 8 GPUs = 8 / 4 per GPU): batch 4 at full size,
    batch 8 at a quarter of the resolution, every pair against the oracle.  (ResNet-18 3D at full size: 5e-3 px against the fp32
    oracle, whose own distance from an fp64 evaluation is 2.2e-3 px -- test_3d_models does that comparison in fp64 for one pair;
    here the first pair must also equal the batch-1 engine's result bit for bit, so the batch adds nothing to that error.)"""
    lib = netlib("gpu")
    weights = O.synth_weights_3d(getattr(O, cfg))
    l, r = pairs(batch, h, w)
    net = lib.create(model, w, h, max_batch=batch, weights=weights, max_disp=disp)
    out = torch.full((batch, 1, h, w), float("nan"), device="cuda")
    net.execute(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out, batch)
    torch.cuda.synchronize()
    out = out.cpu()
    net.destroy()
    if tol > 1e-3:                                            # the loose case: tie it to the single-pair engine, which test_3d_models bounds in fp64
        net1 = lib.create(model, w, h, max_batch=1, weights=weights, max_disp=disp)
        out1 = torch.full((1, 1, h, w), float("nan"), device="cuda")
        net1.execute(torch.from_numpy(l[:1]).cuda(), torch.from_numpy(r[:1]).cuda(), out1, 1)
        torch.cuda.synchronize()
        assert torch.equal(out1.cpu(), out[:1]), (out1.cpu() - out[:1]).abs().max().item()
        net1.destroy()
    with torch.no_grad():
        for i in range(batch):
            ref = O.stereo3d(torch.from_numpy(l[i:i + 1]), torch.from_numpy(r[i:i + 1]), weights, getattr(O, cfg), disp)
            err = (out[i:i + 1] - ref).abs().max().item()
            assert err <= tol, (i, err)


def test_resnet18_2d_interleaved_equals_planar(rt, monkeypatch):
    """the executor stores tensors between 3x3 stride-1 layers channel-interleaved (C/4, H, pitch, 4): addressing
    only -- the disparity must not change by a bit against planar tensors (RT_NO_IL8)"""
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(2, 25, 41)
    il, (_, n1) = run_net(rt, "resnet18_2D", w, l, r, max_disp=8)
    monkeypatch.setenv("RT_NO_IL8", "1")
    planar, (_, n2) = run_net(rt, "resnet18_2D", w, l, r, max_disp=8)
    assert n1 == n2 - 3       # (interleaved, the first member of conv2D_1's concatenation is a plain tensor: the left tower's first block merges too,
                              #  and -- round 6 -- so do the towers' first layers, whose outputs then pair up)
    assert np.abs(il - planar).max() <= 2e-6         # bit-identical convolutions; the correlation kernel differs (see above)


def test_development_knobs_need_opt_in(tmp_path):
    """RT_* environment switches (RT_NO_FUSION, RT_RB, ...) are honoured only with RT_DEV_KNOBS=1 -- what this test session sets in
    conftest.py; a process without it builds the default engine whatever its environment says"""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from redtail_amd import build, capi, synth\n"
            "lib = capi.NetLib(build.build_host_emu(), build.build_emu())\n"
            "net = lib.create('resnet18_2D', 41, 25, weights=synth.synth_weights_resnet18_2d(), max_disp=8)\n"
            "print('LAUNCHES', net.num_launches)\n" % ROOT)
    counts = {}
    for dev in ("0", "1"):
        env = dict(os.environ, RT_NO_FUSION="1", RT_DEV_KNOBS=dev)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        counts[dev] = int(out.stdout.split("LAUNCHES")[1].split()[0])
    assert counts["0"] == 48 - 18 and counts["1"] > 100, counts      # ignored / honoured (one launch per layer)


def test_exact_fp32_is_an_api_option(rt, monkeypatch):
    """rtNetOptions.flags = RT_CONV_EXACT_FP32 (IBuilder::setExactFp32Mode): the engine of the RT_CONV_EXACT_FP32=1 development knob,
    bit for bit, without touching the environment; no residual block is fused and the result is within fp32 roundoff of the default"""
    from redtail_amd import capi
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(2, 25, 41)         # batch 2: the residual of the left tower's first block is a channel range of the folded concatenation
                                    # with its own per-sample stride (round 3: the Winograd kernel read it with the output's)

    def run(**kw):
        net = rt.lib.create("resnet18_2D", 41, 25, max_batch=2, weights=w, max_disp=8, **kw)
        out = rt.empty(2, 1, 25, 41)
        net.execute(rt.dev(l), rt.dev(r), out, 2)
        res = np.array(rt.host(out))
        net.destroy()
        return res
    base = run()
    opt = run(flags=capi.RT_CONV_EXACT_FP32)
    monkeypatch.setenv("RT_CONV_EXACT_FP32", "1")
    knob = run()
    assert np.array_equal(opt, knob)
    assert not np.array_equal(opt, base) and np.abs(opt - base).max() <= 2e-5
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(l), torch.from_numpy(r), w, max_disp=8).numpy()
    assert np.abs(opt - ref).max() <= 2e-4


def test_debug_mode_reports_the_fp16_split_domain(rt):
    """|x| >= 65504 overflows the fp16 split (conv_split.hip.h): the default path then produces inf / NaN -- loud but far from the
    cause.  In debug mode (rt_net_set_debug, IExecutionContext::setDebugSync) execute() fails at the first such launch with its name;
    the exact-fp32 engine has no such domain.  Also rt_check_range / rt_conv_plan_input_limit at the operator level."""
    from redtail_amd import capi
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(1, 25, 41)
    big = l.copy()
    big[0, 1, 5, 7] = 7e4                                     # one pixel just outside the fp16 range
    net = rt.lib.create("resnet18_2D", 41, 25, max_batch=2, weights=w, max_disp=8)
    out = rt.empty(1, 1, 25, 41)
    net.execute(rt.dev(big), rt.dev(r), out, 1)               # default: runs, and the damage is visible
    assert not np.isfinite(np.array(rt.host(out))).all()
    net.set_debug(True)
    net.execute(rt.dev(l), rt.dev(r), out, 1)                 # in-domain input passes the checks
    assert np.isfinite(np.array(rt.host(out))).all()
    with pytest.raises(capi.RtError) as e:
        net.execute(rt.dev(big), rt.dev(r), out, 1)
    assert "left_conv1" in str(e.value) and "fp16-split" in str(e.value), str(e.value)
    # batch 2: the checks step through the padded samples of conv2D_1's interleaved 33-channel input (every group of it, the
    # disparity map's included) without a false alarm
    l2, r2 = pairs(2, 25, 41)
    out2 = rt.empty(2, 1, 25, 41)
    net.execute(rt.dev(l2), rt.dev(r2), out2, 2)
    assert np.isfinite(np.array(rt.host(out2))).all()
    net.destroy()
    exact = rt.lib.create("resnet18_2D", 41, 25, weights=w, max_disp=8, flags=capi.RT_CONV_EXACT_FP32)
    exact.set_debug(True)
    exact.execute(rt.dev(big), rt.dev(r), out, 1)             # no such domain: passes the (vacuous) checks and matches the oracle
    with torch.no_grad():
        ref = O.resnet18_2d(torch.from_numpy(big), torch.from_numpy(r), w, max_disp=8).numpy()
    got = np.array(rt.host(out))
    assert np.isfinite(ref).all() and np.isfinite(got).all() and np.abs(got - ref).max() <= 1e-3
    exact.destroy()
    # operator level
    k = rt.lib.kernels
    x = np.zeros((3, 10), np.float32)
    x[:, :8] = np.arange(24, dtype=np.float32).reshape(3, 8)
    x[:, 8:] = np.nan                                         # row padding is not looked at
    x[1, 3] = -70000.0
    mx, bad = k.check_range(rt.dev(x), 3, 8, 10)
    assert bad == 1 and mx == 70000.0
    x[2, 0] = np.inf
    assert k.check_range(rt.dev(x), 3, 8, 10)[1] == 2
    wt = np.zeros((4, 4, 3, 3), np.float32)
    assert k.conv2d_plan(wt, None, 4, 4, 9, 9, 3, 1, 1).input_limit() == 65504.0
    assert k.conv2d_plan(wt, None, 4, 4, 9, 9, 3, 1, 1, flags=capi.RT_CONV_EXACT_FP32).input_limit() == float("inf")


def test_launch_trace_names_the_launch_that_differs(rt):
    """IExecutionContext::setLaunchTrace (rt_net_set_launch_trace): one hash per launch, equal for two passes over the same pair,
    different from the first launch on for another pair -- and only from the first launch that sees the other image on;
    rt_net_read_launch_output returns a launch's tensor as stored (the last launch writes the `disp` binding)."""
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(1, 25, 41)
    l2, _ = pairs(1, 25, 41, seed=99)
    net = rt.lib.create("resnet18_2D", 41, 25, weights=w, max_disp=8)
    net.set_launch_trace(True)
    out = rt.empty(1, 1, 25, 41)
    net.execute(rt.dev(l), rt.dev(r), out, 1)
    a = net.read_launch_trace()
    first = np.array(rt.host(out))
    net.execute(rt.dev(l), rt.dev(r), out, 1)
    b = net.read_launch_trace()
    assert len(a) == net.num_launches and a == b and len(set(a)) > len(a) // 2
    names = [net.launch_name(i) for i in range(len(a))]
    assert names[0] and net.launch_name(len(a)) is None
    last = net.read_launch_output(len(a) - 1).view(np.float32).reshape(1, 1, 25, 41)
    assert np.array_equal(last, first)
    net.execute(rt.dev(l2), rt.dev(r), out, 1)               # another LEFT image: right-tower-only launches keep their hashes
    c = net.read_launch_trace()
    differs = [i for i in range(len(a)) if a[i] != c[i]]
    same = [names[i] for i in range(len(a)) if a[i] == c[i]]
    assert differs and differs[-1] == len(a) - 1
    assert all(n.startswith("right_") for n in same), same
    net.set_launch_trace(False)
    net.destroy()


@pytest.mark.gpu
def test_graph_mode_survives_a_growing_batch():
    """captured graphs bake in the addresses of the context's internal buffers, and a larger batch reallocates them --
    batch 1 (captured), batch 2, batch 1 again must not replay the stale graph"""
    lib = netlib("gpu")
    w = O.synth_weights_resnet18_2d()
    l, r = pairs(2, 65, 129)
    net = lib.create("resnet18_2D", 129, 65, max_batch=2, weights=w, max_disp=16)
    L, R = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    o1 = torch.full((1, 1, 65, 129), float("nan"), device="cuda")
    o2 = torch.full((2, 1, 65, 129), float("nan"), device="cuda")
    net.execute(L, R, o1, 1)
    torch.cuda.synchronize()
    ref1 = o1.cpu().numpy().copy()
    net.set_graph(True)
    for _ in range(3):                                        # direct, capture + launch, replay
        o1.fill_(float("nan"))
        net.execute(L, R, o1, 1)
        torch.cuda.synchronize()
        assert np.array_equal(o1.cpu().numpy(), ref1)
    for _ in range(3):
        net.execute(L, R, o2, 2)                              # buffers grow: every captured pass is dropped
        torch.cuda.synchronize()
    assert np.array_equal(o2.cpu().numpy()[:1], ref1)
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]     # whatever was freed is somebody else's now
    for _ in range(3):
        o1.fill_(float("nan"))
        net.execute(L, R, o1, 1)
        torch.cuda.synchronize()
        assert np.array_equal(o1.cpu().numpy(), ref1)
    del junk
    net.destroy()


@pytest.mark.gpu
def test_exact_engine_is_deterministic_beside_default_engines():
    """profiles/r04_race.txt: the exact-fp32 engine (interleaved Winograd launches) beside default split-fp16 engines was the configuration
    that deviated in 99.6 % of its passes while the library still contained compiler-formed packed fp32 math (v_pk_add_f32 in the
    interleaved epilogue, beside co-resident fp16-MFMA waves).  Two exact contexts and four default ones, 300 passes each, every launch
    of every pass hashed (launch trace): bit-identical to the first pass throughout."""
    lib = netlib("gpu")
    path = model_files.weight_file("resnet18_2D")
    W_, H_ = 1257, 369
    l, r = O.synth_pair(H_, W_, 1234)
    L, R = torch.from_numpy(l)[None].cuda(), torch.from_numpy(r)[None].cuda()
    flags = [capi.RT_CONV_EXACT_FP32, capi.RT_CONV_EXACT_FP32, 0, 0, 0, 0]
    nets = [lib.create("resnet18_2D", W_, H_, weights_path=path, flags=f) for f in flags]
    for n in nets:
        n.set_streams(1)
        n.set_launch_trace(True)
    streams = [torch.cuda.Stream() for _ in nets]
    outs = [torch.full((1, 1, H_, W_), float("nan"), device="cuda") for _ in nets]
    ref = None
    for it in range(300):
        for c, n in enumerate(nets):
            n.execute(L, R, outs[c], 1, stream=streams[c].cuda_stream)
        traces = [n.read_launch_trace() for n in nets]
        if ref is None:
            ref = traces
            assert traces[0] == traces[1] and traces[2] == traces[3] == traces[4] == traces[5]
            names = [nets[0].launch_name(i) for i in range(nets[0].num_launches)]
        for c in range(len(nets)):
            if traces[c] != ref[c]:
                k = next(i for i in range(len(ref[c])) if traces[c][i] != ref[c][i])
                raise AssertionError("pass %d, context %d (%s engine): launch %d (%s) differs from the first pass" % (
                    it, c, "exact" if flags[c] else "default", k, names[k] if flags[c] else nets[c].launch_name(k)))
    with torch.no_grad():
        want = O.resnet18_2d(torch.from_numpy(l)[None], torch.from_numpy(r)[None], O.read_weights(path))
    for o in outs:
        assert (o.cpu() - want).abs().max().item() <= 1e-3
    for n in nets:
        n.destroy()
