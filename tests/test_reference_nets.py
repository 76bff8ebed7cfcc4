"""Drop-in check of the API surface: the REFERENCE's generated network builders
(/root/reference/stereoDNN/sample_app/*_net.cpp), compiled untouched against our NvInfer.h +
redtail_tensorrt_plugins.h into oracle/_ref/libref_nets.so (redtail_amd/build.py:build_ref_link_check),
must produce exactly the graph our programmatic builders (include/networks.h) produce: same launches,
bit-identical output.  Skipped where oracle/_ref was never built (it needs /root/reference at build time)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from redtail_amd import build, capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_nets.so")


def load_ref():
    if os.path.isdir("/root/reference"):
        build.build_ref_link_check()
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref_nets.so not built (no /root/reference at build time)")
    lib = ctypes.CDLL(REF_SO)
    lib.ref_net_create.restype = ctypes.c_void_p
    lib.ref_net_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    lib.ref_net_execute.argtypes = [ctypes.c_void_p] * 4
    lib.ref_net_num_launches.argtypes = [ctypes.c_void_p]
    lib.ref_net_destroy.argtypes = [ctypes.c_void_p]
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("model,name,w,h", [(0, "resnet18_2D", 513, 257), (2, "nvtiny", 513, 161), (1, "nvsmall", 1025, 321),
                                            (3, "resnet18", 1025, 321)])
def test_reference_generated_graph_matches_ours(model, name, w, h):
    """all four generated builders of the reference (ref_nets_glue.cpp:60-63) at the sizes they were generated for"""
    netlib = capi.NetLib()                   # loads our libraries first (RTLD_GLOBAL)
    ref = load_ref()
    weights = (synth.synth_weights_resnet18_2d() if model == 0 else
               synth.synth_weights_3d({1: synth.NVSMALL_3D, 2: synth.NVTINY_3D, 3: synth.RESNET18_3D}[model]))
    blob = capi.pack_weights(weights)
    l, r = synth.synth_pair(h, w)
    L, R = torch.from_numpy(l)[None].cuda(), torch.from_numpy(r)[None].cuda()
    ours = torch.full((1, 1, h, w), float("nan"), device="cuda")
    theirs = torch.full((1, 1, h, w), float("nan"), device="cuda")
    net = netlib.create(name, w, h, weights=blob)
    net.execute(L, R, ours, 1)
    hnd = ref.ref_net_create(model, w, h, blob, len(blob))
    assert hnd, "the reference-generated builder failed on our API implementation"
    assert ref.ref_net_execute(hnd, L.data_ptr(), R.data_ptr(), theirs.data_ptr()) == 0
    torch.cuda.synchronize()
    assert ref.ref_net_num_launches(hnd) == net.num_launches
    assert torch.equal(ours, theirs)
    ref.ref_net_destroy(hnd)
    net.destroy()


def test_reference_generated_resnet18_2d_on_emulator():
    """CPU tier: the reference's createResNet18_2D_513x257Network (it is resolution-agnostic) on a tiny image"""
    ref = load_ref_emu()
    netlib = capi.NetLib(build.build_host_emu(), build.build_emu())
    w, h = 33, 17
    blob = capi.pack_weights(synth.synth_weights_resnet18_2d())
    l, r = synth.synth_pair(h, w)
    L, R = l[None].copy(), r[None].copy()
    ours, theirs = np.full((1, 1, h, w), np.nan, np.float32), np.full((1, 1, h, w), np.nan, np.float32)
    net = netlib.create("resnet18_2D", w, h, weights=blob)        # default D = 48, like the generated file
    net.execute(L, R, ours, 1)
    hnd = ref.ref_net_create(0, w, h, blob, len(blob))
    assert hnd
    assert ref.ref_net_execute(hnd, L.ctypes.data, R.ctypes.data, theirs.ctypes.data) == 0
    assert ref.ref_net_num_launches(hnd) == net.num_launches
    assert np.array_equal(ours, theirs) and not np.isnan(ours).any()
    ref.ref_net_destroy(hnd)
    net.destroy()


def test_reference_generated_resnet18_2d_fp16_on_emulator():
    """the way sample_app/main.cpp builds ResNet-18 2D for `fp16`: kHALF weights, plugins created for kHALF, half2 mode.
    The executor fuses every such plugin away and stores the tensors as it does for our own builder with an fp16 weight
    file: same launches, same bits (round 1 rejected the kHALF plugins)"""
    ref = load_ref_emu()
    ref.ref_net_create_half.restype = ctypes.c_void_p
    ref.ref_net_create_half.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    netlib = capi.NetLib(build.build_host_emu(), build.build_emu())
    w, h = 33, 17
    blob = capi.pack_weights(synth.synth_weights_resnet18_2d(), fp16=True)
    l, r = synth.synth_pair(h, w)
    L, R = l[None].copy(), r[None].copy()
    ours, theirs = np.full((1, 1, h, w), np.nan, np.float32), np.full((1, 1, h, w), np.nan, np.float32)
    net = netlib.create("resnet18_2D", w, h, weights=blob, fp16_weights=True)
    net.execute(L, R, ours, 1)
    hnd = ref.ref_net_create_half(0, w, h, blob, len(blob))
    assert hnd, "the reference-generated builder failed for DataType::kHALF"
    assert ref.ref_net_execute(hnd, L.ctypes.data, R.ctypes.data, theirs.ctypes.data) == 0
    assert ref.ref_net_num_launches(hnd) == net.num_launches
    assert np.array_equal(ours, theirs) and not np.isnan(ours).any()
    ref.ref_net_destroy(hnd)
    net.destroy()


def load_ref_emu():
    """same reference sources, linked against the emulator build of our libraries"""
    app = "/root/reference/stereoDNN/sample_app"
    if not os.path.isdir(app):
        pytest.skip("needs /root/reference (CPU container only)")
    import subprocess
    host = build.build_host_emu()
    out = os.path.join(build.EMU_BUILD, "libref_nets_emu.so")
    srcs = [os.path.join(app, "resnet18_2D_513x257_net.cpp"), os.path.join(app, "nvtiny_513x161_net.cpp"),
            os.path.join(app, "nvsmall_1025x321_net.cpp"), os.path.join(app, "resnet18_1025x321_net.cpp"),
            os.path.join(ROOT, "oracle", "ref_nets_glue.cpp")]
    if build._newer(out, srcs + [host]):
        libdir, libname = os.path.split(host)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-w", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "redtail_amd", "include")] + srcs +
                              ["-L", libdir, "-l:" + libname, "-Wl,-Bsymbolic", "-Wl,-rpath," + libdir, "-o", out])
    lib = ctypes.CDLL(out)
    lib.ref_net_create.restype = ctypes.c_void_p
    lib.ref_net_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    lib.ref_net_execute.argtypes = [ctypes.c_void_p] * 4
    lib.ref_net_num_launches.argtypes = [ctypes.c_void_p]
    lib.ref_net_destroy.argtypes = [ctypes.c_void_p]
    return lib
