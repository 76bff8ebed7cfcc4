"""Build recipes for the native libraries (all in-tree, so the .so files travel with gpurun).

  lib/librt_stereo_hip.so        HIP kernels + C ABI (include/rt_stereo.h), hipcc, gfx950 only
  lib/libnvstereo_inference.so   host C++: NvInfer.h shim, plugins, executor, networks, net-level C ABI
  tests/emu/build/*.so           the same sources compiled for x86 against the SIMT emulator
                                 (tests only -- see tests/emu/hip/hip_runtime.h)

    python -m redtail_amd.build [hip] [host] [emu] [apps]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "redtail_amd", "csrc")
LIB = os.path.join(ROOT, "redtail_amd", "lib")
EMU = os.path.join(ROOT, "tests", "emu")
EMU_BUILD = os.path.join(EMU, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOSTCXX = os.environ.get("RT_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")
GXX = os.environ.get("CXX", "g++")

HOST_SOURCES = ["host/plugins.cpp", "host/engine.cpp", "host/networks.cpp", "host/net_capi.cpp"]
# Device code is built WITHOUT the SLP vectoriser (round 4, profiles/r04_race.txt): the packed fp32 instructions it forms from adjacent
# scalar adds / multiplies (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32) (a) are slower than their scalar forms beside MFMAs on gfx950 (+4 %
# on the headline without them) and (b) were the trigger of the one run-to-run deviation ever located in this code base -- the exact-fp32
# Winograd kernel's interleaved epilogue beside co-resident fp16-MFMA waves; with -fno-slp-vectorize 0 of 8000 launches deviate, with it
# 40 %.  Packed math that is WRITTEN as such in the kernels (f32x2 types) is unaffected.
DEVICE_FLAGS = ["-fno-slp-vectorize", "-DRT_BUILT_NO_SLP"]      # rt_capi.hip refuses to compile for the device without the pair (profiles/r04_race.txt)


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for s in sources:
        if os.path.isdir(s):
            for d, _, fs in os.walk(s):
                if any(os.path.getmtime(os.path.join(d, f)) > t for f in fs):
                    return True
        elif os.path.exists(s) and os.path.getmtime(s) > t:
            return True
    return False


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_hip(force=False):
    """Cross-compiles the kernels for gfx950 (works without a GPU)."""
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "librt_stereo_hip.so")
    deps = [os.path.join(CSRC, "rt_capi.hip"), os.path.join(CSRC, "kernels"), os.path.join(ROOT, "include")]
    if force or _newer(out, deps):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
              "-Wno-unused-function"] + DEVICE_FLAGS + [os.path.join(CSRC, "rt_capi.hip"), "-o", out])
        check_no_packed_f32(out)
    return out


def check_no_packed_f32(lib):
    """The product library must not contain compiler-formed packed fp32 math outside the allow-list (tools/check_no_packed_f32.py,
    profiles/r04_race.txt): a build that does is deleted and the build fails."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import check_no_packed_f32 as guard
    finally:
        sys.path.pop(0)
    try:
        counts, bad, kernels = guard.check(lib)
    except BaseException:
        # a guard that could not run (missing llvm-objdump, a failed unbundle) must not leave an unchecked library behind: the next
        # build would find it up to date and skip the check
        if os.path.exists(lib):
            os.remove(lib)
        raise
    print("packed-fp32 guard: %d kernels, v_pk_{add,mul,fma}_f32 in %d allow-listed kernels, %d violations" % (kernels, len(counts) - len(bad), len(bad)))
    if bad:
        os.remove(lib)
        raise RuntimeError("packed fp32 instructions outside the allow-list (tools/check_no_packed_f32.py): %s" % sorted(bad.items()))


def build_hip_timing(force=False):
    """Instrumented build of the kernel library for tools/time_phases.py (never used by the product)."""
    outdir = os.path.join(ROOT, "tools", "build")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "librt_stereo_hip_timing.so")
    deps = [os.path.join(CSRC, "rt_capi.hip"), os.path.join(CSRC, "kernels"), os.path.join(ROOT, "include")]
    if force or _newer(out, deps):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DRT_KERNEL_TIMING", "-w"] + DEVICE_FLAGS +
             [os.path.join(CSRC, "rt_capi.hip"), "-o", out])
    return out


def build_hip_ablation(mask, force=False, timing=False):
    """Kernel library with parts of conv_mfma_f32_kernel compiled out (tools/ablate_conv.py; never the product)."""
    outdir = os.path.join(ROOT, "tools", "build")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "librt_stereo_hip_abl%d%s.so" % (mask, "_timing" if timing else ""))
    deps = [os.path.join(CSRC, "rt_capi.hip"), os.path.join(CSRC, "kernels"), os.path.join(ROOT, "include")]
    if force or _newer(out, deps):
        # mask >= 10000: no ablation, conv_f16mma_kernel compiled for (mask - 10000) waves per SIMD
        defs = ["-DRT_ABLATE=0", "-DRT_F16_WAVES=%d" % (mask - 10000)] if mask >= 10000 else ["-DRT_ABLATE=%d" % mask]
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w"] + DEVICE_FLAGS + defs +
             (["-DRT_KERNEL_TIMING"] if timing else []) + [
              os.path.join(CSRC, "rt_capi.hip"), "-o", out])
    return out


def build_variant(name, defines, force=False, device_flags=None, kernels_only=False):
    """Another build of the two product libraries side by side in tools/build/<name>/ (probes: RT_EXPERIMENTAL kernel families, hazard
    probes ...; never the product).  Tools take the directory through RT_VARIANT_DIR."""
    outdir = os.path.join(ROOT, "tools", "build", name)
    os.makedirs(outdir, exist_ok=True)
    kern = os.path.join(outdir, "librt_stereo_hip.so")
    host = os.path.join(outdir, "libnvstereo_inference.so")
    deps = [os.path.join(CSRC, "rt_capi.hip"), os.path.join(CSRC, "kernels"), os.path.join(ROOT, "include")]
    if force or _newer(kern, deps):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w"] + (DEVICE_FLAGS if device_flags is None else list(device_flags)) +
             list(defines) + [os.path.join(CSRC, "rt_capi.hip"), "-o", kern])
    if not kernels_only and (force or _newer(host, [os.path.join(CSRC, "host"), os.path.join(ROOT, "include"), os.path.join(ROOT, "redtail_amd", "include"), kern])):
        _build_host_against(kern, host)
    return outdir


def build_emu(force=False):
    """Host build of the same kernel sources on top of tests/emu (test infrastructure)."""
    os.makedirs(EMU_BUILD, exist_ok=True)
    out = os.path.join(EMU_BUILD, "librt_stereo_emu.so")
    deps = [os.path.join(CSRC, "rt_capi.hip"), os.path.join(CSRC, "kernels"), os.path.join(ROOT, "include"),
            os.path.join(EMU, "hip"), os.path.join(EMU, "hip_emu.cpp")]
    if force or _newer(out, deps):
        # -DRT_EXPERIMENTAL: the CPU test tier also covers the rejected kernel families the product library leaves out (rt_capi.hip: exp_knob)
        _run([HOSTCXX, "-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-DRT_EXPERIMENTAL", "-I", EMU, "-Wall",
              "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-unused-variable", "-Wno-psabi",
              os.path.join(CSRC, "rt_capi.hip"), os.path.join(EMU, "hip_emu.cpp"), "-o", out])
    return out


def _build_host_against(kernels_lib, out):
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    libdir, libname = os.path.split(kernels_lib)
    _run([GXX, "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "redtail_amd", "include")] + srcs +
         ["-L", libdir, "-l:" + libname, "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + libdir, "-o", out])


def build_host(force=False):
    out = os.path.join(LIB, "libnvstereo_inference.so")
    kern = build_hip()
    deps = [os.path.join(CSRC, "host"), os.path.join(ROOT, "include"), os.path.join(ROOT, "redtail_amd", "include"), kern]
    if force or _newer(out, deps):
        _build_host_against(kern, out)
    return out


def build_host_emu(force=False):
    out = os.path.join(EMU_BUILD, "libnvstereo_inference_emu.so")
    kern = build_emu()
    deps = [os.path.join(CSRC, "host"), os.path.join(ROOT, "include"), os.path.join(ROOT, "redtail_amd", "include"), kern]
    if force or _newer(out, deps):
        _build_host_against(kern, out)
    return out


def build_ref_link_check(force=False):
    """Compiles the REFERENCE's generated network builders, untouched and where they lie, against our
    NvInfer.h / redtail_tensorrt_plugins.h and links them to libnvstereo_inference.so: the drop-in
    check of the API surface.  Output goes to oracle/_ref/ (git-ignored; travels to the GPU box so the
    GPU tests can run the reference-defined graphs).  Skipped when /root/reference is absent."""
    ref = os.environ.get("RT_REFERENCE", "/root/reference")
    app = os.path.join(ref, "stereoDNN", "sample_app")
    if not os.path.isdir(app):
        return None
    outdir = os.path.join(ROOT, "oracle", "_ref")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "libref_nets.so")
    nets = [os.path.join(app, f) for f in ("resnet18_2D_513x257_net.cpp", "nvtiny_513x161_net.cpp",
                                           "nvsmall_1025x321_net.cpp", "resnet18_1025x321_net.cpp")]
    glue = os.path.join(ROOT, "oracle", "ref_nets_glue.cpp")
    host = build_host()
    if force or _newer(out, nets + [glue, host]):
        libdir, libname = os.path.split(host)
        _run([GXX, "-std=c++17", "-O1", "-fPIC", "-shared", "-w", "-I", os.path.join(ROOT, "include"),
              "-I", os.path.join(ROOT, "redtail_amd", "include")] + nets + [glue] +
             ["-L", libdir, "-l:" + libname, "-Wl,-Bsymbolic", "-Wl,-rpath,$ORIGIN/../../redtail_amd/lib",
              "-Wl,-rpath," + libdir, "-o", out])
    return out


def build_sample_app(force=False, emu=False):
    """Compiles the REFERENCE's sample application -- stereoDNN/sample_app/main.cpp and its four generated network
    builders, untouched and where they lie -- against our NvInfer.h / redtail_tensorrt_plugins.h / cuda_runtime_api.h and
    the test-only OpenCV subset in tests/shim/, and links it to libnvstereo_inference.so: "sample_app links unchanged".
    Output: oracle/_ref/nvstereo_sample_app (git-ignored, travels to the GPU box); emu=True links the emulator build of
    the libraries instead (CPU tier).  Skipped when /root/reference is absent."""
    ref = os.environ.get("RT_REFERENCE", "/root/reference")
    app = os.path.join(ref, "stereoDNN", "sample_app")
    if not os.path.isdir(app):
        return None
    outdir = EMU_BUILD if emu else os.path.join(ROOT, "oracle", "_ref")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "nvstereo_sample_app_emu" if emu else "nvstereo_sample_app")
    srcs = [os.path.join(app, f) for f in ("main.cpp", "resnet18_2D_513x257_net.cpp", "nvtiny_513x161_net.cpp",
                                           "nvsmall_1025x321_net.cpp", "resnet18_1025x321_net.cpp")]
    host = build_host_emu() if emu else build_host()
    kern = build_emu() if emu else build_hip()
    shim = os.path.join(ROOT, "tests", "shim")
    if force or _newer(out, srcs + [host, shim]):
        libdir, libname = os.path.split(host)
        _run([GXX, "-std=c++17", "-O1", "-w", "-DNDEBUG", "-I", shim, "-I", os.path.join(ROOT, "include"),
              "-I", os.path.join(ROOT, "redtail_amd", "include"), "-I", app] + srcs +
             ["-L", libdir, "-l:" + libname, "-l:" + os.path.basename(kern), "-lz",
              "-Wl,-rpath,$ORIGIN/../../redtail_amd/lib", "-Wl,-rpath," + libdir, "-o", out])
    return out


def build_reference_tests(force=False, emu=False):
    """Compiles the REFERENCE's plugin test suite -- stereoDNN/tests/tests_main.cpp, 23 googletest cases that drive every
    plugin through the public C++ API (one-plugin networks, addShuffle for 4-D inputs, build -> execute -> destroy) on its
    TensorFlow-generated golden tensors -- untouched, against our headers, the test-only googletest / OpenCV subsets in
    tests/shim/ and libnvstereo_inference.so.  Output: oracle/_ref/nvstereo_tests (emu=True: the emulator build)."""
    ref = os.environ.get("RT_REFERENCE", "/root/reference")
    src = os.path.join(ref, "stereoDNN", "tests", "tests_main.cpp")
    if not os.path.exists(src):
        return None
    outdir = EMU_BUILD if emu else os.path.join(ROOT, "oracle", "_ref")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "nvstereo_tests_emu" if emu else "nvstereo_tests")
    host = build_host_emu() if emu else build_host()
    kern = build_emu() if emu else build_hip()
    shim = os.path.join(ROOT, "tests", "shim")
    if force or _newer(out, [src, host, shim]):
        libdir, libname = os.path.split(host)
        _run([GXX, "-std=c++17", "-O1", "-w", "-I", shim, "-I", os.path.join(ROOT, "include"),
              "-I", os.path.join(ROOT, "redtail_amd", "include"), src, os.path.join(shim, "gtest_glue.cpp"),
              "-L", libdir, "-l:" + libname, "-l:" + os.path.basename(kern), "-lz",
              "-Wl,-rpath,$ORIGIN/../../redtail_amd/lib", "-Wl,-rpath," + libdir, "-o", out])
    return out


def build_oracle_c(force=False):
    """oracle/corr_cpu.c -- the plain C restatement of the reference's correlation kernel (checker / CPU baseline, never the product)
    -> oracle/_ref/libcorr_cpu.so (git-ignored, travels to the GPU box)."""
    src = os.path.join(ROOT, "oracle", "corr_cpu.c")
    outdir = os.path.join(ROOT, "oracle", "_ref")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "libcorr_cpu.so")
    if force or _newer(out, [src]):
        _run(["gcc", "-O2", "-shared", "-fPIC", src, "-o", out])
    return out


def build_native_apps(force=False):
    """apps/stereo_throughput.cpp: the native multi-GPU driver (one thread per device, RCCL weight broadcast through the C ABI)
    -> redtail_amd/lib/stereo_throughput (git-ignored, travels to the GPU box)."""
    src = os.path.join(ROOT, "apps", "stereo_throughput.cpp")
    out = os.path.join(LIB, "stereo_throughput")
    host, kern = build_host(), build_hip()
    if force or _newer(out, [src, host, os.path.join(ROOT, "include")]):
        _run([GXX, "-std=c++17", "-O2", "-g", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), src,
              "-L", LIB, "-l:" + os.path.basename(host), "-l:" + os.path.basename(kern), "-Wl,-rpath,$ORIGIN", "-o", out])
    return out


def build_engine_tests(force=False, emu=False):
    """Our own C++ test program for the executor's graph passes on graphs the four models do not contain
    (tests/cpp/engine_graph_tests.cpp; driven by tests/test_engine_graphs.py).  Output: tools/build/engine_graph_tests
    (git-ignored, travels to the GPU box) or, emu=True, tests/emu/build/engine_graph_tests_emu."""
    src = os.path.join(ROOT, "tests", "cpp", "engine_graph_tests.cpp")
    outdir = EMU_BUILD if emu else os.path.join(ROOT, "tools", "build")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "engine_graph_tests_emu" if emu else "engine_graph_tests")
    host = build_host_emu() if emu else build_host()
    kern = build_emu() if emu else build_hip()
    if force or _newer(out, [src, host]):
        libdir, libname = os.path.split(host)
        _run([GXX, "-std=c++17", "-O1", "-g", "-Wall", "-I", os.path.join(ROOT, "include"),
              "-I", os.path.join(ROOT, "redtail_amd", "include"), src,
              "-L", libdir, "-l:" + libname, "-l:" + os.path.basename(kern),
              "-Wl,-rpath,$ORIGIN/../../redtail_amd/lib", "-Wl,-rpath," + libdir, "-o", out])
    return out


def main(argv):
    what = argv or ["hip", "host", "emu"]
    if "hip" in what:
        build_hip()
    if "host" in what:
        build_host()
    if "apps" in what:
        build_sample_app()
        build_reference_tests()
        build_engine_tests()
        build_native_apps()
    if "emu" in what:
        build_emu()
        if os.path.exists(os.path.join(CSRC, "host", "engine.cpp")):
            build_host_emu()


if __name__ == "__main__":
    main(sys.argv[1:])
