#!/usr/bin/env python3
"""Guard of the cure for the round-3 / round-4 nondeterminism (profiles/r04_race.txt): compiler-formed packed fp32 math (v_pk_add_f32 /
v_pk_mul_f32 / v_pk_fma_f32, made by the SLP vectoriser of -O3 out of scalar epilogue arithmetic) computed wrong values in a wave that
shared a CU with fp16-MFMA waves; the library is built with -fno-slp-vectorize since.  This disassembles the gfx950 code object of the
built library and fails when any kernel outside the allow-list contains such an instruction -- a new kernel, a changed default or a new
compiler cannot bring them back silently.  Called by redtail_amd/build.py after every build of librt_stereo_hip.so.
    python tools/check_no_packed_f32.py [library]
Allow-list (kernels whose packed fp32 is WRITTEN as such in the source, f32x2 / f32x4 vector arithmetic):
  conv_wino_f32_kernel   the Winograd input transform (never part of a deviation: tools/race_pair.py, victims and aggressors)
  ew_f32_kernel / ew_f16_kernel   the stand-alone element-wise plugins (ELU, add + activation on 16-byte vectors)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ALLOW = ("conv_wino_f32_kernel", "ew_f32_kernel", "ew_f16_kernel")
PACKED = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")


def packed_f32_by_kernel(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "copy.so")], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co], check=True)
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
    counts, kernels, name = collections.Counter(), 0, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            name = m.group(1)
            kernels += 1
        elif name and PACKED.search(line):
            counts[name] += 1
    return counts, kernels


def check(lib):
    counts, kernels = packed_f32_by_kernel(lib)
    assert kernels > 50, "disassembly of %s found only %d kernels" % (lib, kernels)
    bad = {k: v for k, v in counts.items() if not any(a in k for a in ALLOW)}
    return counts, bad, kernels


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "redtail_amd", "lib", "librt_stereo_hip.so")
    counts, bad, kernels = check(lib)
    print("%s: %d kernels, packed fp32 instructions in %d of them (allow-listed: %d)" % (lib, kernels, len(counts), len(counts) - len(bad)))
    for k, v in sorted(bad.items()):
        print("  NOT ALLOWED: %4d x v_pk_{add,mul,fma}_f32 in %s" % (v, k))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
