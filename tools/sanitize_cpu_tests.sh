#!/bin/bash
export RT_DEV_KNOBS=1      # the RT_* switches below are development knobs (see rt_capi.hip: dev_knobs)
# CPU test tier under sanitizers (test infrastructure; run from the repo root, no GPU needed):
#   1. the HIP kernel sources on the SIMT emulator, compiled with UB checks in trap mode (signed overflow, shifts,
#      bounds, null, division by zero, bool/enum loads): any hit kills the test run with SIGILL;
#   2. the host library (plugins, executor, network builders, C ABI) under AddressSanitizer + UBSan.
# The regular builds are restored at the end.
set -e
B=tests/emu/build
python -c "from redtail_amd import build; build.build_emu(); build.build_host_emu()"
cp $B/librt_stereo_emu.so /tmp/rt_emu.bak; cp $B/libnvstereo_inference_emu.so /tmp/rt_host_emu.bak
restore() { cp /tmp/rt_emu.bak $B/librt_stereo_emu.so; cp /tmp/rt_host_emu.bak $B/libnvstereo_inference_emu.so; touch $B/*.so; }
trap restore EXIT

/opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O1 -g -fPIC -shared -I tests/emu -w \
    -fsanitize=signed-integer-overflow,shift,bounds,null,integer-divide-by-zero,bool,enum,return,unreachable \
    -fsanitize-trap=all redtail_amd/csrc/rt_capi.hip tests/emu/hip_emu.cpp -o $B/librt_stereo_emu.so
touch $B/librt_stereo_emu.so $B/libnvstereo_inference_emu.so
python -m pytest tests -x -q -m "not gpu"

cp /tmp/rt_emu.bak $B/librt_stereo_emu.so
SRCS=$(python -c "from redtail_amd import build; import os; print(' '.join(os.path.join(build.CSRC, s) for s in build.HOST_SOURCES))")
g++ -std=c++17 -O1 -g -fPIC -shared -w -fsanitize=address,undefined -fno-omit-frame-pointer -I include -I redtail_amd/include \
    $SRCS -L $B -l:librt_stereo_emu.so -Wl,-rpath,'$ORIGIN' -o $B/libnvstereo_inference_emu.so
touch $B/*.so
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" \
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_net_parity.py tests/test_reference_nets.py tests/test_capi_symbols.py -x -q -m "not gpu"
echo "sanitizer runs clean"
