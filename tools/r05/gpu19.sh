#!/bin/bash
# where the time of deconv_f16p_kernel goes: instrumented builds (RT_DP_ABL) on NVSmall half2 batch 8
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_19; mkdir -p $O
export RT_DEV_KNOBS=1
for v in "" 7 5 6; do
  if [ -z "$v" ]; then d=redtail_amd/lib; else d=tools/build/dp_$v; fi
  RT_LIB_DIR=$PWD/$d timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 2>&1 | grep "deconv3D_[12] " | tr '\n' ' ' | sed "s/^/ABL ${v:-0}: /"; echo
done | tee $O/dp_abl.txt
