#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04l; mkdir -p $O
for d in expA_p4 expA_p4_noslp; do
  echo "== build $d" >> $O/race_pair_noslp.txt
  VICTIM=il CASE="conv_s3_kernel" RT_VARIANT_DIR=tools/build/$d timeout 300 python tools/race_pair.py 4000 2>&1 | grep -v amdgpu.ids >> $O/race_pair_noslp.txt
  VICTIM=il_out CASE="conv_s3_kernel" RT_VARIANT_DIR=tools/build/$d timeout 300 python tools/race_pair.py 4000 2>&1 | grep -v amdgpu.ids >> $O/race_pair_noslp.txt
done
cat $O/race_pair_noslp.txt
