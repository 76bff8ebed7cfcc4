#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04k; mkdir -p $O
for v in planar il_in il_out il_nores mfma32 split; do
  VICTIM=$v CASE="conv_s3_kernel" RT_VARIANT_DIR=tools/build/expA_p4 timeout 300 python tools/race_pair.py 3000 2>&1 | grep -v amdgpu.ids >> $O/race_pair_victims.txt
done
cat $O/race_pair_victims.txt
