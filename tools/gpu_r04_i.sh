#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04i; mkdir -p $O
RT_VARIANT_DIR=tools/build/expA_p4 timeout 600 python tools/race_pair.py 3000 > $O/race_pair_p4.txt 2>&1
grep -v amdgpu.ids $O/race_pair_p4.txt
