#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_16; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 600 python tools/iso_conv3d.py fold 8 2>&1 | grep -v amdgpu > $O/fold_b8.txt; cat $O/fold_b8.txt
timeout 600 python tools/iso_conv3d.py fold 1 2>&1 | grep -v amdgpu | head -4 > $O/fold_b1.txt; cat $O/fold_b1.txt
