#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_13; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 900 python tools/iso_conv3d.py run 8 2>&1 | grep -v amdgpu > $O/iso_b8.txt; cat $O/iso_b8.txt
timeout 900 python tools/iso_conv3d.py run 1 2>&1 | grep -v amdgpu > $O/iso_b1.txt; cat $O/iso_b1.txt
