#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python tools/race_pair3d.py 1500 > $O/race_pair3d.txt 2>&1
grep -v amdgpu.ids $O/race_pair3d.txt | tail -20
