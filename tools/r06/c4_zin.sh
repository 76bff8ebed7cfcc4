#!/bin/bash
# C4 (ResNet-18 3D fp32, batch 4) with the split kernel's depth slices folded into grid.x (RT_Z_INNER=2) against the default
cd ${GRAFT_REPO_ROOT:-/root/repo}
export RT_DEV_KNOBS=1
O=gpurun_out/${1:-r06_c4zin}; mkdir -p $O
for Z in 1 2 1 2; do
  RT_Z_INNER=$Z timeout 600 python bench.py --model resnet18 --batch 4 --steps 20 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RT_Z_INNER=$Z', d['value'], d['unit'], d['ms_per_step'])" | tee -a $O/run.txt
done
