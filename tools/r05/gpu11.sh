#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_11; mkdir -p $O
export RT_DEV_KNOBS=1
for z in 1 2; do
  RT_Z_INNER=$z timeout 300 python tools/bench_3d.py resnet18 --batch=4 2>&1 | grep -v amdgpu > $O/resnet18_b4_z$z.txt; echo "RT_Z_INNER=$z"; head -n 7 $O/resnet18_b4_z$z.txt
done
unset RT_DEV_KNOBS
for c in 5 6 7 8; do
  python bench.py --no-secondary --no-cpu-baseline --contexts $c --steps 200 --warmup 20 > $O/c2_ctx$c.json 2> /dev/null
  python -c "
import json; d = json.load(open('$O/c2_ctx$c.json')); print('C2 contexts $c:', round(d['value'], 1), 'pairs/s')"
done
