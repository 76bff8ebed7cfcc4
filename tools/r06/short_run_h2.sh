#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_short_h2}; mkdir -p $O
for rep in 1 2; do
for C in 6 3 4 2; do
  python bench.py --half2 --batch 8 --contexts $C --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('half2 b8 contexts $C', round(d['value'],1), d['ms_per_step'])" | tee -a $O/run.txt
done
done
