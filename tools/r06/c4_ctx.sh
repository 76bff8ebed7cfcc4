#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_c4ctx}; mkdir -p $O
for C in 3 1 2 5; do
  RT_BENCH_3D_CONTEXTS=$C timeout 600 python bench.py --model resnet18 --batch 4 --steps 20 --warmup 2 --contexts $C 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 contexts $C', round(d['value'],1), d['ms_per_step'], d['config'])" | tee -a $O/run.txt
done
for B in 2 8; do
  timeout 600 python bench.py --model resnet18 --batch $B --steps 20 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 batch $B', round(d['value'],1), d['ms_per_step'])" | tee -a $O/run.txt
done
