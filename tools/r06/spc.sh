#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_spc}; mkdir -p $O
for rep in 1 2; do
for CS in "3 1" "3 2" "2 2" "5 2"; do
  set -- $CS
  python bench.py --contexts $1 --streams-per-context $2 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contexts $1 streams $2 steps 200', round(d['value'],1), d['ms_per_step'])" | tee -a $O/run.txt
  python bench.py --steps 20 --warmup 5 --contexts $1 --streams-per-context $2 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contexts $1 streams $2 steps 20', round(d['value'],1), d['ms_per_step'])" | tee -a $O/run.txt
done
done
