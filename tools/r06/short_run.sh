#!/bin/bash
# the driver's command (20 timed steps after 5 warm-up steps) against the number of contexts: fill / drain of a 7 ms timed region
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_short}; mkdir -p $O
for rep in 1 2 3 4; do
for C in 6 3 2 7; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --contexts $C --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps 20 contexts $C', round(d['value'],1), d['ms_per_step'], d.get('latency_ms_per_pair'))" | tee -a $O/run.txt
done
done
for rep in 1 2; do
for C in 6 3 2; do
  python bench.py --contexts $C --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps 200 contexts $C', round(d['value'],1), d['ms_per_step'])" | tee -a $O/run.txt
done
done
