#!/bin/bash
# tower-block segment height (rows per workgroup) against three contexts
cd ${GRAFT_REPO_ROOT:-/root/repo}
export RT_DEV_KNOBS=1
O=gpurun_out/${1:-r06_seg}; mkdir -p $O
for rep in 1 2; do
for S in 64 48 32 96; do
  RT_RBS_SEG=$S python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('seg $S steps 20', round(d['value'],1), d['ms_per_step'], d['config'].get('contexts'))" | tee -a $O/run.txt
  RT_RBS_SEG=$S python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('seg $S steps 200', round(d['value'],1), d['ms_per_step'], d.get('latency_ms_per_pair'))" | tee -a $O/run.txt
done
done
