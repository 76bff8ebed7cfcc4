#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04y; mkdir -p $O
for v in 0 1 unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  python bench.py --no-secondary --no-cpu-baseline > $O/bench_$v.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('HIP_FORCE_DEV_KERNARG=$v', round(d['value'],1), 'latency', round(d['latency_ms_per_pair'],4), 'single_ctx', {k: (round(x,1) if isinstance(x,float) else x) for k,x in d.get('single_context',{}).items() if k in ('value','synchronous_execute_ms')})"
done
