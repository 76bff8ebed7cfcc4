#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_ctx20}; mkdir -p $O
for rep in 1 2; do
for c in 4 5 6 7 10; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --contexts $c --no-cpu-baseline --no-secondary > $O/b_$c.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/b_$c.json')); print('steps 20, contexts $c:', round(d['value'],1), 'pairs/s')"
done
done
