#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_ctx}; mkdir -p $O
for cfg in "4 1" "6 1" "8 1" "10 1" "3 2" "4 2"; do
  set -- $cfg
  python bench.py --contexts $1 --streams-per-context $2 --no-cpu-baseline --no-secondary > $O/b_$1_$2.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/b_$1_$2.json')); print('contexts $1 streams $2:', round(d['value'],1), 'pairs/s')"
done
for cfg in "2 2" "4 2" "6 2" "4 1" "6 1" "8 1"; do
  set -- $cfg
  python bench.py --half2 --batch 8 --contexts $1 --streams-per-context $2 --no-cpu-baseline --no-secondary > $O/h_$1_$2.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/h_$1_$2.json')); print('half2 b8 contexts $1 streams $2:', round(d['value'],1), 'pairs/s')"
done
