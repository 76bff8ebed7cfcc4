#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_10; mkdir -p $O
timeout 300 python -m pytest tests/test_deconv3d_half2.py -x -q -m gpu -k "factored" 2>&1 | tail -n 1
for c in 2 3 4; do
  RT_BENCH_3D_CONTEXTS=$c python bench.py --model nvsmall --half2 --batch 8 --steps 24 --warmup 4 --contexts $c > $O/c5_ctx$c.json 2> /dev/null
  python -c "
import json; d = json.load(open('$O/c5_ctx$c.json')); print('C5 contexts $c:', round(d['value'], 1), 'pairs/s', round(d['ms_per_pair'], 4), 'ms/pair', d['config']['workload'][-30:])"
done
RT_BENCH_3D_CONTEXTS=3 python bench.py --model resnet18 --batch 4 --steps 20 --warmup 3 --contexts 3 > $O/c4_ctx3.json 2> /dev/null
python -c "
import json; d = json.load(open('$O/c4_ctx3.json')); print('C4 contexts 3:', round(d['value'], 1), 'pairs/s', round(d['ms_per_pair'], 4))"
