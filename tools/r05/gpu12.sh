#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_12; mkdir -p $O
timeout 900 python -m pytest tests/test_deconv3d_half2.py tests/test_f16_storage.py -x -q -m gpu 2>&1 | tail -n 1
timeout 900 python -m pytest tests/test_net_parity.py -x -q -m gpu -k "nvsmall" 2>&1 | tail -n 1
for i in 1 2; do
python bench.py --model nvsmall --half2 --batch 8 --steps 24 --warmup 4 > $O/c5_$i.json 2> /dev/null
python -c "
import json; d = json.load(open('$O/c5_$i.json')); print('C5:', round(d['value'], 1), 'pairs/s', round(d['ms_per_pair'], 4), 'ms/pair', d['config']['workload'][-30:], 'dominant', d['roofline']['kernel'], round(d['roofline']['frac'], 3))"
done
RT_DEV_KNOBS=1 timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 2>&1 | grep -v amdgpu > $O/nvsmall_h2_b8.txt; head -n 14 $O/nvsmall_h2_b8.txt
timeout 600 python tools/host_contention.py 8 300 2>&1 | grep -v amdgpu > $O/host_contention.txt; cat $O/host_contention.txt
