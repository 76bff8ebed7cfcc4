#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_18; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 600 python -m pytest tests/test_deconv3d_half2.py -x -q -m gpu -k "softarg or last_deconv3d" 2>&1 | tail -n 2
timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 2>&1 | grep -v amdgpu > $O/layers_b8.txt; head -13 $O/layers_b8.txt
for i in 1 2; do
timeout 600 python bench.py --model nvsmall --half2 --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5', round(d['value'],1), d['unit'], round(d['roofline']['frac'],3))"
done
