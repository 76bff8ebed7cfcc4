#!/bin/bash
# the combining pass of the factored first Conv3D with its T window in LDS; the last layer's workgroups in XCD order
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_15; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 900 python -m pytest tests/test_deconv3d_half2.py tests/test_split_parity.py tests/test_net_parity.py -x -q -m gpu -k "softarg or last_deconv3d or factored or fold or nvsmall" 2>&1 | tail -n 4
timeout 600 python tools/iso_conv3d.py fold 8 2>&1 | grep -v amdgpu | grep "product\|memory" > $O/fold_b8.txt; cat $O/fold_b8.txt
for x in 1 0; do
RT_CONV_XCD=$x timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 2>&1 | grep "deconv3D_3\|pairs/s" | sed "s/^/xcd order $x: /"
done
for i in 1 2; do
timeout 600 python bench.py --model nvsmall --half2 --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5', round(d['value'],1), d['unit'], round(d['roofline']['frac'],3))"
done
timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 2>&1 | grep -v amdgpu > $O/layers_b8.txt; head -12 $O/layers_b8.txt
