#!/bin/bash
# epilogues with the activation as a compile-time constant (deconv_f16pw / f16p, conv_f16r4, deconv_s3p; the last layer's walk: one branch per step)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_22; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 900 python -m pytest tests/test_deconv3d_half2.py tests/test_net_parity.py -x -q -m gpu -k "walks_down or transpose or nvsmall or half2 or four_rows or softarg or last_deconv" 2>&1 | tail -n 3
timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 2>&1 | grep -v amdgpu > $O/layers_b8.txt; head -13 $O/layers_b8.txt
timeout 300 python tools/bench_3d.py resnet18 --batch=4 2>&1 | grep -v amdgpu > $O/layers_c4.txt; head -12 $O/layers_c4.txt
for i in 1 2; do
timeout 600 python bench.py --model nvsmall --half2 --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5', round(d['value'],1), d['unit'], round(d['roofline']['frac'],3))"
done
timeout 600 python bench.py --model resnet18 --batch 4 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4', round(d['value'],1))"
timeout 600 python bench.py --model resnet18 --half2 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet18 3D half2 b4', round(d['value'],1))"
