#!/bin/bash
# deconv_f16pw_kernel with both depth classes in one walk: parity on the GPU, per-layer times, C5
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_21; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 900 python -m pytest tests/test_deconv3d_half2.py tests/test_net_parity.py tests/test_determinism.py -x -q -m gpu -k "walks_down or transpose_f16 or nvsmall or half2 or determin" 2>&1 | tail -n 4
for c in 1 0; do
RT_F16P_CLASSES=$c timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 2>&1 | grep "deconv3D_[12] \|pairs/s" | tr '\n' ' ' | sed "s/^/classes in one walk $c: /"; echo
done | tee $O/walk_b8.txt
for w in 2 3 4 6; do
RT_F16P_WALK=$w timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 2>&1 | grep "deconv3D_[12] \|pairs/s" | tr '\n' ' ' | sed "s/^/two classes, $w segments: /"; echo
done | tee -a $O/walk_b8.txt
RT_F16P_CLASSES=1 timeout 300 python tools/bench_3d.py nvsmall --half2 2>&1 | grep "deconv3D_[12] \|pairs/s" | tr '\n' ' ' | sed "s/^/batch 1, one walk: /"; echo
for i in 1 2; do
timeout 600 python bench.py --model nvsmall --half2 --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5', round(d['value'],1), d['unit'], round(d['roofline']['frac'],3))"
done
RT_F16P_CLASSES=0 timeout 600 python bench.py --model nvsmall --half2 --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5, a launch per class', round(d['value'],1))"
timeout 600 python bench.py --model resnet18 --half2 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet18 3D half2 b4', round(d['value'],1))"
