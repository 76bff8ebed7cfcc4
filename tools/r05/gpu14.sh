#!/bin/bash
# the soft-argmin of the 3-D models inside the last transposed layer: operator + network parity, then C5 with and without it, batch 1 as well
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_14; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 1200 python -m pytest tests/test_deconv3d_half2.py tests/test_net_parity.py -x -q -m gpu -k "softarg or nvsmall or last_deconv3d" 2>&1 | tail -n 6
for fuse in 1 0; do
  for b in 8 1; do
    RT_SOFTARG_FUSE=$fuse timeout 600 python bench.py --model nvsmall --half2 --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/c5_b${b}_fuse$fuse.json 2> $O/c5_b${b}_fuse$fuse.err
    python - <<PY
import json
d = json.load(open("$O/c5_b${b}_fuse$fuse.json")); print("nvsmall half2 batch $b fuse $fuse:", round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3), "roofline", d["roofline"]["kernel"] if "kernel" in d["roofline"] else "", round(d["roofline"]["frac"], 3))
PY
  done
done
RT_SOFTARG_FUSE=1 timeout 600 python bench.py --model resnet18 --half2 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet18 3D half2 b4 fused', round(d['value'],1))"
RT_SOFTARG_FUSE=0 timeout 600 python bench.py --model resnet18 --half2 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('resnet18 3D half2 b4 plain', round(d['value'],1))"
