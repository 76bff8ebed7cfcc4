#!/bin/bash
# the bench lines of tools/r06/gpu_full.sh without the test tier (box-to-box spread of the final tree)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_lines}; mkdir -p $O
python bench.py > $O/bench_plain.json 2> $O/bench_plain.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> /dev/null
python bench.py --half2 --batch 8 --no-cpu-baseline > $O/bench_half2_b8.json 2> /dev/null
python bench.py --model nvsmall --half2 --batch 8 --steps 24 --warmup 3 --check > $O/bench_nvsmall_half2_b8.json 2> /dev/null
python bench.py --model resnet18 --batch 4 --steps 20 --warmup 2 --check > $O/bench_resnet18_3d_b4.json 2> /dev/null
python - <<PY
import json
for f in ("bench_plain", "bench_20_5", "bench_half2_b8", "bench_nvsmall_half2_b8", "bench_resnet18_3d_b4"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["value"], 1), d["unit"], "frac", round(r["frac"], 4), "traffic", r.get("traffic"), "latency", d.get("latency_ms_per_pair"))
PY
