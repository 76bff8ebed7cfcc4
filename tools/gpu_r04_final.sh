#!/bin/bash
# What the driver runs at round end, on the builder's box: the GPU test tier, smoke(), the default bench line and the driver's bench command.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04final; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
python bench.py > $O/bench_plain.json 2> $O/bench_plain.err; tail -n 2 $O/bench_plain.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> /dev/null
python bench.py --model nvsmall --half2 --batch 8 --steps 24 --warmup 3 --check > $O/bench_nvsmall_half2_b8.json 2> /dev/null
python bench.py --model resnet18 --batch 4 --steps 20 --warmup 2 --check > $O/bench_resnet18_3d_b4.json 2> /dev/null
python - <<PY
import json
for f in ("bench_plain", "bench_20_5", "bench_nvsmall_half2_b8", "bench_resnet18_3d_b4"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"], 1), d["unit"], "frac", round(r["frac"], 4), r["bound"], "traffic", r.get("traffic"), "latency", d.get("latency_ms_per_pair"), "diff", d.get("contexts_max_abs_diff"))
PY
