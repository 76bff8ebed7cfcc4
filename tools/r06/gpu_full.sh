#!/bin/bash
# What the driver runs at round end, on the builder's box: the GPU test tier (with the skip reasons), smoke(), the default bench line and the driver's command
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_full}; mkdir -p $O
timeout 3000 python -m pytest tests/ -x -q -m gpu -rs > $O/pytest.log 2>&1; tail -n 50 $O/pytest.log | grep -v "^$" | tail -n 45
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
python bench.py > $O/bench_plain.json 2> $O/bench_plain.err; tail -n 2 $O/bench_plain.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> /dev/null
python bench.py --half2 --batch 8 --no-cpu-baseline > $O/bench_half2_b8.json 2> /dev/null
python bench.py --model nvsmall --half2 --batch 8 --steps 24 --warmup 3 --check > $O/bench_nvsmall_half2_b8.json 2> /dev/null
python bench.py --model resnet18 --batch 4 --steps 20 --warmup 2 --check > $O/bench_resnet18_3d_b4.json 2> /dev/null
python - <<PY
import json
for f in ("bench_half2_b8", "bench_nvsmall_half2_b8", "bench_resnet18_3d_b4"):
    d = json.load(open("$O/%s.json" % f)); print(f, round(d["value"], 1), d["unit"], "frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"].get("traffic"), d.get("pairs_bit_equal_to_batch1_engine"), d.get("parity_max_abs_err"))
for f in ("bench_plain", "bench_20_5"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"], 1), d["unit"], "frac", round(r["frac"], 4), r["bound"], "traffic", r.get("traffic"), "latency", d.get("latency_ms_per_pair"), "diff", d.get("contexts_max_abs_diff"))
    for e in (d.get("secondary") or []):
        r2 = e.get("roofline") or {}
        print("   ", (e.get("config") or {}).get("workload", e.get("id", "?"))[:72], round(e["value"], 1) if "value" in e else e.get("error"), "frac", r2.get("frac"), "traffic", r2.get("traffic"))
PY
