#!/bin/bash
# round-6 baseline on this round's box: default bench line, stand-alone op table, resblock alone
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_base}; mkdir -p $O
python bench.py > $O/bench_plain.json 2> $O/bench_plain.err; tail -n 2 $O/bench_plain.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> /dev/null
RT_DEV_KNOBS=1 python tools/bench_ops.py > $O/ops.txt 2>&1; tail -n 40 $O/ops.txt
(for a in "block 40 2 0" "block 40 2 1" "block 40 1 0" "conv 40 2 0"; do RT_DEV_KNOBS=1 python tools/iso_layer.py $a; done) > $O/iso_layer.txt 2>&1; tail -n 20 $O/iso_layer.txt
python - <<PY
import json
for f in ("bench_plain", "bench_20_5"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"], 1), d["unit"], "frac", round(r["frac"], 4), r["bound"], "traffic", r.get("traffic"), "latency", d.get("latency_ms_per_pair"))
PY
