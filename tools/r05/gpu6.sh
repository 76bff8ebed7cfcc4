#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_6; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 600 python tools/iso_conv3d.py fold 8 > $O/fold_b8.txt 2>&1; cat $O/fold_b8.txt
timeout 600 python tools/iso_conv3d.py fold 1 > $O/fold_b1.txt 2>&1; cat $O/fold_b1.txt
unset RT_DEV_KNOBS
timeout 900 python bench.py --no-secondary > $O/bench_c2.json 2> $O/bench_c2.err; tail -n 3 $O/bench_c2.err; python - <<PY
import json
d = json.load(open("$O/bench_c2.json")); r = d["roofline"]
print("C2", round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), "frac", round(r["frac"], 4), "latency", d.get("latency_ms_per_pair"), "diff", d.get("contexts_max_abs_diff"), "parity", d.get("parity_max_abs_err"))
PY
