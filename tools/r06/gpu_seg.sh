#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_seg}; mkdir -p $O
for seg in 0 32 48 64 96; do
  RT_DEV_KNOBS=1 RT_RBS_SEG=$seg python bench.py --no-cpu-baseline > $O/bench_seg$seg.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("$O/bench_seg$seg.json")); r = d["roofline"]
print("seg $seg", round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), "avg_launch_us", round(r.get("avg_launch_us", 0), 1), "latency", d.get("latency_ms_per_pair"))
PY
done
