#!/bin/bash
# after the integration of the pre-split tower blocks: parity tests of the block and the 2-D nets, then the bench lines (with / without)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_step1}; mkdir -p $O
timeout 1500 python -m pytest tests/test_split_parity.py tests/test_net_parity.py -x -q -m gpu -k "resblock or resnet18_2d or 2d" > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
python bench.py > $O/bench_plain.json 2> $O/bench_plain.err; tail -n 2 $O/bench_plain.err
RT_DEV_KNOBS=1 RT_NO_RBD=1 python bench.py > $O/bench_nordb.json 2> /dev/null
python bench.py > $O/bench_plain2.json 2> /dev/null
python - <<PY
import json
for f in ("bench_plain", "bench_nordb", "bench_plain2"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"], 1), d["unit"], "frac", round(r["frac"], 4), "avg_launch_us", r.get("avg_launch_us"), "iso", r.get("isolated_launch_us"), "latency", d.get("latency_ms_per_pair"), "parity", d.get("parity_max_abs_err"))
PY
