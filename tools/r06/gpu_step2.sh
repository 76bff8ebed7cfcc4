#!/bin/bash
# half2 tower block in one launch: parity tests, then C3 with / without it
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_step2}; mkdir -p $O
timeout 1500 python -m pytest tests/test_f16_storage.py tests/test_net_parity.py tests/test_split_parity.py -x -q -m gpu -k "resblock or resnet18_2d or 2d" > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
python bench.py --half2 --batch 8 --no-cpu-baseline > $O/bench_half2_b8.json 2> $O/bench_half2.err; tail -n 2 $O/bench_half2.err
RT_DEV_KNOBS=1 RT_NO_RBH=1 python bench.py --half2 --batch 8 --no-cpu-baseline > $O/bench_half2_b8_norbh.json 2> /dev/null
python - <<PY
import json
for f in ("bench_half2_b8", "bench_half2_b8_norbh"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"], 1), d["unit"], "ms/step", d["ms_per_step"], "frac", round(r["frac"], 4), r["bound"], "avg_launch_us", r.get("avg_launch_us"), "iso", r.get("isolated_launch_us"), "parity", d.get("parity_max_abs_err"), d.get("config"))
PY
