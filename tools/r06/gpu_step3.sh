#!/bin/bash
# half2 block alone (segment lengths), then C3
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_step3}; mkdir -p $O
(for seg in 0 16 32 64 96 188; do echo "RT_RBS_SEG=$seg"; RT_DEV_KNOBS=1 RT_RBS_SEG=$seg python tools/iso_layer.py block 40 16 1 1; RT_DEV_KNOBS=1 RT_RBS_SEG=$seg python tools/iso_layer.py block 40 2 1 1; done; python tools/iso_layer.py conv 40 16 0 1) 2>&1 | grep -v amdgpu.ids | tee $O/iso.txt
timeout 600 python -m pytest tests/test_f16_storage.py -x -q -m gpu -k "resblock_f16" 2>&1 | tail -n 2
python bench.py --half2 --batch 8 --no-cpu-baseline > $O/bench_half2_b8.json 2> /dev/null
python - <<PY
import json
d = json.load(open("$O/bench_half2_b8.json")); r = d["roofline"]
print(round(d["value"], 1), d["unit"], "ms/step", d["ms_per_step"], "frac", round(r["frac"], 4), r["bound"], "avg_launch_us", r.get("avg_launch_us"), "iso", r.get("isolated_launch_us"))
PY
