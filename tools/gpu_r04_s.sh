#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04s; mkdir -p $O
timeout 1500 python -m pytest tests/test_split_parity.py tests/test_net_parity.py tests/test_engine_graphs.py tests/test_ops_parity.py -x -q -m gpu > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
python tools/bench_ops.py --only "corr+softargmax" > $O/ops_corr.txt 2>&1; grep -v amdgpu.ids $O/ops_corr.txt
for i in 1 2; do
python bench.py --no-secondary --no-cpu-baseline > $O/bench$i.json 2> $O/bench$i.err; python -c "
import json; d=json.load(open('$O/bench$i.json')); print('IL concat', d['value'], d.get('latency_ms_per_pair'), d['roofline']['frac'], d.get('contexts_max_abs_diff'), d['config'].get('launches_per_pair'))"
RT_NO_IL_CONCAT=1 python bench.py --no-secondary --no-cpu-baseline > $O/bench_old$i.json 2> $O/bench_old$i.err; python -c "
import json; d=json.load(open('$O/bench_old$i.json')); print('planar concat', d['value'], d.get('latency_ms_per_pair'), d['roofline']['frac'], d.get('contexts_max_abs_diff'), d['config'].get('launches_per_pair'))"
done
tail -n 3 $O/bench1.err
