#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04r; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I redtail_amd/csrc/kernels -I redtail_amd/csrc -I include tools/micro/split_mix.hip -o /tmp/split_mix 2> $O/split_build.log && /tmp/split_mix | tee $O/split_mix.txt
timeout 900 python -m pytest tests/test_ops_parity.py tests/test_split_parity.py tests/test_conv_parity.py tests/test_pitch_parity.py -x -q -m gpu > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log
python tools/bench_ops.py --only corr > $O/ops_corr.txt 2>&1; grep -v amdgpu.ids $O/ops_corr.txt
python tools/bench_ops.py --only "conv3x3 32->32" > $O/ops_conv.txt 2>&1; grep -v amdgpu.ids $O/ops_conv.txt
(python tools/bench_3d.py nvsmall; RT_Z_INNER=2 python tools/bench_3d.py nvsmall) > $O/bench_3d.txt 2>&1
grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep -v "0.0[0-3][0-9] ms" | head -30
python bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d.get('latency_ms_per_pair'), d['roofline']['frac'], d.get('contexts_max_abs_diff'))"
