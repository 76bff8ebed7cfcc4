#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04t; mkdir -p $O
timeout 1500 python -m pytest tests/test_split_parity.py tests/test_deconv3d_half2.py tests/test_ops_parity.py tests/test_net_parity.py tests/test_engine_graphs.py -x -q -m gpu > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
python tools/bench_ops.py --only "corr" > $O/ops_corr.txt 2>&1; grep -v amdgpu.ids $O/ops_corr.txt | grep -v CPU
(python tools/bench_3d.py nvsmall; RT_NO_IL_FEAT_F32=1 python tools/bench_3d.py nvsmall; python tools/bench_3d.py resnet18; python tools/bench_3d.py resnet18 --batch=4) > $O/bench_3d.txt 2>&1
grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep -v "0.[01][0-9][0-9] ms" | head -40
