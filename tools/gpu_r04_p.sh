#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04p; mkdir -p $O
timeout 600 python -m pytest tests/test_deconv3d_half2.py -x -q -m gpu -k "transpose" > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log
(python tools/bench_3d.py nvsmall --half2; python tools/bench_3d.py nvsmall --half2 --batch=8) > $O/bench_3d.txt 2>&1; grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep -v "0.0[0-3][0-9] ms" | head -30
