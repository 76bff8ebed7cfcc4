#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04o; mkdir -p $O
timeout 900 python -m pytest tests/test_deconv3d_half2.py tests/test_wino_parity.py tests/test_net_parity.py -x -q -m gpu -k "deconv3d or transpose or channel_major or four_rows or wino or exact_engine or nvsmall or 3d_models" > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log
(python tools/bench_3d.py nvsmall --half2; python tools/bench_3d.py nvsmall --half2 --batch=8; RT_NO_DECONV_P4=1 python tools/bench_3d.py nvsmall --half2 --batch=8; python tools/bench_3d.py resnet18 --half2) > $O/bench_3d.txt 2>&1; grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep -v "0.0[0-3][0-9] ms" | head -70
