#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_8; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 600 python -m pytest tests/test_deconv3d_half2.py -x -q -m gpu -k "factored or last_deconv" > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
timeout 600 python tools/iso_conv3d.py fold 8 > $O/fold_b8.txt 2>&1; head -1 $O/fold_b8.txt
timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 > $O/nvsmall_h2_b8.txt 2>&1; head -n 14 $O/nvsmall_h2_b8.txt
timeout 300 python tools/bench_3d.py resnet18 --batch=4 > $O/resnet18_b4.txt 2>&1; head -n 9 $O/resnet18_b4.txt
