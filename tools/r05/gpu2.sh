#!/bin/bash
# round 5, GPU call 2: the depth-walk kernel layer by layer, ablations
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_2; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 600 python -m pytest tests/test_conv3d_depth_walk.py -x -q -m gpu > $O/pytest_dw.log 2>&1; tail -n 3 $O/pytest_dw.log
timeout 900 python tools/iso_conv3d.py run 8 > $O/iso_b8.txt 2>&1; cat $O/iso_b8.txt
timeout 600 python tools/iso_conv3d.py run 1 none > $O/iso_b1.txt 2>&1; cat $O/iso_b1.txt
RT_F16_DW=1 timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 > $O/nvsmall_h2_b8_dw1.txt 2>&1; head -n 14 $O/nvsmall_h2_b8_dw1.txt
