#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04e; mkdir -p $O
RT_WINO_IL8=1 RACE_SHOW=4 RT_VARIANT_DIR=tools/build/expA_w2 timeout 300 python tools/race_locate.py 3000 6 exact 1 > $O/locate_expA_w2.txt 2>&1
echo "== expA_w2"; grep '"mode"' $O/locate_expA_w2.txt | cut -c1-500
timeout 600 python -m pytest tests/test_deconv3d_half2.py -x -q -m gpu > $O/pytest_r4.log 2>&1; tail -n 4 $O/pytest_r4.log
(python tools/bench_3d.py nvsmall --half2; python tools/bench_3d.py nvsmall --half2 --batch=8; python tools/bench_3d.py resnet18 --half2) > $O/bench_3d.txt 2>&1; grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | head -50
