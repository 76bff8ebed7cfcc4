#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04d; mkdir -p $O
for v in expA_p5 expA_p6; do
  RT_WINO_IL8=1 RACE_SHOW=4 RT_VARIANT_DIR=tools/build/$v timeout 300 python tools/race_locate.py 3000 6 exact 1 > $O/locate_$v.txt 2>&1
  echo "== $v"; grep '"mode"' $O/locate_$v.txt | cut -c1-500
done
timeout 900 python -m pytest tests/test_deconv3d_half2.py tests/test_net_parity.py -x -q -m gpu -k "deconv3d or transpose or channel_major or 3d or nvsmall or nvtiny" > $O/pytest_3d.log 2>&1; tail -n 5 $O/pytest_3d.log
(python tools/bench_3d.py nvsmall --half2; python tools/bench_3d.py nvsmall --half2 --batch=8; python tools/bench_3d.py resnet18 --half2) > $O/bench_3d.txt 2>&1; grep -v "^      [lr]" $O/bench_3d.txt | head -70
RT_NO_DECONV_IL=1 RT_NO_SMALL_IL=1 python tools/bench_3d.py nvsmall --half2 > $O/bench_3d_old.txt 2>&1; head -8 $O/bench_3d_old.txt
