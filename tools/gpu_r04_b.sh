#!/bin/bash
# round 4, GPU call B: probes of the interleaved-Winograd deviation (conv_wino.hip.h: RT_WINO_PROBE), 3000 passes x 6 contexts each
export RT_DEV_KNOBS=1 RT_WINO_IL8=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04b; mkdir -p $O
for v in expA expA_p1 expA_p2 expA_p3 expA_p4; do
  RACE_SHOW=40 RT_VARIANT_DIR=tools/build/$v timeout 300 python tools/race_locate.py 3000 6 exact 1 > $O/locate_$v.txt 2>&1
  echo "== $v"; grep '"mode"' $O/locate_$v.txt | cut -c1-700
done
RACE_SHOW=60 RT_VARIANT_DIR=tools/build/expB timeout 300 python tools/race_locate.py 600 6 exact 1 > $O/locate_expB.txt 2>&1
grep '"mode"' $O/locate_expB.txt | cut -c1-700
