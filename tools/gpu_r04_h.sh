#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04h; mkdir -p $O
RT_WINO_IL8=1 RACE_SHOW=2 RACE_NEIGHBOURS=split RT_VARIANT_DIR=tools/build/expA_p4 timeout 300 python tools/race_locate.py 3000 6 exact 1 > $O/locate_p4_nb_split.txt 2>&1
echo "== p4, neighbours = default engine"; grep '"mode"' $O/locate_p4_nb_split.txt | cut -c1-400
RT_WINO_IL8=1 RACE_SHOW=2 RT_VARIANT_DIR=tools/build/expA_p4 timeout 300 python tools/race_locate.py 500 6 exact 1 > $O/locate_p4_same.txt 2>&1
echo "== p4, neighbours = same"; grep '"mode"' $O/locate_p4_same.txt | cut -c1-400
RT_WINO_IL8=1 RACE_SHOW=2 RT_VARIANT_DIR=tools/build/expA_p4 timeout 300 python tools/race_locate.py 3000 1 exact 1 > $O/locate_p4_alone.txt 2>&1
echo "== p4, one context"; grep '"mode"' $O/locate_p4_alone.txt | cut -c1-400
RT_WINO_IL8=1 RACE_SHOW=2 RT_VARIANT_DIR=tools/build/expA_p4 timeout 300 python tools/race_locate.py 1500 2 exact 1 > $O/locate_p4_two.txt 2>&1
echo "== p4, two contexts"; grep '"mode"' $O/locate_p4_two.txt | cut -c1-400
(python tools/bench_3d.py nvsmall --half2; python tools/bench_3d.py nvsmall --half2 --batch=8) > $O/bench_3d.txt 2>&1; grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | head -30
