#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04n; mkdir -p $O
for d in slp_p0 slp_p7 noslp_exp; do
  echo "== build $d" >> $O/race_pair.txt
  VICTIM=il CASE="conv_s3_kernel" RT_VARIANT_DIR=tools/build/$d timeout 300 python tools/race_pair.py 4000 2>&1 | grep -v amdgpu.ids >> $O/race_pair.txt
done
cat $O/race_pair.txt
# the whole exact engine with interleaved Winograd launches, built without SLP packing: six contexts and beside five default-engine contexts
RT_WINO_IL8=1 RACE_SHOW=2 RT_VARIANT_DIR=tools/build/noslp_exp timeout 300 python tools/race_locate.py 3000 6 exact 1 > $O/locate_noslp_same.txt 2>&1; grep '"mode"' $O/locate_noslp_same.txt | cut -c1-300
RT_WINO_IL8=1 RACE_SHOW=2 RACE_NEIGHBOURS=split RT_VARIANT_DIR=tools/build/noslp_exp timeout 300 python tools/race_locate.py 3000 6 exact 1 > $O/locate_noslp_nb.txt 2>&1; grep '"mode"' $O/locate_noslp_nb.txt | cut -c1-300
# product library (no SLP): default engine
timeout 300 python tools/race_locate.py 2500 6 split 1 > $O/locate_split.txt 2>&1; grep '"mode"' $O/locate_split.txt | cut -c1-300
(python tools/bench_3d.py nvsmall --half2 --batch=8; python tools/bench_3d.py resnet18 --batch=4) > $O/bench_3d.txt 2>&1; grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | head -34
