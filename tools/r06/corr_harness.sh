#!/bin/bash
# tools/r06/corr_harness.sh <outdir> "<defines of variant 1>" "<defines of variant 2>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_corrh}; shift; mkdir -p $O tools/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DRT_BUILT_NO_SLP -w"
[ $# -eq 0 ] && set -- ""
i=0
for V in "$@"; do
  i=$((i+1))
  echo "=== variant $i: [$V]" | tee -a $O/run.txt
  /opt/rocm/bin/hipcc $F $V tools/dev/corr_harness.hip -o tools/build/corr_harness_$i || continue
  timeout 300 tools/build/corr_harness_$i 2>&1 | tee -a $O/run.txt
done
