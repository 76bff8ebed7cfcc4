#!/bin/bash
# conv_s3rbd_kernel development run on the GPU box: tools/r06/rbd_dev.sh <outdir> "<defines of variant 1>" "<defines of variant 2>" ...
# builds tools/dev/rbd_harness.hip per variant (plain + -DRT_KERNEL_TIMING) there and runs check (first variant) / time / phases
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_rbd}; shift; mkdir -p $O tools/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DRT_BUILT_NO_SLP -w"
[ $# -eq 0 ] && set -- ""
i=0
for V in "$@"; do
  i=$((i+1))
  echo "=== variant $i: [$V]" | tee -a $O/run.txt
  /opt/rocm/bin/hipcc $F $V tools/dev/rbd_harness.hip -o tools/build/rbd_harness_$i || continue
  /opt/rocm/bin/hipcc $F $V -DRT_KERNEL_TIMING tools/dev/rbd_harness.hip -o tools/build/rbd_harness_t$i || continue
  M="time"; [ $i -eq 1 ] && M="check time"
  timeout 600 tools/build/rbd_harness_$i $M 2>&1 | tee -a $O/run.txt
  timeout 300 tools/build/rbd_harness_t$i phases 2>&1 | grep -v '^rbs\|^   ' > /dev/null
  timeout 300 tools/build/rbd_harness_t$i phases 2>&1 | tee -a $O/run.txt
done
