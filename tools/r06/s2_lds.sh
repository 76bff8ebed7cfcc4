#!/bin/bash
# stride-2 fp16-operand convolution, parity-split LDS columns against the round-5 layout: same box, per-launch times of NVSmall half2 (batch 8).
# The variant libraries (tools/build/s2old = the layout that was kept, s2pad, default = split) were built with redtail_amd.build.build_variant
# from a working tree that had the split layout behind a macro; the change was not kept (profiles/r06_s2_lds.txt), so this script is a record.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_s2}; mkdir -p $O
for rep in 1 2; do
for V in "" tools/build/s2old tools/build/s2pad; do
  echo "== lib [$V]" | tee -a $O/run.txt
  RT_LIB_DIR=$V python tools/bench_3d.py nvsmall --half2 --batch=8 2>&1 | grep -E "pairs/s|3ds|6ds" | tee -a $O/run.txt
done
done
