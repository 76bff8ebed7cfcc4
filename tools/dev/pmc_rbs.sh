#!/bin/bash
# Hardware-counter passes over the residual-block dev harness (old and new streaming kernels), one counter group per run.
#   tools/dev/pmc_rbs.sh <outdir> [harness args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/pmc_rbs}; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- $ROOT/tools/build/rbs_dev ${@:-629 185 32 4} > $OUT/g$i.log 2>&1
done <<'GRPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_ANY
SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG
GRPS
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = "new" if "rbs2" in r["Kernel_Name"] else ("old" if "rbs" in r["Kernel_Name"] else None)
        if k: acc[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
print("%-32s %14s %14s" % ("counter (mean per launch)", "old kernel", "new kernel"))
for c, d in acc.items():
    m = lambda v: sum(v) / len(v) if v else float("nan")
    print("%-32s %14.6g %14.6g" % (c, m(d["old"]), m(d["new"])))
PY
