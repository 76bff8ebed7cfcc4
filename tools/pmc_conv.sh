#!/bin/bash
export RT_DEV_KNOBS=1      # the RT_* switches below are development knobs (see rt_capi.hip: dev_knobs)
# Hardware-counter passes over the dominant convolution alone (tools/bench_ops.py, one counter group per run).
#   tools/pmc_conv.sh <outdir> [bench_ops --only filter]
OUT=${1:-/root/repo/gpurun_out/pmc_conv}; FILTER=${2:-"+res+ELU b8"}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- python /root/repo/tools/bench_ops.py --iters 5 --only "$FILTER" > $OUT/g$i.log 2>&1
done <<'GRPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum
GRPS
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "conv_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-40s %14.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
