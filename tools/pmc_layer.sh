#!/bin/bash
export RT_DEV_KNOBS=1      # the RT_* switches below are development knobs (see rt_capi.hip: dev_knobs)
# Hardware-counter passes over the dominant layer alone (tools/iso_layer.py), one counter group per run.
#   tools/pmc_layer.sh <outdir> <conv|block> [kernel-name filter] [batch] [hints] [half2]     (batch 2 hints 1 = the launch bench.py times:
#   two images, 64-row segments)
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc_layer}; KIND=${2:-conv}; FILTER=${3:-conv_s3}
BATCH=${4:-1}; HINTS=${5:-0}; HALF=${6:-0}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- python $ROOT/tools/iso_layer.py $KIND 5 $BATCH $HINTS $HALF > $OUT/g$i.log 2>&1
done <<'GRPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_ANY
GRBM_GUI_ACTIVE FETCH_SIZE
WRITE_SIZE
GRPS
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "$FILTER" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-40s %14.6g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
