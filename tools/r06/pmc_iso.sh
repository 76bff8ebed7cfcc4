#!/bin/bash
# rocprofv3 counter passes over one Conv3D layer alone: tools/r05/pmc_iso.sh <outdir> <batch> <shape index> <dw 0|1>
export RT_DEV_KNOBS=1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/$1; B=$2; SH=$3; DW=$4
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- python $R/tools/iso_conv3d.py one $B $SH $DW > $OUT/g$i.log 2>&1
done <<'GRPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_LDS
FETCH_SIZE TCC_HIT_sum
WRITE_SIZE TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum
GRPS
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list); dur = []
for f in sorted(glob.glob("$OUT/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "conv_f16" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob("$OUT/g1/p_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "conv_f16" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel us under pmc: avg %.1f over %d launches" % (sum(dur) / max(1, len(dur)), len(dur)))
for k, v in acc.items():
    print("%-40s %14.5g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
