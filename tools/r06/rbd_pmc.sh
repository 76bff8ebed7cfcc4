#!/bin/bash
# rocprofv3 counter passes over the harness' timing run: tools/r06/rbd_pmc.sh <outdir> "<defines>"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/$1; V="$2"; mkdir -p $OUT tools/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DRT_BUILT_NO_SLP -w $V tools/dev/rbd_harness.hip -o tools/build/rbd_harness_p || exit 1
cd /tmp; export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- $R/tools/build/rbd_harness_p time > $OUT/g$i.log 2>&1
done <<'GRPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS
SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
FETCH_SIZE
WRITE_SIZE
GRPS
python3 - <<PY
import csv, glob, collections
for kern in ("conv_s3rbd_kernel<true>", "conv_s3rbd_kernel<false>", "conv_s3rbs"):
    acc = collections.defaultdict(list); dur = []
    for f in sorted(glob.glob("$OUT/g*/p_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Grid_Size"] in ("64512", "32256"):      # seg 64: 63 x 2 workgroups of 512
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", kern, "(seg 64 launches)")
    for k, v in acc.items():
        print("%-40s %14.5g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
