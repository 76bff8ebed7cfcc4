#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_4; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 900 python tools/iso_conv3d.py run 8 > $O/iso_b8.txt 2>&1; grep "conv3D_2\|conv3D_4" $O/iso_b8.txt
bash tools/r05/pmc_iso.sh $O/pmc_dw 8 0 1 > $O/pmc_dw.txt 2>&1; cat $O/pmc_dw.txt
bash tools/r05/pmc_iso.sh $O/pmc_r4 8 0 0 > $O/pmc_r4.txt 2>&1; cat $O/pmc_r4.txt
rm -rf $O/pmc_dw/g*/p_*.csv $O/pmc_r4/g*/p_*.csv
