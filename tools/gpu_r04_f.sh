#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04f; mkdir -p $O
timeout 900 bash tools/pmc_3d.sh $PWD/$O/pmc_nvsmall nvsmall --half2 > $O/pmc_nvsmall.txt 2>&1
cat $O/pmc_nvsmall.txt | cut -c1-200
