#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_7; mkdir -p $O
python tools/micro/hbm_rates.py > $O/hbm_rates.txt 2>&1; cat $O/hbm_rates.txt
