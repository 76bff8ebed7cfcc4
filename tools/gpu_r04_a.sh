#!/bin/bash
# round 4, GPU call A: where does the exact-fp32 / interleaved-Winograd deviation happen (tools/race_locate.py on RT_EXPERIMENTAL builds),
# the op-level table, and the SQ / traffic counters of the tower block at the shape bench.py times (two images, 64-row segments)
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04a; mkdir -p $O
timeout 400 python -m pytest tests/test_net_parity.py -q -m gpu -k "graph_mode or launch_trace" > $O/pytest_new.log 2>&1
RT_WINO_IL8=1 RT_VARIANT_DIR=tools/build/expB timeout 400 python tools/race_locate.py 1500 6 exact 1 > $O/locate_expB.txt 2>&1
RT_WINO_IL8=1 RT_VARIANT_DIR=tools/build/expB RACE_NO_TRACE=1 timeout 300 python tools/race_locate.py 1000 6 exact 1 > $O/locate_expB_notrace.txt 2>&1
RT_WINO_IL8=1 RT_VARIANT_DIR=tools/build/expA timeout 400 python tools/race_locate.py 2500 6 exact 1 > $O/locate_expA.txt 2>&1
timeout 300 python tools/race_locate.py 2500 6 split 1 > $O/locate_split.txt 2>&1
timeout 400 python tools/bench_ops.py --json $O/ops.json > $O/ops.txt 2>&1
timeout 600 bash tools/pmc_layer.sh $PWD/$O/pmc_block block conv_s3rbs 2 1 > $O/pmc_block.txt 2>&1
tail -n 5 $O/pytest_new.log $O/locate_*.txt | cut -c1-600
