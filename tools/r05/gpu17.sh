#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_17; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 600 python -m pytest tests/test_ops_parity.py tests/test_reference_plugin_tests.py -x -q -m gpu 2>&1 | tail -n 3
timeout 900 python tools/bench_ops.py > $O/ops.txt 2> $O/ops.err; grep -i "cost volume\|transform\|pad D\|slice D\|softarg" $O/ops.txt
RT_NO_CV_X4=1 timeout 900 python tools/bench_ops.py 2>/dev/null | grep -i "default cost volume"
