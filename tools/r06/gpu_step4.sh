#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r06_step4}; mkdir -p $O
RT_DEV_KNOBS=1 python tools/bench_ops.py 2>&1 | grep -v amdgpu.ids > $O/ops.txt; grep -i 'corr\|default cost' $O/ops.txt
timeout 1200 python -m pytest tests/test_kitti.py tests/test_ops_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -s 2>&1 | grep 'D1-all\|passed\|failed\|Error' | tail -5
