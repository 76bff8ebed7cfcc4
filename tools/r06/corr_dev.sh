#!/bin/bash
# planar correlation development run on the GPU box: tools/r06/corr_dev.sh <outdir>
cd ${GRAFT_REPO_ROOT:-/root/repo}
export RT_DEV_KNOBS=1
O=gpurun_out/${1:-r06_corr}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_parity.py -x -q -m gpu -k "corr" 2>&1 | tail -15 | tee $O/pytest.txt
timeout 600 python tools/bench_ops.py --only corr --iters 200 2>&1 | tee $O/ops.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_reference_plugin_tests.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_full.txt
bash tools/r06/corr_harness.sh ${1:-r06_corr}
