#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_9; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 600 python -m pytest tests/test_deconv3d_half2.py -x -q -m gpu -k "factored" > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
timeout 600 python tools/iso_conv3d.py fold 8 2>&1 | grep -v amdgpu > $O/fold_b8.txt; cat $O/fold_b8.txt
timeout 600 python tools/iso_conv3d.py fold 1 2>&1 | grep -v amdgpu > $O/fold_b1.txt; cat $O/fold_b1.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/tools/iso_conv3d.py fold 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; head -n 12 $O/prof/p_kernel_stats.csv | cut -c1-150; rm -f $O/prof/p_kernel_trace.csv $O/prof/*.db
