#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_5; mkdir -p $O
export RT_DEV_KNOBS=1
timeout 1500 python -m pytest tests/test_deconv3d_half2.py tests/test_conv3d_depth_walk.py tests/test_determinism.py tests/test_imgproc_parity.py tests/test_oracle_golden.py -x -q -m gpu -rs > $O/pytest_ops.log 2>&1; tail -n 5 $O/pytest_ops.log
timeout 1500 python -m pytest tests/test_net_parity.py -x -q -m gpu -rs -k "nvsmall or 3d" > $O/pytest_net3d.log 2>&1; tail -n 5 $O/pytest_net3d.log
timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 > $O/nvsmall_h2_b8.txt 2>&1; head -n 14 $O/nvsmall_h2_b8.txt
timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=1 > $O/nvsmall_h2_b1.txt 2>&1; head -n 8 $O/nvsmall_h2_b1.txt
timeout 300 python tools/bench_3d.py resnet18 --batch=4 > $O/resnet18_b4.txt 2>&1; head -n 14 $O/resnet18_b4.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c5 -o p -- python $GRAFT_REPO_ROOT/tools/iso_3d.py nvsmall 4 --half2 --batch=8 > $GRAFT_REPO_ROOT/$O/prof_c5.log 2>&1
cd $GRAFT_REPO_ROOT; head -n 16 $O/prof_c5/p_kernel_stats.csv | cut -c1-150; rm -f $O/prof_c5/p_kernel_trace.csv $O/prof_c5/*.db
