#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r06_trace}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $O/trace.log 2>&1
python $R/tools/concurrency.py $O/trace | tee $O/concurrency.txt
find $O/trace -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
