#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04g; mkdir -p $O
timeout 600 python -m pytest tests/test_deconv3d_half2.py tests/test_net_parity.py -x -q -m gpu -k "deconv3d or transpose or channel_major or four_rows or 3d or nvsmall or nvtiny" > $O/pytest_3d.log 2>&1; tail -n 3 $O/pytest_3d.log
(python tools/bench_3d.py nvsmall --half2; python tools/bench_3d.py nvsmall --half2 --batch=8; RT_Z_INNER=0 python tools/bench_3d.py nvsmall --half2; python tools/bench_3d.py resnet18 --half2; python tools/bench_3d.py resnet18; RT_Z_INNER=0 python tools/bench_3d.py resnet18; python tools/bench_3d.py nvsmall) > $O/bench_3d.txt 2>&1; grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep -v "0.0[0-4][0-9] ms" | head -120
timeout 600 bash tools/pmc_3d.sh $PWD/$O/pmc_nvsmall nvsmall --half2 > $O/pmc_nvsmall.txt 2>&1
cut -c1-170 $O/pmc_nvsmall.txt | tail -22
