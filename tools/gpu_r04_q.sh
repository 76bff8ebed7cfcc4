#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04q; mkdir -p $O
timeout 900 python -m pytest tests/test_deconv3d_half2.py tests/test_net_parity.py -x -q -m gpu -k "fp32 or 3d" > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log
(python tools/bench_3d.py nvsmall; RT_NO_IL8_3D_F32=1 python tools/bench_3d.py nvsmall; python tools/bench_3d.py resnet18) > $O/bench_3d.txt 2>&1
grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep -v "0.0[0-3][0-9] ms" | head -70
timeout 900 bash tools/pmc_3d.sh $PWD/$O/pmc_c5 nvsmall --half2 > $O/pmc_c5.txt 2>&1; tail -n 25 $O/pmc_c5.txt
timeout 900 bash tools/pmc_3d.sh $PWD/$O/pmc_c4 nvsmall > $O/pmc_c4.txt 2>&1; tail -n 25 $O/pmc_c4.txt
rm -rf $O/pmc_c5/g*/ $O/pmc_c4/g*/
