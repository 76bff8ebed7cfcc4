#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04w; mkdir -p $O
timeout 1500 python -m pytest tests/test_deconv3d_half2.py tests/test_conv_parity.py tests/test_f16_storage.py -x -q -m gpu > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
(for z in 1 0; do echo "== RT_SMALL_Z_INNER=$z"; RT_SMALL_Z_INNER=$z python tools/bench_3d.py nvsmall; RT_SMALL_Z_INNER=$z python tools/bench_3d.py resnet18; RT_SMALL_Z_INNER=$z python tools/bench_3d.py nvtiny; done; python tools/bench_3d.py resnet18 --batch=4) > $O/bench_3d.txt 2>&1
grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep "==\|batch\|deconv"
