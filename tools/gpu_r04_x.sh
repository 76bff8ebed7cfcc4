#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04x; mkdir -p $O
timeout 1500 python -m pytest tests/test_deconv3d_half2.py tests/test_conv_parity.py tests/test_f16_storage.py tests/test_net_parity.py -x -q -m gpu -k "fp32 or 3d or conv3d or tran or deconv or nvtiny or nvsmall or half2 or f16" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
(for z in 0 1; do echo "== RT_NO_SMALL_IL_F32=$z"; RT_NO_SMALL_IL_F32=$z python tools/bench_3d.py nvsmall; RT_NO_SMALL_IL_F32=$z python tools/bench_3d.py resnet18; done; python tools/bench_3d.py nvtiny;  python tools/bench_3d.py resnet18 --batch=4; python tools/bench_3d.py nvsmall --half2 --batch=8) > $O/bench_3d.txt 2>&1
grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep "==\|batch\|deconv"
