#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04v; mkdir -p $O
timeout 1500 python -m pytest tests/test_deconv3d_half2.py tests/test_conv_parity.py tests/test_net_parity.py -x -q -m gpu -k "fp32 or 3d or conv3d or tran or nvtiny or nvsmall" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
(for p in 0 1; do echo "== RT_NO_DECONV_P4F32=$p"; RT_NO_DECONV_P4F32=$p python tools/bench_3d.py nvsmall; RT_NO_DECONV_P4F32=$p python tools/bench_3d.py resnet18; RT_NO_DECONV_P4F32=$p python tools/bench_3d.py nvtiny; done; python tools/bench_3d.py resnet18 --batch=4) > $O/bench_3d.txt 2>&1
grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep "==\|batch\|deconv"
