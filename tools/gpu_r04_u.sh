#!/bin/bash
export RT_DEV_KNOBS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04u; mkdir -p $O
timeout 1500 python -m pytest tests/test_deconv3d_half2.py tests/test_conv_parity.py tests/test_f16_storage.py tests/test_net_parity.py -x -q -m gpu -k "3d or conv3d or deconv or nvsmall or nvtiny or f16 or half2" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
(for nb in 1 0; do echo "== RT_NB_INNER=$nb"; RT_NB_INNER=$nb python tools/bench_3d.py nvsmall --half2; RT_NB_INNER=$nb python tools/bench_3d.py nvsmall --half2 --batch=8; RT_NB_INNER=$nb python tools/bench_3d.py nvsmall; RT_NB_INNER=$nb python tools/bench_3d.py resnet18; done) > $O/bench_3d.txt 2>&1
grep -v "^      [lr]" $O/bench_3d.txt | grep -v amdgpu.ids | grep "==\|batch\|ds \|conv3D_[4578]\|conv3D_2[ab]"
