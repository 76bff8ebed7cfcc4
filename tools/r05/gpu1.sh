#!/bin/bash
# round 5, GPU call 1: LDS-DMA out-of-range semantics, depth-walk kernel parity on the GPU, A/B of the half2 3-D models with and without it
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_1; mkdir -p $O
export RT_DEV_KNOBS=1
tools/build/lds_dma_oob > $O/lds_dma_oob.txt 2>&1; cat $O/lds_dma_oob.txt
timeout 600 python -m pytest tests/test_conv3d_depth_walk.py -x -q -m gpu > $O/pytest_dw.log 2>&1; tail -n 3 $O/pytest_dw.log
for dw in 0 1; do
  RT_F16_DW=$dw timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=8 > $O/nvsmall_h2_b8_dw$dw.txt 2>&1; head -n 14 $O/nvsmall_h2_b8_dw$dw.txt
done
RT_F16_DW=1 RT_CONV_TRACE=1 timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=1 > $O/nvsmall_h2_b1_dw1.txt 2>&1; grep -v "^\[rt\]" $O/nvsmall_h2_b1_dw1.txt | head -n 14; grep "conv_f16dw" $O/nvsmall_h2_b1_dw1.txt | sort | uniq -c | head
RT_F16_DW=0 timeout 300 python tools/bench_3d.py nvsmall --half2 --batch=1 > $O/nvsmall_h2_b1_dw0.txt 2>&1; head -n 8 $O/nvsmall_h2_b1_dw0.txt
for dw in 0 1; do
  RT_F16_DW=$dw timeout 300 python tools/bench_3d.py resnet18 --half2 --batch=4 > $O/resnet18_h2_b4_dw$dw.txt 2>&1; head -n 12 $O/resnet18_h2_b4_dw$dw.txt
done
timeout 900 python -m pytest tests/test_net_parity.py -x -q -m gpu -k "nvsmall or 3d" > $O/pytest_net3d.log 2>&1; tail -n 3 $O/pytest_net3d.log
