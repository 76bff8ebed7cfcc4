#!/bin/bash
export RT_DEV_KNOBS=1      # the RT_* switches below are development knobs (see rt_capi.hip: dev_knobs)
# Round 6's profiles/ in one gpurun call (see profiles/README.md):
#   gpurun -- 'bash tools/collect_r06.sh r06z' ; python tools/summarize_profiles.py gpurun_out/r06z r06 ;
#   python tools/traffic_3d_json.py profiles/r06_traffic_3d.json "nvsmall half2=gpurun_out/r06z/pmc_c5/summary.json" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-prof}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
env -u RT_DEV_KNOBS python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
env -u RT_DEV_KNOBS python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $O/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --steps 10 --warmup 2 --contexts 1 --streams-per-context 1 --spinup-ms 0 --no-cpu-baseline --no-secondary > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 10 --warmup 2 --contexts 1 --streams-per-context 1 --spinup-ms 0 --no-cpu-baseline --no-secondary > $O/pmc_sq.log 2>&1
env -u RT_DEV_KNOBS python $R/bench.py --half2 --batch 8 --no-cpu-baseline > $O/bench_half2_b8.json 2> /dev/null
env -u RT_DEV_KNOBS python $R/bench.py --model nvsmall --half2 --batch 8 --steps 24 --warmup 3 --check > $O/bench_nvsmall_half2_b8.json 2> /dev/null
env -u RT_DEV_KNOBS python $R/bench.py --model resnet18 --batch 4 --steps 20 --warmup 2 --check > $O/bench_resnet18_3d_b4.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -o p -- python $R/bench.py --half2 --batch 8 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1
cp $O/trace_c3/p_kernel_stats.csv $O/kernel_stats_half2_b8.csv; rm -rf $O/trace_c3
# rocprofv3 kernel stats of the C5 / C4 bench commands
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -o p -- python $R/bench.py --model nvsmall --half2 --batch 8 --steps 12 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -o p -- python $R/bench.py --model resnet18 --batch 4 --steps 10 --warmup 2 > /dev/null 2>&1
cp $O/trace_c5/p_kernel_stats.csv $O/kernel_stats_nvsmall_half2_b8.csv; cp $O/trace_c4/p_kernel_stats.csv $O/kernel_stats_resnet18_3d_b4.csv
rm -rf $O/trace_c5 $O/trace_c4
(python $R/tools/bench_3d.py nvtiny nvsmall resnet18; python $R/tools/bench_3d.py nvsmall resnet18 --half2; python $R/tools/bench_3d.py nvsmall --half2 --batch=8; python $R/tools/bench_3d.py resnet18 --batch=4) 2>&1 | grep -v amdgpu.ids > $O/bench_3d.txt
# per-launch counters: the 3-D models (batch 1), NVTiny, and the 2-D model at the sizes of the secondary bench lines
bash $R/tools/pmc_3d.sh $O/pmc_c5 nvsmall --half2 > $O/pmc_c5.txt 2>&1
bash $R/tools/pmc_3d.sh $O/pmc_c4s nvsmall > $O/pmc_c4s.txt 2>&1
bash $R/tools/pmc_3d.sh $O/pmc_c4 resnet18 > $O/pmc_c4.txt 2>&1
bash $R/tools/pmc_3d.sh $O/pmc_c1 nvtiny > $O/pmc_c1.txt 2>&1
bash $R/tools/pmc_3d.sh $O/pmc_c3 resnet18_2D --half2 --batch=8 > $O/pmc_c3.txt 2>&1
bash $R/tools/pmc_3d.sh $O/pmc_513 resnet18_2D_513 > $O/pmc_513.txt 2>&1
rm -rf $O/pmc_c5/g*/ $O/pmc_c4/g*/ $O/pmc_c4s/g*/ $O/pmc_c1/g*/ $O/pmc_c3/g*/ $O/pmc_513/g*/
python $R/tools/bench_ops.py --json $O/ops.json 2>&1 | grep -v amdgpu.ids > $O/ops.txt
bash $R/tools/pmc_layer.sh $O/pmc_block block conv_s3rbd 2 1 > $O/pmc_layer_resblock.txt 2>&1
bash $R/tools/pmc_layer.sh $O/pmc_block_h block conv_f16rbd 16 1 1 > $O/pmc_layer_resblock_half2.txt 2>&1
rm -rf $O/pmc_block $O/pmc_block_h
rocprofv3 --kernel-trace --output-format csv -d $O/sync_trace -o t -- python $R/tools/sync_trace.py run > /dev/null 2>&1
python $R/tools/sync_trace.py show $O/sync_trace > $O/sync_timeline.txt 2>&1
rm -rf $O/sync_trace
ls $O $O/trace | head -60
