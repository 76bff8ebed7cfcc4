#!/bin/bash
export RT_DEV_KNOBS=1      # the RT_* switches below are development knobs (see rt_capi.hip: dev_knobs)
# Round 4's profiles/ in one gpurun call (see profiles/README.md):
#   gpurun -- 'bash tools/collect_r04.sh r04z' ; python tools/summarize_profiles.py gpurun_out/r04z r04 ;
#   python tools/traffic_3d_json.py profiles/r04_traffic_3d.json "nvsmall half2=gpurun_out/r04z/pmc_c5/summary.json" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-prof}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $O/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --steps 10 --warmup 2 --contexts 1 --streams-per-context 1 --spinup-ms 0 --no-cpu-baseline --no-secondary > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 10 --warmup 2 --contexts 1 --streams-per-context 1 --spinup-ms 0 --no-cpu-baseline --no-secondary > $O/pmc_sq.log 2>&1
python $R/bench.py --half2 --batch 8 --no-cpu-baseline > $O/bench_half2_b8.json 2> /dev/null
python $R/bench.py --model nvsmall --half2 --batch 8 --steps 24 --warmup 3 --check > $O/bench_nvsmall_half2_b8.json 2> /dev/null
python $R/bench.py --model resnet18 --batch 4 --steps 20 --warmup 2 --check > $O/bench_resnet18_3d_b4.json 2> /dev/null
(python $R/tools/bench_3d.py nvtiny nvsmall resnet18; python $R/tools/bench_3d.py nvsmall resnet18 --half2; python $R/tools/bench_3d.py nvsmall --half2 --batch=8; python $R/tools/bench_3d.py resnet18 --batch=4) 2>&1 | grep -v amdgpu.ids > $O/bench_3d.txt
bash $R/tools/pmc_3d.sh $O/pmc_c5 nvsmall --half2 > $O/pmc_c5.txt 2>&1
bash $R/tools/pmc_3d.sh $O/pmc_c4s nvsmall > $O/pmc_c4s.txt 2>&1
bash $R/tools/pmc_3d.sh $O/pmc_c4 resnet18 > $O/pmc_c4.txt 2>&1
rm -rf $O/pmc_c5/g*/ $O/pmc_c4/g*/ $O/pmc_c4s/g*/
python $R/tools/bench_ops.py --json $O/ops.json 2>&1 | grep -v amdgpu.ids > $O/ops.txt
bash $R/tools/pmc_layer.sh $O/pmc_block block conv_s3rbs 2 1 > $O/pmc_layer_resblock.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/sync_trace -o t -- python $R/tools/sync_trace.py run > /dev/null 2>&1
python $R/tools/sync_trace.py show $O/sync_trace > $O/sync_timeline.txt 2>&1
(VICTIM=il CASE=conv_s3_kernel python $R/tools/race_pair.py 2000) 2>&1 | grep -v amdgpu.ids > $O/race_pair.txt
ls $O $O/trace | head -40
