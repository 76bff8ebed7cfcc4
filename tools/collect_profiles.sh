#!/bin/bash
export RT_DEV_KNOBS=1      # the RT_* switches below are development knobs (see rt_capi.hip: dev_knobs)
# Everything profiles/ is made of, in one gpurun call (see profiles/README.md):
#   gpurun -- 'bash tools/collect_profiles.sh r02p' ; python tools/summarize_profiles.py gpurun_out/r02p r02
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-prof}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> /dev/null
RT_RB=0 python $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_layer_by_layer.json 2> /dev/null      # the tower blocks as two launches each
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $O/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --steps 10 --warmup 2 --contexts 1 --streams-per-context 1 --spinup-ms 0 --no-cpu-baseline --no-secondary > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 10 --warmup 2 --contexts 1 --streams-per-context 1 --spinup-ms 0 --no-cpu-baseline --no-secondary > $O/pmc_sq.log 2>&1
python $R/bench.py --half2 --batch 1 --no-cpu-baseline > $O/bench_half2_b1.json 2> /dev/null
python $R/bench.py --half2 --batch 8 --no-cpu-baseline > $O/bench_half2_b8.json 2> /dev/null
python $R/bench.py --model nvsmall --half2 --batch 8 --steps 20 --warmup 3 > $O/bench_nvsmall_half2_b8.json 2> /dev/null
python $R/bench.py --model resnet18 --batch 4 --steps 10 --warmup 2 > $O/bench_resnet18_3d_b4.json 2> /dev/null
(python $R/tools/bench_3d.py nvtiny nvsmall resnet18; python $R/tools/bench_3d.py nvsmall resnet18 --half2; python $R/tools/bench_3d.py nvsmall --half2 --batch=8; python $R/tools/bench_3d.py resnet18 --batch=4) 2>&1 | grep -v amdgpu.ids > $O/bench_3d.txt
python $R/tools/layer_profile.py 2>&1 | grep -v amdgpu.ids > $O/layers.txt
python $R/tools/race_hunt.py 2>&1 | grep -v amdgpu.ids > $O/race.txt
python -c "import sys; sys.path.insert(0, '$R'); from redtail_amd import build; build.build_hip_timing()" > /dev/null 2>&1    # instrumented library of the phase tools (rebuilt only when stale)
bash $R/tools/pmc_layer.sh $O/pmc_block block conv_s3rbs > $O/pmc_layer_resblock.txt 2>&1
bash $R/tools/pmc_layer.sh $O/pmc_conv conv conv_s3_kernel > $O/pmc_layer_conv_s3.txt 2>&1
(RT_TIME_BLOCK=1 python $R/tools/time_phases_split.py 1; python $R/tools/time_phases_split.py 1; python $R/tools/host_overhead.py) 2>&1 | grep -v amdgpu.ids > $O/phases.txt
(python $R/tools/time_tail.py; RT_S3_KSPLIT=0 python $R/tools/time_tail.py | grep "per launch") 2>&1 | grep -v amdgpu.ids > $O/tail.txt
rocprofv3 --kernel-trace --output-format csv -d $O/sync_trace -o t -- python $R/tools/sync_trace.py run > /dev/null 2>&1
python $R/tools/sync_trace.py show $O/sync_trace > $O/sync_timeline.txt 2>&1
python $R/tools/micro/cold_code_probe.py 2>&1 | grep -v amdgpu.ids > $O/cold_code.txt
ls $O $O/trace | head -40
