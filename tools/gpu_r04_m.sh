#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04m; mkdir -p $O
./tools/build/pk_probe > $O/pk_probe.txt 2>&1; cat $O/pk_probe.txt
python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err; python -c "import json;d=json.load(open('$O/bench_default.json'));print('default', d['value'], d['ms_per_step'], d.get('latency_ms_per_pair'), d['contexts_max_abs_diff'])"
RT_LIB_DIR=$PWD/tools/build/noslp python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_noslp.json 2> $O/bench_noslp.err; python -c "import json;d=json.load(open('$O/bench_noslp.json'));print('noslp', d['value'], d['ms_per_step'], d.get('latency_ms_per_pair'), d['contexts_max_abs_diff'])"
python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_default2.json 2> /dev/null; python -c "import json;d=json.load(open('$O/bench_default2.json'));print('default', d['value'], d['ms_per_step'], d.get('latency_ms_per_pair'))"
RT_LIB_DIR=$PWD/tools/build/noslp python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_noslp2.json 2> /dev/null; python -c "import json;d=json.load(open('$O/bench_noslp2.json'));print('noslp', d['value'], d['ms_per_step'], d.get('latency_ms_per_pair'))"
tail -3 $O/bench_default.err
