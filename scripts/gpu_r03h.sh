#!/bin/bash
# round 3: sanitizer run, full GPU suite, final benches + traces
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash scripts/asan_gpu.sh $O/ubsan.log; tail -4 $O/ubsan.log
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; r=json.load(open('$O/bench_c3.json')); print(r['value'], r['phases_ms'], r['roofline']['frac'], r['roofline']['traffic'], r.get('cpu_baseline',{}).get('value'))"
LINES_OUT=14 bash scripts/trace_config.sh r03h/c3 c3 > $O/trace_c3.txt 2>&1; cat $O/trace_c3.txt | cut -c1-165
python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; python -c "
import json; r=json.load(open('$O/bench_c5.json')); print(r['value'], r['phases_ms'], r['roofline']['frac'], r['roofline']['traffic'])"
for C in c2 c1p c1; do python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$C.json 2> $O/bench_$C.err; python -c "
import json; r=json.load(open('$O/bench_$C.json')); print('$C', r['value'], r['phases_ms'])"; done
