#!/bin/bash
# Round 6, fifth GPU call: in-process multi-device tests (runtime cache), UBSan run, scale-day dry run (c3), PMC of the F-solve -> traffic,
# the round's bench lines and kernel traces.
TAG=${1:-r06e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_devices.py tests/test_bench_contract.py -x -q -m gpu > $O/pytest_devices.log 2>&1; echo "pytest exit $?" >> $O/pytest_devices.log; tail -4 $O/pytest_devices.log
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.json; echo
LINES_OUT=16 bash scripts/trace_config.sh $TAG/c3 c3 --no-one-shot --repeat 1 > $O/trace_c3.txt 2>&1; cut -c1-165 $O/trace_c3.txt
bash scripts/pmc_fsolve.sh $TAG/pmc_c3 c3 > $O/pmc_fsolve_c3.txt 2>&1; tail -8 $O/pmc_fsolve_c3.txt
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); w=d['windows']; print('$1', round(d['value'],1), 'iter/s median', round(w['iter_per_s_median'],1), 'spread', round(w['spread'],4), 'R', w['repeat'], 'F', d['phases_ms']['F'], 'X', d['phases_ms']['X'], 'Fk', d['roofline']['avg_kernel_ms'], 'frac', round(d['roofline']['frac'],3), 'Xgram', (d.get('roofline_x') or {}).get('gram',{}).get('avg_ms'), 'survey', round((d.get('value_survey_protocol') or {}).get('iter_per_s',0),1))"; }
for cfg in imp zipf imp60 c2 c1 c1p; do
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot 2>$O/bench_$cfg.err | tee $O/bench_$cfg.json | line $cfg
done
LINES_OUT=12 bash scripts/trace_config.sh $TAG/zipf zipf --no-one-shot --repeat 1 > $O/trace_zipf.txt 2>&1; cut -c1-165 $O/trace_zipf.txt
python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-one-shot 2>$O/bench_c5.err | tee $O/bench_c5.json | line c5
CFGS=c3 DRY=1 timeout 900 bash scripts/scale_day.sh gpurun_out/$TAG/scale_day "1 2 4" > $O/scale_day.log 2>&1; cat $O/scale_day/summary.txt
timeout 1500 bash scripts/asan_gpu.sh $O/ubsan_gpu_run.txt; tail -4 $O/ubsan_gpu_run.txt
