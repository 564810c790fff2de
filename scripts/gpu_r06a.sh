#!/bin/bash
# Round 6, first GPU call: the split path of long rows -- parity tests, bit-identity of the uniform path (digests), config 3 unchanged,
# the new workloads with the split path and (TRMF_LONG_ROW=0) on the static row mapping, kernel traces of both.
TAG=${1:-r06a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/test_evidence.txt
timeout 2400 python -m pytest tests/test_gpu_split.py -x -q -s > $O/pytest_split.log 2>&1; echo "pytest exit $?" >> $O/pytest_split.log; tail -5 $O/pytest_split.log
cp gpurun_out/test_evidence.txt $O/test_evidence.txt 2>/dev/null
python scripts/digest_run.py > $O/digests.txt 2>&1; cat $O/digests.txt
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['value'],1), 'iter/s  F', round(d['phases_ms']['F'],4), 'X', round(d['phases_ms']['X'],4), 'Fkernel ms', round(d['roofline']['avg_kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'Xgram', (d.get('roofline_x') or {}).get('gram',{}).get('avg_ms'), 'cg', d['phases_ms']['cg_iter'][:6], d['config']['parallelism'][-150:])"; }
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot 2>$O/bench_c3.err | tee $O/bench_c3_$i.json | line c3; done
for cfg in imp zipf imp60; do
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot 2>$O/bench_$cfg.err | tee $O/bench_$cfg.json | line $cfg
  TRMF_TEST=1 TRMF_LONG_ROW=0 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot 2>$O/bench_${cfg}_static.err | tee $O/bench_${cfg}_static.json | line ${cfg}-static
done
for cfg in imp zipf; do
  LINES_OUT=12 bash scripts/trace_config.sh $TAG/$cfg $cfg --no-one-shot > $O/trace_$cfg.txt 2>&1; cut -c1-150 $O/trace_$cfg.txt
  TRMF_TEST=1 TRMF_LONG_ROW=0 LINES_OUT=8 bash scripts/trace_config.sh $TAG/${cfg}_static $cfg --no-one-shot > $O/trace_${cfg}_static.txt 2>&1; cut -c1-150 $O/trace_${cfg}_static.txt
done
