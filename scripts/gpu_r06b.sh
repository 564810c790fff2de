#!/bin/bash
# Round 6, second GPU call: split path with the reduce kernel -- remaining parity tests, A/B of config 3 against the round-5 build
# (build/old), imp / zipf bench lines + traces.
TAG=${1:-r06b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/test_evidence.txt
timeout 2400 python -m pytest tests/test_gpu_split.py -x -q -s -k "(forced_split_every and (float64 or float32-64 or float32-40)) or not forced_split_every" > $O/pytest_split.log 2>&1; echo "pytest exit $?" >> $O/pytest_split.log; tail -5 $O/pytest_split.log
cp gpurun_out/test_evidence.txt $O/test_evidence.txt 2>/dev/null
bash scripts/ab_builds.sh $R/exp-trmf-nips16_amd/build/old > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['value'],1), 'iter/s  F', round(d['phases_ms']['F'],4), 'X', round(d['phases_ms']['X'],4), 'Fkernel ms', round(d['roofline']['avg_kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'Xgram', (d.get('roofline_x') or {}).get('gram',{}).get('avg_ms'), 'cg', d['phases_ms']['cg_iter'][:6])"; }
for cfg in imp zipf imp60; do
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot 2>$O/bench_$cfg.err | tee $O/bench_$cfg.json | line $cfg
done
for cfg in imp zipf; do
  LINES_OUT=14 bash scripts/trace_config.sh $TAG/$cfg $cfg --no-one-shot > $O/trace_$cfg.txt 2>&1; cut -c1-150 $O/trace_$cfg.txt
done
