#!/bin/bash
# Round 6, third GPU call: the new tests first (in-process multi-device, split rows), then the whole GPU suite; the dropped closing pass
# (A/B against the round-5 build: digests must stay identical), PMC of the F-solve at c3 (-> profiles/fsolve_traffic.json), bench lines.
TAG=${1:-r06c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/test_evidence.txt
timeout 1500 python -m pytest tests/test_gpu_devices.py -x -q -s > $O/pytest_devices.log 2>&1; echo "pytest exit $?" >> $O/pytest_devices.log; tail -12 $O/pytest_devices.log
bash scripts/ab_builds.sh $R/exp-trmf-nips16_amd/build/old > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt
timeout 3300 python -m pytest tests -x -q -s -m gpu --deselect tests/test_gpu_devices.py > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -5 $O/pytest.log
cp gpurun_out/test_evidence.txt $O/test_evidence.txt 2>/dev/null
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['value'],1), 'iter/s', d['windows'], 'F', d['phases_ms']['F'], 'X', d['phases_ms']['X'], 'Fkernel ms', d['roofline']['avg_kernel_ms'], 'frac', round(d['roofline']['frac'],3), 'Xgram', (d.get('roofline_x') or {}).get('gram',{}).get('avg_ms'), 'survey', (d.get('value_survey_protocol') or {}).get('iter_per_s'))"; }
for cfg in c3 imp zipf; do
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot 2>$O/bench_$cfg.err | tee $O/bench_$cfg.json | line $cfg
done
bash scripts/pmc_fsolve.sh $TAG/pmc_c3 c3 > $O/pmc_fsolve_c3.txt 2>&1; tail -30 $O/pmc_fsolve_c3.txt
