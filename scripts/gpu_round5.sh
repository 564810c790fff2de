#!/bin/bash
# One gpurun call that reproduces round 5's evidence.  usage: scripts/gpu_round5.sh <tag> [tests|notests]
#   GPU test suite (+ gpurun_out/test_evidence.txt); bench c3 with the CPU baselines, one_shot and roofline_x; kernel-trace stats of c3
#   (rocprofv3 --kernel-trace --stats); PMC of the F-solve kernels at c3 / c5 -> profiles/fsolve_traffic.json; PMC of the persistent CG
#   kernel; bench + trace of c5; bench lines of c2 / c1 / c1p; the one-shot split; the per-rank cost of the sharded persistent CG.
TAG=${1:-r05}; MODE=${2:-tests}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/test_evidence.txt
if [ "$MODE" = tests ]; then
  timeout 3300 python -m pytest tests -x -q -s -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
  grep -o "FUZZ-MARGIN.*" $O/pytest.log > $O/fuzz_margins.txt
  cp gpurun_out/test_evidence.txt $O/test_evidence.txt 2>/dev/null
fi
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 400 $O/bench_c3.json
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['phases_ms']['F'], d['phases_ms']['X'], d['roofline']['frac'], d['roofline_x']['gram']['frac'], d['roofline_x']['cg']['us_per_pass'])"; done > $O/bench_c3_repeats.txt; cat $O/bench_c3_repeats.txt
LINES_OUT=14 bash scripts/trace_config.sh $TAG/c3 c3 --no-one-shot > $O/trace_c3.txt 2>&1; cut -c1-165 $O/trace_c3.txt
bash scripts/pmc_fsolve.sh $TAG/pmc_c3 c3 > $O/pmc_fsolve_c3.txt 2>&1; tail -30 $O/pmc_fsolve_c3.txt
bash scripts/pmc_kernel.sh $TAG/pmc_persist "cg_persist" > $O/pmc_persist.txt 2>&1; tail -26 $O/pmc_persist.txt
python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --one-shot-iters 2 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 600 $O/bench_c5.json
LINES_OUT=14 bash scripts/trace_config.sh $TAG/c5 c5 --steps 6 --warmup 2 --no-one-shot > $O/trace_c5.txt 2>&1; cut -c1-165 $O/trace_c5.txt
bash scripts/pmc_fsolve.sh $TAG/pmc_c5 c5 > $O/pmc_fsolve_c5.txt 2>&1; tail -30 $O/pmc_fsolve_c5.txt
for cfg in c2 c1 c1p; do python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$cfg.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_$cfg.json')); print('$cfg', round(d['value'],1), 'iter/s', d['phases_ms']['F'], d['phases_ms']['X'], (d.get('one_shot') or {}).get('steady'))"; done
python scripts/oneshot_profile.py c3 10 5 > $O/oneshot_c3.txt 2>&1
python scripts/oneshot_profile.py c2 10 4 > $O/oneshot_c2.txt 2>&1
timeout 600 python scripts/shard_persist_solo.py c3 1,2,4,8 > $O/shard_persist_solo.txt 2>&1; grep -v "^\[" $O/shard_persist_solo.txt | cut -c1-230
