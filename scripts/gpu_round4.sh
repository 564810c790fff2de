#!/bin/bash
# One gpurun call that reproduces round 4's evidence.  usage: scripts/gpu_round4.sh <tag> [tests|notests]
#   GPU test suite (+ gpurun_out/test_evidence.txt); bench c3 with the CPU baseline; kernel-trace stats of c3 (persistent kernel) and of the
#   launch-per-step path; PMC of the F-solve kernels at c3 / c5 -> profiles/fsolve_traffic.json; PMC of the persistent CG kernel;
#   bench + trace of c5; bench lines of c2 / c1 / c1p; the phase timeline of the persistent kernel (build/prof)
TAG=${1:-r04}; MODE=${2:-tests}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/test_evidence.txt
if [ "$MODE" = tests ]; then
  timeout 3300 python -m pytest tests -x -q -s -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
  grep -o "FUZZ-MARGIN.*" $O/pytest.log > $O/fuzz_margins.txt
  cp gpurun_out/test_evidence.txt $O/test_evidence.txt 2>/dev/null
fi
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 700 $O/bench_c3.json
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['phases_ms']['F'], d['phases_ms']['X'], d['roofline']['frac'])"; done > $O/bench_c3_repeats.txt; cat $O/bench_c3_repeats.txt
TRMF_PERSIST=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3_launch_per_step.json 2>/dev/null
LINES_OUT=14 bash scripts/trace_config.sh $TAG/c3 c3 > $O/trace_c3.txt 2>&1; cut -c1-165 $O/trace_c3.txt
TRMF_PERSIST=0 LINES_OUT=14 bash scripts/trace_config.sh $TAG/c3_launch c3 > $O/trace_c3_launch.txt 2>&1
bash scripts/pmc_fsolve.sh $TAG/pmc_c3 c3 > $O/pmc_fsolve_c3.txt 2>&1; tail -30 $O/pmc_fsolve_c3.txt
bash scripts/pmc_kernel.sh $TAG/pmc_persist "cg_persist" > $O/pmc_persist.txt 2>&1; tail -26 $O/pmc_persist.txt
python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 600 $O/bench_c5.json
LINES_OUT=14 bash scripts/trace_config.sh $TAG/c5 c5 --steps 6 --warmup 2 > $O/trace_c5.txt 2>&1; cut -c1-165 $O/trace_c5.txt
bash scripts/pmc_fsolve.sh $TAG/pmc_c5 c5 > $O/pmc_fsolve_c5.txt 2>&1; tail -30 $O/pmc_fsolve_c5.txt
bash scripts/bench_small_configs.sh > $O/small_configs.txt 2>&1; cat $O/small_configs.txt
for cfg in c2 c1 c1p; do python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$cfg.json 2>/dev/null; done
if [ -f exp-trmf-nips16_amd/build/prof/trmf_float32.so ]; then
  TRMF_CORELIB_DIR=$R/exp-trmf-nips16_amd/build/prof TRMF_PERSIST_PROF=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/persist_prof.err
  grep PERSIST_PROF $O/persist_prof.err | tail -72 | head -36 > $O/persist_timeline.txt; grep PERSIST_TILES $O/persist_prof.err | tail -1 > $O/persist_tiles.txt
fi
