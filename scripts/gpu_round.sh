#!/bin/bash
# One gpurun call that reproduces a round's evidence (round 3 flow).  usage: scripts/gpu_round.sh <tag> [tests|notests]
#   GPU test suite; bench c3 (with the CPU baseline) + kernel-trace stats; PMC of the F-solve kernels at c3 (fp32) and c5 (fp64)
#   -> profiles/fsolve_traffic.json via scripts/make_traffic_json.py; PMC of the CG tile kernel; bench c5 + trace; a rank's compute
#   share under the peer-less communicator; the host-side UBSan run (after `make -C exp-trmf-nips16_amd asan SAN=undefined`).
TAG=${1:-r03}; MODE=${2:-tests}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "$MODE" = tests ]; then
  timeout 3000 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
  grep "FUZZ-MARGIN" $O/pytest.log | sed 's/.*FUZZ-MARGIN/FUZZ-MARGIN/' > $O/fuzz_margins.txt
fi
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.json
LINES_OUT=16 bash scripts/trace_config.sh $TAG/c3 c3 > $O/trace_c3.txt 2>&1; cut -c1-165 $O/trace_c3.txt
bash scripts/pmc_fsolve.sh $TAG/pmc_c3 c3 > $O/pmc_fsolve_c3.txt 2>&1; tail -36 $O/pmc_fsolve_c3.txt
bash scripts/pmc_kernel.sh $TAG/pmc_hv "hv_tile_kernel" > $O/pmc_hv_tile.txt 2>&1; tail -24 $O/pmc_hv_tile.txt
python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 700 $O/bench_c5.json
LINES_OUT=16 bash scripts/trace_config.sh $TAG/c5 c5 --steps 6 --warmup 2 > $O/trace_c5.txt 2>&1; cut -c1-165 $O/trace_c5.txt
bash scripts/pmc_fsolve.sh $TAG/pmc_c5 c5 > $O/pmc_fsolve_c5.txt 2>&1; tail -36 $O/pmc_fsolve_c5.txt
timeout 900 python scripts/shard_compute_times.py c3 1,2,4,8 replicate,timeshard > $O/shard_compute_times.txt 2>&1; grep "^c3" $O/shard_compute_times.txt
bash scripts/hv_ablation.sh $TAG/abl 2>/dev/null | tee $O/hv_ablation.txt
[ -f exp-trmf-nips16_amd/build/asan/trmf_float32.so ] && bash scripts/asan_gpu.sh $O/ubsan.log && tail -3 $O/ubsan.log
scripts/ubench/f64_pipe > $O/f64_pipe.txt 2>&1; scripts/ubench/gridsync > $O/gridsync.txt 2>&1
