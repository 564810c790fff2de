#!/bin/bash
# Round 6, seventh GPU call (final sources): the full-size split cases incl. imp60, the SPMD split case, the bench-contract GPU tests, a complete
# UBSan run, the round's config-3 bench line with the PMC traffic figure attached.
TAG=${1:-r06g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/test_evidence.txt
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_dist.py tests/test_bench_contract.py -x -q -s -m gpu -k "full_size or split or bench" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -4 $O/pytest.log
cp gpurun_out/test_evidence.txt $O/test_evidence.txt 2>/dev/null
timeout 1200 bash scripts/asan_gpu.sh $O/ubsan_gpu_run.txt; tail -3 $O/ubsan_gpu_run.txt
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 300 $O/bench_c3.json; echo
