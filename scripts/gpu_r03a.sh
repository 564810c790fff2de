#!/bin/bash
# round 3, first GPU pass: fp64 pipe micro-benchmark, the sharded-path tests, the parity suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout 120 scripts/ubench/f64_pipe > $O/f64_pipe.txt 2>&1; cat $O/f64_pipe.txt
timeout 2400 python -m pytest tests/test_dist.py -m gpu -x -q -s > $O/pytest_dist.log 2>&1; echo "pytest dist exit $?" >> $O/pytest_dist.log
tail -15 $O/pytest_dist.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_abi.py tests/test_python_frontend.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity exit $?" >> $O/pytest_parity.log
tail -8 $O/pytest_parity.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 900 $O/bench_c3.json
