#!/bin/bash
# round 3: fp64 pipe probe + the new fp64 F-solve against the parity tests, both kernels timed at config 5 (scaled and full)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout 120 scripts/ubench/f64_pipe > $O/f64_pipe.txt 2>&1; tail -8 $O/f64_pipe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity exit $?" >> $O/pytest_parity.log
tail -8 $O/pytest_parity.log
for M in mfma grid; do
  TRMF_FSOLVE=$M timeout 300 python scripts/bench_c5_scaled.py 200000 10000 > $O/c5s_$M.log 2>&1; tail -2 $O/c5s_$M.log
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "config5" > $O/pytest_c5.log 2>&1; echo "pytest c5 exit $?" >> $O/pytest_c5.log
tail -5 $O/pytest_c5.log
