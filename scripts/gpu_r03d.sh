#!/bin/bash
# round 3: peer-to-peer exchange of the time-sharded CG (processes sharing the GPU) + fp64 F-solve with the VALU trailing update
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_dist.py -m gpu -x -q -s -k "time_sharded or two_ranks" > $O/pytest_dist.log 2>&1; echo "pytest dist exit $?" >> $O/pytest_dist.log
grep -E "passed|failed|Error|error|timed out" $O/pytest_dist.log | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity exit $?" >> $O/pytest_parity.log
tail -4 $O/pytest_parity.log
timeout 300 python scripts/bench_c5_scaled.py 200000 5000 > $O/c5s.log 2>&1; tail -1 $O/c5s.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "config5" > $O/pytest_c5.log 2>&1; echo "pytest c5 exit $?" >> $O/pytest_c5.log
grep -E "c5 full|passed|failed" $O/pytest_c5.log
timeout 1500 python -m pytest tests/test_dist.py -m gpu -x -q -s -k "config4_full" > $O/pytest_c4.log 2>&1; echo "pytest c4 exit $?" >> $O/pytest_c4.log
grep -E "config 4 full|passed|failed|rror" $O/pytest_c4.log | tail -12
