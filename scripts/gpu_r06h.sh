#!/bin/bash
# Round 6, eighth GPU call: the cases of the final sources not yet run on the GPU (imp60 at full size, the SPMD split case, the bench-contract tests)
TAG=${1:-r06h}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/test_evidence.txt
timeout 900 python -m pytest tests/test_gpu_split.py -x -q -s -m gpu -k "full_size" > $O/pytest_a.log 2>&1; echo "pytest exit $?" >> $O/pytest_a.log; tail -3 $O/pytest_a.log
timeout 600 python -m pytest tests/test_dist.py -x -q -s -m gpu -k "c4-split" > $O/pytest_b.log 2>&1; echo "pytest exit $?" >> $O/pytest_b.log; tail -3 $O/pytest_b.log
timeout 900 python -m pytest tests/test_bench_contract.py tests/test_gpu_devices.py -x -q -m gpu > $O/pytest_c.log 2>&1; echo "pytest exit $?" >> $O/pytest_c.log; tail -3 $O/pytest_c.log
cp gpurun_out/test_evidence.txt $O/test_evidence.txt 2>/dev/null
