#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash scripts/asan_gpu.sh $O/ubsan.log; tail -3 $O/ubsan.log
timeout 2400 python -m pytest tests/test_dist.py -m gpu -x -q > $O/pytest_dist.log 2>&1; echo "pytest dist exit $?" >> $O/pytest_dist.log; tail -3 $O/pytest_dist.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
