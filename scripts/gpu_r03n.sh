#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests/test_dist.py -m gpu -x -q > $O/pytest_dist.log 2>&1; echo "pytest dist exit $?" >> $O/pytest_dist.log; tail -5 $O/pytest_dist.log | cut -c1-300
