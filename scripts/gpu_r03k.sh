#!/bin/bash
# round 3: time-sharded unfused CG + regression of the unfused single-GPU path
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests/test_dist.py -m gpu -x -q -k "unfused or gram_product or two_ranks" > $O/pytest_dist.log 2>&1; echo "pytest dist exit $?" >> $O/pytest_dist.log; tail -15 $O/pytest_dist.log | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_python_frontend.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "paper_script or config5 or fused_cg" > $O/pytest_full.log 2>&1; echo "pytest full exit $?" >> $O/pytest_full.log; tail -3 $O/pytest_full.log
