#!/bin/bash
# round 3: after moving the CG's closing work into cg_close_kernel -- tests, bench + trace c3, PMC of the tile kernel, shard compute shares
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests/test_dist.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_abi.py tests/test_python_frontend.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "config3 or fused_cg" > $O/pytest_full.log 2>&1; echo "pytest full exit $?" >> $O/pytest_full.log
grep -E "config 3 vs|c3 full|passed|failed" $O/pytest_full.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; r=json.load(open('$O/bench_c3.json')); print(r['value'], r['phases_ms'], r['roofline']['frac'], r['roofline']['traffic'])"
LINES_OUT=14 bash scripts/trace_config.sh r03f/c3 c3 > $O/trace_c3.txt 2>&1; cat $O/trace_c3.txt | cut -c1-165
bash scripts/pmc_kernel.sh r03f/pmc_hv "hv_tile_kernel" > $O/pmc_hv_tile.txt 2>&1; tail -24 $O/pmc_hv_tile.txt
timeout 900 python scripts/shard_compute_times.py c3 1,2,4,8 replicate,timeshard > $O/shard_compute_times.txt 2>&1; grep -v "^\[" $O/shard_compute_times.txt | head -30
