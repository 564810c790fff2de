#!/bin/bash
# round 3: upper-triangle Gram reads in hv_tile_kernel -- parity + bench + trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_dist.py tests/test_python_frontend.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "config3 or fused_cg" > $O/pytest_full.log 2>&1; echo "pytest full exit $?" >> $O/pytest_full.log
grep -E "config 3 vs|c3 full|passed|failed" $O/pytest_full.log
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3_$i.json 2> $O/bench_c3.err; python -c "
import json; r=json.load(open('$O/bench_c3_$i.json')); print(r['value'], r['phases_ms'], r['roofline']['frac'])"; done
LINES_OUT=12 bash scripts/trace_config.sh r03i/c3 c3 > $O/trace_c3.txt 2>&1; cat $O/trace_c3.txt | cut -c1-165
python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null; python -c "
import json; r=json.load(open('$O/bench_c2.json')); print('c2', r['value'], r['phases_ms'])"
