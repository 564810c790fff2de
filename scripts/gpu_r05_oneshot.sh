#!/bin/bash
# round 5: the one-shot c_trmf_train path -- its tests, its wall-time split, the bench line with one_shot / roofline_x
TAG=${1:-r05b}; O=gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_oneshot.py tests/test_gpu_persist.py -x -q -m gpu > $O/pytest_oneshot.log 2>&1; tail -15 $O/pytest_oneshot.log
python scripts/oneshot_profile.py c3 10 5 > $O/oneshot_c3.txt 2>&1; cat $O/oneshot_c3.txt
python scripts/oneshot_profile.py c2 10 4 > $O/oneshot_c2.txt 2>&1; cat $O/oneshot_c2.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 2500 $O/bench_c3.json; tail -5 $O/bench_c3.err
