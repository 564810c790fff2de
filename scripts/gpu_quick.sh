#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/gpu_quick.sh <tag> [pytest -k expression]
# persistent-kernel tests + the config-3 bench line (no CPU baseline); everything under gpurun_out/<tag>/
TAG=${1:-quick}; K=${2:-}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_persist.py -q -m gpu ${K:+-k "$K"} > $O/persist.log 2>&1; tail -4 $O/persist.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python -c "
import json; d=json.load(open('$O/bench_c3.json')); print(d['value'], d['phases_ms'])"
