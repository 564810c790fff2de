#!/bin/bash
# kernel-trace statistics of bench.py on one config.  usage (inside gpurun): scripts/trace_config.sh <tag> <config> [extra bench flags]
TAG=$1; CFG=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o $CFG -- python $R/bench.py --config $CFG --no-cpu-baseline --steps 20 --warmup 5 "$@" > $O/trace.log 2>&1
grep -o '"value": [0-9.]*' $O/trace.log
python $R/scripts/stats_table.py $O/trace > $O/kernel_stats.txt 2>&1; head -${LINES_OUT:-10} $O/kernel_stats.txt
