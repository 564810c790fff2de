#!/bin/bash
# round 3: kernel-trace of a rank's share (solo communicator) at N = 2, 4, 8, time-sharded; ASan/UBSan host run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for N in 2 4 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo$N -o t -- python $R/scripts/shard_compute_times.py c3 $N timeshard > $O/solo$N.log 2>&1
  python $R/scripts/stats_table.py $O/solo$N > $O/solo${N}_kernel_stats.txt 2>&1; echo "== N=$N"; head -12 $O/solo${N}_kernel_stats.txt | cut -c1-150
done
cd $R
bash scripts/asan_gpu.sh $O/asan.log; tail -5 $O/asan.log
