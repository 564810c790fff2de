#!/bin/bash
# round 3: a rank's compute share at config 5 (fp64, unfused CG), solo communicator, N = 1 and 8
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1500 python $R/scripts/shard_compute_times.py c5 1,8 replicate,shard,timeshard > $O/c5_shares.txt 2>&1; grep "^c5" $O/c5_shares.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo8 -o t -- python $R/scripts/shard_compute_times.py c5 8 timeshard > $O/solo8.log 2>&1
python $R/scripts/stats_table.py $O/solo8 > $O/solo8_kernel_stats.txt 2>&1; head -14 $O/solo8_kernel_stats.txt | cut -c1-70,96-165
