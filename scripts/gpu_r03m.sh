#!/bin/bash
# round 3: more apply slots per rank -- dist tests + c5 share at N=8 again
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests/test_dist.py -m gpu -x -q -k "unfused or gram_product or two_ranks" > $O/pytest_dist.log 2>&1; echo "pytest dist exit $?" >> $O/pytest_dist.log; tail -3 $O/pytest_dist.log | cut -c1-300
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo8 -o t -- python $R/scripts/shard_compute_times.py c5 8 timeshard > $O/solo8.log 2>&1
grep "^c5" $O/solo8.log
python $R/scripts/stats_table.py $O/solo8 > $O/solo8_kernel_stats.txt 2>&1; head -10 $O/solo8_kernel_stats.txt | cut -c1-70,96-165
