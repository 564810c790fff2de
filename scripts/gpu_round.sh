#!/bin/bash
# One gpurun call: GPU test suite, kernel-trace stats of bench.py (config 3), PMC passes of the F-solve kernel,
# WRITE_SIZE calibration.  usage: scripts/gpu_round.sh <tag> [tests|notests]
TAG=${1:-r02}; MODE=${2:-tests}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ "$MODE" = tests ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
  tail -5 $O/pytest.log
fi
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c3 -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/trace.log 2>&1
python $R/scripts/stats_table.py $O/trace > $O/kernel_stats.txt 2>&1; head -30 $O/kernel_stats.txt
bash $R/scripts/pmc_fsolve.sh $TAG/pmc > $O/pmc_fsolve.txt 2>&1; tail -45 $O/pmc_fsolve.txt
bash $R/scripts/pmc_kernel.sh $TAG/pmc_gramx "gram_x_kernel" > $O/pmc_gram_x.txt 2>&1; tail -22 $O/pmc_gram_x.txt
bash $R/scripts/pmc_kernel.sh $TAG/pmc_hv "hv_tile_kernel" > $O/pmc_hv_tile.txt 2>&1; tail -22 $O/pmc_hv_tile.txt
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/wcal_$C -o w -- $R/scripts/ubench/write_calib > $O/wcal_$C.log 2>&1
done
python - <<PY > $O/write_calib.txt 2>&1
import csv, glob, collections
for C in ('WRITE_SIZE', 'FETCH_SIZE'):
    agg = collections.defaultdict(list)
    for f in glob.glob('$O/wcal_%s/**/*counter_collection.csv' % C, recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']].append(float(row['Counter_Value']))
    for k, v in sorted(agg.items()):
        print('%-12s %-40s mean %.1f (n=%d)  [19200000 bytes written = 18750 KB]' % (C, k[:40], sum(v) / len(v), len(v)))
PY
cat $O/write_calib.txt
