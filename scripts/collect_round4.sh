#!/bin/bash
# usage (build container, after `gpurun -- bash scripts/gpu_round4.sh <tag> tests`): scripts/collect_round4.sh <tag>
# copies the run's summaries from gpurun_out/<tag>/ to profiles/r04_* and regenerates profiles/fsolve_traffic.json
set -e
TAG=${1:-r04}; O=gpurun_out/$TAG; P=profiles
for c in c1 c1p c2 c3 c5 c3_launch_per_step; do cp $O/bench_$c.json $P/r04_bench_$c.json; done
cp $O/bench_c3_repeats.txt $P/r04_bench_c3_repeats.txt
cp $O/c3/kernel_stats.txt $P/r04_c3_kernel_stats.txt
cp $O/c3_launch/kernel_stats.txt $P/r04_c3_kernel_stats_launch_per_step.txt
cp $O/c5/kernel_stats.txt $P/r04_c5_kernel_stats.txt
grep -v "rocprofv3\] tool init" $O/pmc_fsolve_c3.txt > $P/r04_pmc_fsolve_c3.txt
grep -v "rocprofv3\] tool init" $O/pmc_fsolve_c5.txt > $P/r04_pmc_fsolve_c5.txt
cp $O/pmc_persist.txt $P/r04_pmc_persist.txt
[ -s $O/persist_timeline.txt ] && cp $O/persist_timeline.txt $P/r04_persist_timeline.txt
cp $O/small_configs.txt $P/r04_small_configs_persist_vs_launch.txt
[ -s $O/test_evidence.txt ] && cp $O/test_evidence.txt $P/r04_test_evidence.txt
[ -s $O/fuzz_margins.txt ] && cp $O/fuzz_margins.txt $P/r04_fuzz_margins.txt
tail -2 $O/pytest.log > $P/r04_gpu_tests_summary.txt
python scripts/make_traffic_json.py $P/r04_pmc_fsolve_c3.txt c3
python scripts/make_traffic_json.py $P/r04_pmc_fsolve_c5.txt c5
