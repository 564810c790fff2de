#!/bin/bash
# round 3 measurement pass: kernel-trace stats (c3, c5), PMC of the F-solve kernels (c3 fp32, c5 fp64), PMC of the CG tile kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 400 $O/bench_c3.json
LINES_OUT=24 bash scripts/trace_config.sh r03e/c3 c3 > $O/trace_c3.txt 2>&1; cat $O/trace_c3.txt | cut -c1-170
bash scripts/pmc_fsolve.sh r03e/pmc_c3 c3 > $O/pmc_fsolve_c3.txt 2>&1; tail -40 $O/pmc_fsolve_c3.txt
bash scripts/pmc_kernel.sh r03e/pmc_hv "hv_tile_kernel" > $O/pmc_hv_tile.txt 2>&1; tail -24 $O/pmc_hv_tile.txt
python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 700 $O/bench_c5.json
LINES_OUT=24 bash scripts/trace_config.sh r03e/c5 c5 --steps 6 --warmup 2 > $O/trace_c5.txt 2>&1; cat $O/trace_c5.txt | cut -c1-170
bash scripts/pmc_fsolve.sh r03e/pmc_c5 c5 > $O/pmc_fsolve_c5.txt 2>&1; tail -40 $O/pmc_fsolve_c5.txt
