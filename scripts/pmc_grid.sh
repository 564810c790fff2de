#!/bin/bash
# PMC passes of the fp64 F-solve kernel (fsolve_grid_kernel) on a config-5-like problem (50 entries per rank-64 system).
# usage (inside gpurun): scripts/pmc_grid.sh <outdir>
OUT=$1
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH" \
         "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "fsolve_grid" --output-format csv -d $R/gpurun_out/$OUT/p$i -o pmc -- python $R/scripts/bench_c5_scaled.py 200000 5000 > $R/gpurun_out/$OUT/p$i.log 2>&1 || echo "pass $i failed"
done
tail -1 $R/gpurun_out/$OUT/p1.log
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in sorted(glob.glob('$R/gpurun_out/$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        agg[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in agg.items():
    print('  %-28s %16.1f  (n=%d)' % (k, sum(v)/len(v), len(v)))
PY
