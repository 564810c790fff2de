#!/bin/bash
# PMC passes for the F-solve kernel (separate passes; kernel-trace only, per the gpurun rules).
# usage: scripts/pmc_fsolve.sh <outdir> [config: c3 | c5] [extra env assignments]
set -e
OUT=$1; shift
CFG=c3
case "$1" in c3|c5) CFG=$1; shift;; esac
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
         "GRBM_GUI_ACTIVE GRBM_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "fsolve" --output-format csv -d $R/gpurun_out/$OUT/p$i -o pmc -- python $R/scripts/bench_fsolve.py $CFG > $R/gpurun_out/$OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv, glob, collections, os
agg=collections.defaultdict(list)
for f in sorted(glob.glob('$R/gpurun_out/$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        agg[row['Counter_Name']].append(float(row['Counter_Value']))
print(open('$R/gpurun_out/$OUT/p1.log').read().strip().splitlines()[0])
print('counter averages per dispatch (fsolve kernels), config $CFG:')
for k,v in agg.items(): print('  %-32s %16.1f  (n=%d)' % (k, sum(v)/len(v), len(v)))
PY
