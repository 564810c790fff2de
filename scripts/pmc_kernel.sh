#!/bin/bash
# generic PMC passes for kernels matching a regex while running bench.py: [CFG=c5] scripts/pmc_kernel.sh <outdir> <regex>
OUT=$1; RE=$2
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
         "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$RE" --output-format csv -d $R/gpurun_out/$OUT/p$i -o pmc -- python $R/bench.py --config ${CFG:-c3} --no-cpu-baseline --no-one-shot --repeat 1 --steps 4 --warmup 1 > $R/gpurun_out/$OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in sorted(glob.glob('$R/gpurun_out/$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        agg[row['Counter_Name']].append(float(row['Counter_Value']))
# only the "real" (non early-exit) dispatches: keep dispatches whose wave cycles are above 25% of the max
print('counter: mean over dispatches with value > 25% of max (n)')
for k,v in agg.items():
    m=max(v); big=[x for x in v if x>0.25*m] or v
    print('  %-28s %16.1f  (n=%d of %d)' % (k, sum(big)/len(big), len(big), len(v)))
PY
