#!/bin/bash
# occupancy-oriented PMC pass for the F-solve kernel: usage scripts/pmc_occ.sh <outdir> [ENV=VAL ...]
OUT=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "fsolve" --output-format csv -d $R/gpurun_out/$OUT/p1 -o pmc -- python $R/scripts/bench_fsolve.py c3 > $R/gpurun_out/$OUT/p1.log 2>&1
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in sorted(glob.glob('$R/gpurun_out/$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        agg[row['Counter_Name']].append(float(row['Counter_Value']))
        extra={k:row[k] for k in ('VGPR_Count','Accum_VGPR_Count','SGPR_Count','LDS_Block_Size','Scratch_Size','Workgroup_Size','Grid_Size') if k in row}
print(extra)
for k,v in agg.items(): print('  %-28s %16.1f' % (k, sum(v)/len(v)))
g=sum(agg['GRBM_GUI_ACTIVE'])/len(agg['GRBM_GUI_ACTIVE'])/8
wc=sum(agg['SQ_WAVE_CYCLES'])/len(agg['SQ_WAVE_CYCLES'])*4
print('  cycles/dispatch %.0f ; avg resident waves per SIMD = %.2f ; MFMA busy %.1f%%' % (g, wc/(1024*g), 100*sum(agg['SQ_VALU_MFMA_BUSY_CYCLES'])/len(agg['SQ_VALU_MFMA_BUSY_CYCLES'])/(1024*g)))
PY
