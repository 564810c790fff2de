#!/bin/bash
# Where does a CG launch spend its time?  bench.py with the fp32 library rebuilt under -DTRMF_HV_ABL=1 (Gram loads replaced by
# constants), =2 (AR phases off), =3 (both): the per-launch times of hv_tile_kernel against the production build.
# (Results are wrong by construction; only kernel durations are read.)  Build the variants first: make -C exp-trmf-nips16_amd abl  usage inside gpurun: scripts/hv_ablation.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for A in 0 1 2 3; do
  D=$R/exp-trmf-nips16_amd/trmf/corelib
  if [ $A != 0 ]; then D=/tmp/abl$A; mkdir -p $D; cp $R/exp-trmf-nips16_amd/build/abl/trmf_float32_abl$A.so $D/trmf_float32.so; cp $R/exp-trmf-nips16_amd/trmf/corelib/trmf_float64.so $D/; fi
  TRMF_CORELIB_DIR=$D rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl$A -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/abl$A.log 2>&1
  python $R/scripts/stats_table.py $O/abl$A | grep -E "hv_tile_kernel" | cut -c1-40,96-165 | sed "s/^/ABL=$A /"
done
