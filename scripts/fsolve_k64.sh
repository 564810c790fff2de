mkdir -p gpurun_out/r04f; O=gpurun_out/r04f/fsolve_k64.txt; : > $O
for cfg in c3k56 c3k64; do
  echo "== $cfg, quad form, 2 wavefronts per SIMD (round 4 default, no scratch)" >> $O; python scripts/bench_fsolve.py $cfg >> $O 2>&1
  echo "== $cfg, quad form, 3 wavefronts per SIMD (round 3: 296 / 460 B of scratch)" >> $O; TRMF_CORELIB_DIR=$PWD/exp-trmf-nips16_amd/build/q3 python scripts/bench_fsolve.py $cfg >> $O 2>&1
  echo "== $cfg, wave form (TRMF_FSOLVE=wave)" >> $O; TRMF_FSOLVE=wave python scripts/bench_fsolve.py $cfg >> $O 2>&1
done
cat $O
