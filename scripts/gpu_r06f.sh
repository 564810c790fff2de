#!/bin/bash
# Round 6, sixth GPU call: final kernel sources -- digests against the round-5 build, the in-process / split tests, the longest-first
# dispatch order on zipf, PMC of the F-solve at c3 and c5 (-> profiles/fsolve_traffic.json), the round's config-3 bench line + kernel
# trace, UBSan run.
TAG=${1:-r06f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/test_evidence.txt
TRMF_CORELIB_DIR=$R/exp-trmf-nips16_amd/build/old python scripts/digest_run.py > $O/digest_old.txt 2>&1; python scripts/digest_run.py > $O/digest_new.txt 2>&1
diff $O/digest_old.txt $O/digest_new.txt > /dev/null && echo "DIGESTS IDENTICAL to the round-5 build" || { echo "DIGESTS DIFFER"; diff $O/digest_old.txt $O/digest_new.txt | head; }
timeout 1800 python -m pytest tests/test_gpu_devices.py tests/test_gpu_split.py tests/test_gpu_persist.py -x -q -m gpu -k "not full_size" > $O/pytest_new.log 2>&1; echo "pytest exit $?" >> $O/pytest_new.log; tail -4 $O/pytest_new.log
line() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); w=d['windows']; print('$1', round(d['value'],1), 'iter/s median', round(w['iter_per_s_median'],1), 'F', d['phases_ms']['F'], 'X', d['phases_ms']['X'], 'Fk', d['roofline']['avg_kernel_ms'], 'frac', round(d['roofline']['frac'],3), 'Xgram', (d.get('roofline_x') or {}).get('gram',{}).get('avg_ms'))"; }
for cfg in zipf imp; do
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot 2>$O/bench_$cfg.err | tee $O/bench_$cfg.json | line $cfg
done
LINES_OUT=12 bash scripts/trace_config.sh $TAG/zipf zipf --no-one-shot --repeat 1 > $O/trace_zipf.txt 2>&1; cut -c1-165 $O/trace_zipf.txt
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 300 $O/bench_c3.json; echo
LINES_OUT=16 bash scripts/trace_config.sh $TAG/c3 c3 --no-one-shot --repeat 1 > $O/trace_c3.txt 2>&1; cut -c1-165 $O/trace_c3.txt
bash scripts/pmc_fsolve.sh $TAG/pmc_c3 c3 > $O/pmc_fsolve_c3.txt 2>&1; grep -E "FETCH_SIZE|WRITE_SIZE|TCC_" $O/pmc_fsolve_c3.txt
bash scripts/pmc_fsolve.sh $TAG/pmc_c5 c5 > $O/pmc_fsolve_c5.txt 2>&1; grep -E "FETCH_SIZE|WRITE_SIZE|TCC_" $O/pmc_fsolve_c5.txt
timeout 1500 bash scripts/asan_gpu.sh $O/ubsan_gpu_run.txt; tail -3 $O/ubsan_gpu_run.txt
