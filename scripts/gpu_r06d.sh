#!/bin/bash
# Round 6, fourth GPU call: A/B of the DPP pivot broadcast (build/nodpp = this tree with ds_swizzle) and against the round-5 build; the
# whole GPU suite on the final sources; UBSan run; the scale-day script as a dry run (virtual ranks on the one device).
TAG=${1:-r06d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/test_evidence.txt
bash scripts/ab_builds.sh $R/exp-trmf-nips16_amd/build/nodpp > $O/ab_dpp.txt 2>&1; cat $O/ab_dpp.txt
bash scripts/ab_builds.sh $R/exp-trmf-nips16_amd/build/old > $O/ab_r05.txt 2>&1; cat $O/ab_r05.txt
timeout 3300 python -m pytest tests -x -q -s -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -5 $O/pytest.log
cp gpurun_out/test_evidence.txt $O/test_evidence.txt 2>/dev/null
bash scripts/asan_gpu.sh $O/ubsan_gpu_run.txt; tail -4 $O/ubsan_gpu_run.txt
DRY=1 bash scripts/scale_day.sh gpurun_out/$TAG/scale_day "1 2 4" > $O/scale_day.log 2>&1; cat $O/scale_day/summary.txt
