#!/bin/bash
# phase stamps of the persistent CG kernel (make -C exp-trmf-nips16_amd prof first).  usage (inside gpurun): scripts/persist_prof.sh [bench flags]
R=$GRAFT_REPO_ROOT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TRMF_TEST=1
cp exp-trmf-nips16_amd/trmf/corelib/trmf_float64.so exp-trmf-nips16_amd/build/prof/ 2>/dev/null
TRMF_CORELIB_DIR=$R/exp-trmf-nips16_amd/build/prof/ TRMF_PERSIST_PROF=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-one-shot "$@" 2>&1 | grep -E "PERSIST_PROF tile mid row +[4-9]:"
