#!/bin/bash
# A/B of two library builds on the GPU box: digests (bit-neutrality) and bench lines.  usage (inside gpurun): scripts/ab_builds.sh <dir of the OLD libs>
R=$GRAFT_REPO_ROOT; cd $R; OLD=$1
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
TRMF_CORELIB_DIR=$OLD python scripts/digest_run.py > gpurun_out/digest_old.txt 2>&1
python scripts/digest_run.py > gpurun_out/digest_new.txt 2>&1
diff gpurun_out/digest_old.txt gpurun_out/digest_new.txt > /dev/null && echo "DIGESTS IDENTICAL" || { echo "DIGESTS DIFFER"; diff gpurun_out/digest_old.txt gpurun_out/digest_new.txt | head; }
rep() { python bench.py --no-cpu-baseline --no-one-shot "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k:round(v,4) for k,v in d['phases_ms'].items() if k in ('F','X','Theta')}, (d.get('roofline_x') or {}).get('cg',{}).get('us_per_pass'))"; }
for lib in old new old new; do
  if [ $lib = old ]; then export TRMF_CORELIB_DIR=$OLD; else unset TRMF_CORELIB_DIR; fi
  echo "== $lib c3"; rep
  echo "== $lib c2"; rep --config c2 --steps 40 --warmup 10
done
