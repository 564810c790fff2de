R=$GRAFT_REPO_ROOT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rep() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['phases_ms'].items() if k!='cg_iter'})"; }
echo "== c1p follow only"; TRMF_TEST=1 TRMF_NO_OVERLAP=1 rep --config c1p --steps 40 --warmup 10
echo "== c1p overlap only"; TRMF_TEST=1 TRMF_NO_CG_FOLLOW=1 rep --config c1p --steps 40 --warmup 10
export TRMF_TEST=1 TRMF_NO_OVERLAP=1
LINES_OUT=12 bash scripts/trace_config.sh r05b/c1p_follow c1p --steps 40 --warmup 10 --no-one-shot
