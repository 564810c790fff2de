R=$GRAFT_REPO_ROOT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TRMF_TEST=1
rep() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['phases_ms'].items() if k!='cg_iter'}, d['roofline_x']['cg']['us_per_pass'], d['config']['parallelism'])"; }
echo "== c3 narrow"; rep; rep
echo "== c3 wide"; TRMF_WIDE=1 rep; TRMF_WIDE=1 rep
