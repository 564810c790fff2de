R=$GRAFT_REPO_ROOT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rep() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['phases_ms'].items() if k!='cg_iter'}, (d.get('roofline_x') or {}).get('cg',{}).get('us_per_pass'), d['config']['parallelism'])"; }
echo "== c3 default (wide)"; rep; rep; rep
echo "== c3 narrow"; TRMF_TEST=1 TRMF_TILE=narrow rep; TRMF_TEST=1 TRMF_TILE=narrow rep
echo "== c3 wide, launch per step"; TRMF_PERSIST=0 rep
echo "== c3 narrow, launch per step"; TRMF_TEST=1 TRMF_TILE=narrow TRMF_PERSIST=0 rep
echo "== c2"; rep --config c2 --steps 40 --warmup 10
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_dist.py 2>&1 | tail -5
