R=$GRAFT_REPO_ROOT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rep() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['phases_ms'].items() if k!='cg_iter'})"; }
for cfg in c3 c2 c1 c1p; do
  echo "== $cfg new"; rep --config $cfg --steps 40 --warmup 10; rep --config $cfg --steps 40 --warmup 10
done
echo "== c1p forced overlap off / follow off"; TRMF_TEST=1 TRMF_NO_OVERLAP=1 rep --config c1p --steps 40 --warmup 10; TRMF_TEST=1 TRMF_NO_CG_FOLLOW=1 rep --config c1p --steps 40 --warmup 10
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py --deselect tests/test_dist.py 2>&1 | tail -5
