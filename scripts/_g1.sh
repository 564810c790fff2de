R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
rep() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-shot "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['phases_ms'], d['roofline']['frac'], (d.get('roofline_x') or {}).get('cg'))"; }
echo "== c3 default"; rep; rep
for ti in 20 22 24; do echo "== c3 TI=$ti"; TRMF_TEST=1 TRMF_HV_TI=$ti rep; TRMF_TEST=1 TRMF_HV_TI=$ti rep; done
echo "== c1p trace"
LINES_OUT=30 bash scripts/trace_config.sh r05b/c1p c1p --steps 40 --warmup 10 --no-one-shot
