mkdir -p gpurun_out/r04h
for cfg in c2 c1 c1p; do for p in 1 0; do
  TRMF_PERSIST=$p python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg persist=$p', round(d['value'],1), 'iter/s', {k:round(v,4) for k,v in d['phases_ms'].items() if k!='cg_iter'}, d['config']['parallelism'])"
done; done 2>&1 | tee gpurun_out/r04h/small.txt
