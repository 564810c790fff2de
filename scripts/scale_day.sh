#!/bin/bash
# Hardware-day checklist for DESIGN.md section 6: everything that has only ever run with processes / threads standing in
# for GPUs, on a real multi-GPU node, in one go.   usage:  scripts/scale_day.sh [outdir] [gpu counts, default "1 2 4 8"]
#   CFGS=c3 restricts the configurations.
#   DRY=1 scripts/scale_day.sh out "1 2 4"      one-GPU box: N ranks as virtual ranks on device 0 (in-process mode, TRMF_DEVICES=0,0,...);
#                                                the SPMD legs (torchrun, one process per GPU) are skipped
# Per N and config (c3 = config 4's workload, c5): (1) SPMD launch -- one process per GPU under torch.distributed.run, RCCL over xGMI -- and
# (2) the in-process launch (no launcher: ranks as threads behind the unchanged entry points, TRMF_DEVICES); for each the bench line's
# value, the library's describe() (which candidate the measure-once rule picked, every candidate's slowest-rank time, whether the peer-to-peer
# arenas came up) and the phase split.  Then tests/test_dist.py + tests/test_gpu_devices.py with a device per rank, and the table against
# the prediction of DESIGN.md section 6.
OUT=${1:-gpurun_out/scale_day}; NS=${2:-"1 2 4 8"}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python -c "import sys; sys.path.insert(0, 'exp-trmf-nips16_amd'); import numpy as np; from trmf import session; print(session.lib_for(np.float32).trmf_device_count())")
echo "devices visible: $NDEV   dry run: ${DRY:-0}" | tee $OUT/summary.txt
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
    w = d.get('windows') or {}
    print('%-26s %8.1f iter/s  (median %s, spread %s)  ms/step %.3f  F %s X %s  | %s' % (sys.argv[2], d['value'], w.get('iter_per_s_median'), w.get('spread'),
          d['ms_per_step'], (d.get('phases_ms') or {}).get('F'), (d.get('phases_ms') or {}).get('X'), d['config']['parallelism']))
except Exception as exc:
    print('%-26s FAILED (%s)' % (sys.argv[2], exc))
PY
}
for CFG in ${CFGS:-c3 c5}; do
  STEPS=20; WARM=5; [ $CFG = c5 ] && { STEPS=6; WARM=2; }
  for N in $NS; do
    if [ "${DRY:-0}" = 1 ]; then DEVS=$(python -c "print(','.join(['0'] * $N))"); else DEVS=$(python -c "print(','.join(str(i) for i in range($N)))"); fi
    if [ $N -gt 1 ] && [ "${DRY:-0}" != 1 ] && [ $N -le $NDEV ]; then
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --config $CFG \
        --steps $STEPS --warmup $WARM --no-cpu-baseline --no-one-shot > $OUT/spmd_${CFG}_$N.json 2> $OUT/spmd_${CFG}_$N.err
      summ $OUT/spmd_${CFG}_$N.json "spmd $CFG N=$N" | tee -a $OUT/summary.txt
      for FORM in replicate timeshard p2p persist; do          # every candidate by itself: what the measure-once rule chose FROM
        python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --config $CFG \
          --steps $STEPS --warmup $WARM --no-cpu-baseline --no-one-shot --cg $FORM > $OUT/spmd_${CFG}_${N}_$FORM.json 2> $OUT/spmd_${CFG}_${N}_$FORM.err
        summ $OUT/spmd_${CFG}_${N}_$FORM.json "spmd $CFG N=$N $FORM" | tee -a $OUT/summary.txt
      done
    fi
    if [ $N -eq 1 ]; then
      python bench.py --config $CFG --steps $STEPS --warmup $WARM --no-cpu-baseline --no-one-shot > $OUT/one_${CFG}.json 2> $OUT/one_${CFG}.err
      summ $OUT/one_${CFG}.json "one GPU $CFG" | tee -a $OUT/summary.txt
    elif [ "${DRY:-0}" = 1 ] || [ $N -le $NDEV ]; then
      python bench.py --gpus $N --devices $DEVS --config $CFG --steps $STEPS --warmup $WARM --no-cpu-baseline --no-one-shot > $OUT/inproc_${CFG}_$N.json 2> $OUT/inproc_${CFG}_$N.err
      summ $OUT/inproc_${CFG}_$N.json "in-process $CFG N=$N" | tee -a $OUT/summary.txt
    fi
  done
done
if [ "${DRY:-0}" != 1 ]; then
  # the multi-rank tests with a device per rank (dist_worker picks device = rank when TRMF_TEST_DEVICE_PER_RANK is set)
  TRMF_TEST_DEVICE_PER_RANK=1 timeout 3000 python -m pytest tests/test_dist.py tests/test_gpu_devices.py -x -q -m gpu > $OUT/pytest_multi.log 2>&1; tail -3 $OUT/pytest_multi.log | tee -a $OUT/summary.txt
fi
python - $OUT <<'PY' | tee -a $OUT/summary.txt
# measured against DESIGN.md section 6's prediction for config 4 (c3's workload on N GPUs): 8 GPUs ~2x, never near-linear
import glob, json, os, sys
out = sys.argv[1]
def val(f):
    try: return json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])['value']
    except Exception: return None
one = val(os.path.join(out, 'one_c3.json'))
pred = {2: (1.3, 1.6), 4: (1.6, 2.0), 8: (1.7, 2.2)}
for kind in ('spmd', 'inproc'):
    for n in (2, 4, 8):
        v = val(os.path.join(out, '%s_c3_%d.json' % (kind, n)))
        if one and v:
            lo, hi = pred[n]
            print('config 4 on %d GPUs (%s): %.1f iter/s = %.2fx of one GPU; DESIGN.md section 6 predicted %.1f-%.1fx' % (n, kind, v, v / one, lo, hi))
PY
