#!/bin/bash
# round 3: IPC flag ping-pong between two processes, full GPU suite with margins, bench c3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
for MEM in plain uncached; do
  rm -f /tmp/ipch.bin /tmp/ipch.bin.ready
  (timeout 60 scripts/ubench/ipc_flag server /tmp/ipch.bin $MEM > $O/ipc_server_$MEM.txt 2>&1 &)
  timeout 60 scripts/ubench/ipc_flag client /tmp/ipch.bin $MEM > $O/ipc_client_$MEM.txt 2>&1
  sleep 1; cat $O/ipc_server_$MEM.txt $O/ipc_client_$MEM.txt
done
timeout 3000 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
grep -E "FUZZ-MARGIN" $O/pytest.log | sed 's/.*FUZZ-MARGIN/FUZZ-MARGIN/' > $O/fuzz_margins.txt
grep -E "config 3 vs the reference|config 4 full size|c5 full size|c3 full size" $O/pytest.log
tail -4 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; r=json.load(open('$O/bench_c3.json')); print(r['value'], r['phases_ms'], r['roofline']['frac'])"
