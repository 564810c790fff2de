#!/usr/bin/env python3
"""A rank's COMPUTE share of the sharded ALS loop, measured alone on one GPU (compute columns).

    python scripts/shard_compute_times.py [config=c3] [worlds=1,2,4,8] [modes=replicate,timeshard]

For every world size N the session acts as rank 0 (and as the last rank) of N under the peer-less communicator
(trmf_dist_init_solo): it launches exactly the kernels that rank would launch -- F-solve on its item rows, X-side Gram on
its timestamps, the CG replicated (all tiles) or sharded over time (its tiles) -- and skips every exchange.  The times are
HIP-event phase times of the solver stream (F-solve kernel; X phase; Theta), means over the last iterations; with the
exchanges skipped the factors are not a solution, but the kernels' work does not depend on the values.  The time-sharded
X phase includes one host read-back of the CG's stop flag per solve (~15 us), as in a real run through the communicator."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))
os.environ.setdefault('TRMF_TEST', '1')   # TRMF_FSHARD / TRMF_GRAMX below are test knobs (read only under TRMF_TEST)
import numpy as np   # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c3'
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '1,2,4,8').split(',')]
modes = (sys.argv[3] if len(sys.argv) > 3 else 'replicate,timeshard').split(',')
os.environ['TRMF_FSHARD'] = 'shard'
os.environ['TRMF_GRAMX'] = 'shard'
from trmf import dist as tdist, session, synth   # noqa: E402

cfg = synth.CONFIGS[cfgname]
dt = np.dtype(cfg['dtype'])
p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=dt, seed=0)
rows = []
for N in worlds:
    for mode in modes if N > 1 else ['replicate']:
        os.environ['TRMF_CG'] = mode
        for rank in sorted({0, N - 1}):
            if N > 1:
                tdist.init_solo(rank, N, dt)
            m = synth.initial_model(p['Y'], p['lag_set'], cfg['k'], seed=0)
            with session.Session(p['Y'], m, missing=True, log_norms=False, **synth.HYPER) as s:
                s.run(8); st = s.stats(6)
            if N > 1:
                tdist.finalize(dt)
            r = dict(config=cfgname, world=N, rank=rank, cg=mode, ms_F_kernel=float(np.mean([x['ms_F_kernel'] for x in st])),
                     ms_X=float(np.mean([x['ms_X'] for x in st])), ms_Theta=float(np.mean([x['ms_LV'] for x in st][1::2] or [0])),
                     cg_iter=[int(x['cg_iter']) for x in st])
            rows.append(r)
            print('%s  N=%d rank %d  CG %-9s  F-solve kernel %.3f ms   X phase %.3f ms   Theta %.3f ms   CG steps %s' % (
                cfgname, N, rank, mode, r['ms_F_kernel'], r['ms_X'], r['ms_Theta'], r['cg_iter']))
            sys.stdout.flush()
print(json.dumps(rows))
