#!/usr/bin/env python3
"""The per-rank cost of the time-sharded CG as ONE persistent kernel per rank (last row of the table), measured
with the rank's kernel ALONE on the GPU in loop-back (the round-4 table rested on round-3 timings of the
launch-per-step kernel plus a guessed exchange).

    python scripts/shard_persist_solo.py [config=c3] [worlds=1,2,4,8]

For every world size N the session acts as rank 0, a middle rank and the last rank of N under the peer-less communicator
(trmf_dist_init_solo) with TRMF_CG=persist: F-solve on the rank's item rows, X-side Gram on its timestamps, and the persistent
SHARD kernel over its block of tiles -- which publishes its records into all N table copies (here: N stores into its own arena),
polls the FULL record table and its halo rows, and counts the other ranks' records / rows as arrived (PersistArgs::solo).  What is
missing against a real node is only the wait for the slowest peer and the xGMI store-to-load latency of an exchange (one hop,
priced separately in DESIGN.md).  N = 1 is the ordinary one-GPU persistent kernel (all tiles, real exchanges inside the GPU).
Times are HIP-event phase times on the solver stream, means over the last iterations; `us/pass` = CG part / (CG steps + 2)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))
import numpy as np   # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c3'
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '1,2,4,8').split(',')]
os.environ.update(TRMF_TEST='1', TRMF_FSHARD='shard', TRMF_GRAMX='shard', TRMF_CG='persist')
from trmf import dist as tdist, session, synth   # noqa: E402

cfg = synth.CONFIGS[cfgname]
dt = np.dtype(cfg['dtype'])
p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=dt, seed=0)
rows = []
for N in worlds:
    for rank in sorted({0, N // 2, N - 1}):
        if N > 1:
            tdist.init_solo(rank, N, dt)
        m = synth.initial_model(p['Y'], p['lag_set'], cfg['k'], seed=0)
        with session.Session(p['Y'], m, missing=True, log_norms=False, **synth.HYPER) as s:
            s.run(10); st = s.stats(8); desc = s.describe()
        if N > 1:
            tdist.finalize(dt)
        cg = float(np.mean([x['cg_iter'] for x in st]))
        ms_x, ms_xg = float(np.mean([x['ms_X'] for x in st])), float(np.mean([x['ms_X_gram'] for x in st]))
        r = dict(config=cfgname, world=N, rank=rank, ms_F_kernel=float(np.mean([x['ms_F_kernel'] for x in st])), ms_X_gram=ms_xg, ms_CG=ms_x - ms_xg,
                 cg_steps=cg, us_per_pass=1e3 * (ms_x - ms_xg) / (cg + 2), ms_Theta=float(np.mean([x['ms_LV'] for x in st][1::2] or [0])))
        rows.append(r)
        print('%s  N=%d rank %d  F-solve kernel %.3f ms   X-side Gram %.3f ms   CG (one persistent kernel) %.3f ms = %.1f steps + 2 passes x %.2f us   Theta %.3f ms   [%s]' % (
            cfgname, N, rank, r['ms_F_kernel'], ms_xg, r['ms_CG'], cg, r['us_per_pass'], r['ms_Theta'], desc[:90]))
        sys.stdout.flush()
# The same block of timestamps as a self-contained ONE-rank problem (rank 0's rows of Y, every item): the ordinary one-GPU persistent
# kernel over T / N timestamps = the same number of tiles per GPU, the same chain per tile, but records and halo rows exchanged at
# device scope inside the GPU by workgroups that really are in step -- the loop-back figure above minus the emulated peer's
# reaction time (its poll + ~3000 system-scope stores by one workgroup come AFTER the rank's own publish; real peers publish at the
# same time).  A rank's step on a real node lies between the two, plus one xGMI store-to-load latency.
os.environ.pop('TRMF_CG'); os.environ.pop('TRMF_FSHARD'); os.environ.pop('TRMF_GRAMX')
for N in worlds:
    TI = 25 if cfg['k'] == 40 else None
    if TI is None:
        break
    nbt = (cfg['T'] + TI - 1) // TI
    rows_n = min(cfg['T'], ((nbt + N - 1) // N) * TI)
    Y = p['Y'][:rows_n]
    m = synth.initial_model(Y, p['lag_set'], cfg['k'], seed=0)
    with session.Session(Y, m, missing=True, log_norms=False, **synth.HYPER) as s:
        s.run(10); st = s.stats(8); desc = s.describe()
    cg = float(np.mean([x['cg_iter'] for x in st]))
    ms_x, ms_xg = float(np.mean([x['ms_X'] for x in st])), float(np.mean([x['ms_X_gram'] for x in st]))
    r = dict(config=cfgname, surrogate_T=rows_n, world_equivalent=N, ms_X_gram=ms_xg, ms_CG=ms_x - ms_xg, cg_steps=cg, us_per_pass=1e3 * (ms_x - ms_xg) / (cg + 2))
    rows.append(r)
    print('%s  one rank over the first %d timestamps (= the block of one rank at N=%d; %d narrow tiles, the session says: %s): X-side Gram %.3f ms   CG (one persistent kernel) %.3f ms = %.1f steps + 2 passes x %.2f us   [%s]' % (
        cfgname, rows_n, N, (rows_n + TI - 1) // TI, desc.split('; ')[-1], ms_xg, r['ms_CG'], cg, r['us_per_pass'], desc[:70]))
    sys.stdout.flush()
print(json.dumps(rows))
