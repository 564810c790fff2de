"""Pass time of the persistent CG kernel against the timestamps per tile, on the block of timestamps ONE rank of N owns at config 3
(T / N timestamps as a self-contained one-rank problem: the tiles a rank would run, exchanges at device scope).  VERDICT r4 item 2 asks
whether a rank that owns fewer tiles than CUs can shorten its pass by spreading the work over more workgroups: smaller tiles ARE that.
usage (GPU box): python scripts/tile_rows_sweep.py [config] [N,N,...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'exp-trmf-nips16_amd'))
os.environ['TRMF_TEST'] = '1'
import numpy as np
from trmf import session, synth
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c3'
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '8,4,2').split(',')]
cfg = synth.CONFIGS[cfgname]
p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=np.float32, seed=0)
for N in worlds:
    rows_n = (cfg['T'] + N - 1) // N
    Y = p['Y'][:rows_n]
    for TI in (25, 20, 16, 13, 10, 8, 5, 4):
        os.environ['TRMF_HV_TI'] = str(TI)
        m = synth.initial_model(Y, p['lag_set'], cfg['k'], seed=0)
        with session.Session(Y, m, missing=True, log_norms=False, **synth.HYPER) as s:
            s.run(12); st = s.stats(8); desc = s.describe()
        cg = float(np.mean([x['cg_iter'] for x in st]))
        ms_x, ms_xg = float(np.mean([x['ms_X'] for x in st])), float(np.mean([x['ms_X_gram'] for x in st]))
        print('%s  block of one rank at N=%d (%d timestamps)  TI=%2d: CG %.3f ms = %.1f steps + 2 passes x %.2f us   [%s]' % (
            cfgname, N, rows_n, TI, ms_x - ms_xg, cg, 1e3 * (ms_x - ms_xg) / (cg + 2), desc.split('; ', 1)[-1]))
        sys.stdout.flush()
