#!/usr/bin/env python3
"""Soak test of the persistent CG kernel: thousands of solves back to back (every one a fresh epoch range of the tagged buffers) at two
shapes; fails on any error of trmf_session_sync (a poll that ran into its bound) or a non-finite / increasing objective."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))
import numpy as np
from trmf import session, synth
for cfgname, iters in (('c3', 3000), ('c2', 20000)):
    cfg = synth.CONFIGS[cfgname]
    p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=np.dtype(cfg['dtype']), seed=0)
    m = synth.initial_model(p['Y'], p['lag_set'], cfg['k'], seed=0)
    with session.Session(p['Y'], m, missing=True, log_norms=False, **synth.HYPER) as s:
        J = [s.run(5).objective()]
        t0 = time.time()
        for chunk in range(10):
            s.run(iters // 10).sync()
            J.append(s.objective())
        dt = time.time() - t0
        st = s.stats(8)
        print('%s: %d iterations in %.2f s (%.0f iter/s incl. 10 objective evaluations); %s; J %.8g -> %.8g; last CG counts %s; accepted %s' % (
            cfgname, iters, dt, iters / dt, s.describe(), J[0], J[-1], [x['cg_iter'] for x in st], all(x['accepted'] == 1 for x in st)))
        # (J has no lambdaLag |Theta|^2 term, so the Theta step may raise it in the last digits once converged)
        assert all(np.isfinite(J)) and J[-1] < J[0] and all(b <= a * (1 + 1e-4) for a, b in zip(J, J[1:])), J
print('soak ok')
