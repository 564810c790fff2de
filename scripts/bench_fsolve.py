#!/usr/bin/env python3
"""F-solve only: kernel time (HIP events on the solver stream) + parity of a row sample vs closed form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))
import numpy as np
from trmf import session, synth
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c3'
cfg = synth.CONFIGS[cfgname]; dt = np.dtype(cfg['dtype'])
p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=dt, seed=0)
m = synth.initial_model(p['Y'], p['lag_set'], cfg['k'], seed=0)
W0 = m.W.copy()
big = 10 ** 6
with session.Session(p['Y'], m, missing=True, period_W=big, period_Lag=big, **synth.HYPER) as s:
    s.run(12); st = s.stats(10); s.download(); B = s.fsolve_bytes()
ms = np.array([x['ms_F_kernel'] for x in st])
print('%s nnz=%d fsolve kernel: avg %.1f us  min %.1f  max %.1f  ->  %.0f GB/s (%.1f%% of 8 TB/s) on B_F=%.3f GB' % (
    cfgname, p['Y'].nnz, 1e3 * ms.mean(), 1e3 * ms.min(), 1e3 * ms.max(), B / ms.mean() / 1e6, 100 * B / ms.mean() / 1e6 / 8000, B / 1e9))
Yc = p['Y'].tocsc(); k = cfg['k']; worst = 0
for i in np.random.RandomState(0).choice(cfg['n'], 300, replace=False):
    tt = Yc.indices[Yc.indptr[i]:Yc.indptr[i + 1]]
    if len(tt) == 0: continue
    P = W0[tt].astype(np.float64); y = Yc.data[Yc.indptr[i]:Yc.indptr[i + 1]].astype(np.float64)
    ref = np.linalg.solve(P.T @ P + synth.HYPER['lambdaI'] * np.eye(k), P.T @ y)
    worst = max(worst, np.abs(m.H[i] - ref).max() / np.abs(ref).max())
print('max rel err of 300 sampled rows vs fp64 closed form: %.2e' % worst)
