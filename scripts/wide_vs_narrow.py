import os, sys
sys.path.insert(0, 'exp-trmf-nips16_amd'); sys.path.insert(0, 'tests')
os.environ['TRMF_TEST'] = '1'
import numpy as np
from helpers import make_model
from trmf import session, synth
def run(p, m0, dtype, iters, tile):
    os.environ['TRMF_TILE'] = tile
    model = make_model(m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
    with session.Session(p['Y'].astype(dtype), model, missing=True, **synth.HYPER) as s:
        s.run(iters); st = s.stats(iters); s.download()
    return model, st
for shape in [dict(n=3000, T=1200, k=40, nlag=16, density=0.04), dict(n=1500, T=700, k=64, nlag=6, density=0.05), dict(n=1200, T=2600, k=40, nlag=16, density=0.02)]:
    p = synth.sparse_problem(n=shape['n'], T=shape['T'], k=shape['k'], nlag=shape['nlag'], density=shape['density'], dtype=np.float64, seed=21)
    m0 = synth.initial_model(p['Y'], p['lag_set'], shape['k'], seed=21)
    for dtype in (np.float32, np.float64):
        a, sa = run(p, m0, dtype, 4, 'wide'); b, sb = run(p, m0, dtype, 4, 'narrow')
        rel = lambda u, v: np.linalg.norm(u.astype(np.float64) - v.astype(np.float64)) / np.linalg.norm(v.astype(np.float64))
        print(shape['k'], dtype.__name__, 'W', rel(a.W, b.W), 'H', rel(a.H, b.H), 'Th', rel(a.lag_val, b.lag_val), 'cg', [x['cg_iter'] for x in sa], [x['cg_iter'] for x in sb],
              'f', [abs(x['f'] - y['f']) / abs(y['f']) for x, y in zip(sa, sb)])
