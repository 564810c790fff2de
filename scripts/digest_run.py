"""SHA-256 digests of the factors after a few ALS iterations at several shapes -- run with TRMF_CORELIB_DIR pointing at two builds to check
that a change is bit-neutral (usage: python scripts/digest_run.py)."""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'exp-trmf-nips16_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
os.environ.setdefault('TRMF_TEST', '1')
import numpy as np
from helpers import make_model
from trmf import session, synth
def dig(*arrs):
    h = hashlib.sha256()
    for a in arrs: h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]
cases = [('sparse k40 fused', dict(n=3000, T=1200, k=40, nlag=16, density=0.04), None, True),
         ('sparse k16', dict(n=2000, T=2500, k=16, nlag=8, density=0.02), None, True),
         ('sparse k64', dict(n=1500, T=700, k=64, nlag=6, density=0.05), None, True),
         ('sparse long reach (unfused)', dict(n=500, T=1500, k=8, nlag=4, density=0.05), [1, 2, 24, 191], True),
         ('sparse k80 (generic)', dict(n=600, T=500, k=80, nlag=4, density=0.2), None, True)]
for name, c, lags, missing in cases:
    p = synth.sparse_problem(n=c['n'], T=c['T'], k=c['k'], nlag=c['nlag'], density=c['density'], dtype=np.float64, seed=41)
    if lags: p['lag_set'] = np.array(lags, dtype=np.uint32)
    m0 = synth.initial_model(p['Y'], p['lag_set'], c['k'], seed=41)
    for dtype in (np.float32, np.float64):
        model = make_model(m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
        with session.Session(p['Y'].astype(dtype), model, missing=missing, **synth.HYPER) as s:
            s.run(5); st = s.stats(5); s.download()
        print('%-30s %-8s %s  f %s  cg %s' % (name, np.dtype(dtype).name, dig(model.W, model.H, model.lag_val), repr(st[-1]['f']), [x['cg_iter'] for x in st]))
pd = synth.dense_problem(60, 1200, 6, [1, 2, 24, 168, 191], dtype=np.float64, seed=13)
m0 = synth.initial_model(pd['Y'], pd['lag_set'], 6, seed=7)
for dtype in (np.float32, np.float64):
    model = make_model(m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), pd['lag_set'])
    with session.Session(pd['Y'].astype(dtype), model, missing=False, **synth.HYPER) as s:
        s.run(5); st = s.stats(5); s.download()
    print('%-30s %-8s %s  f %s  cg %s' % ('dense full-observation', np.dtype(dtype).name, dig(model.W, model.H, model.lag_val), repr(st[-1]['f']), [x['cg_iter'] for x in st]))
