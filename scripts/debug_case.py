import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,'exp-trmf-nips16_amd')); sys.path.insert(0, os.path.join(ROOT,'oracle')); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, oracle_py as O
from trmf import synth, session
from helpers import make_model, relfro
dtype=np.float32; k=64; nlag=32
p = synth.sparse_problem(n=1500, T=700, k=k, nlag=nlag, density=0.05, dtype=dtype, seed=7)
m0 = synth.initial_model(p['Y'], p['lag_set'], k, seed=7)
for iters in (1,2,3,4):
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    log = O.train_port(p['Y'], p['lag_set'], W, H, Th, synth.HYPER, max_iter=iters)
    model = make_model(m0.W, m0.H, m0.lag_val, p['lag_set'])
    with session.Session(p['Y'], model, missing=True, **synth.HYPER) as s:
        s.run(iters); st=s.stats(iters); s.download()
    print(iters, 'cg oracle', [l['cg_iter'] for l in log], 'gpu', [x['cg_iter'] for x in st],
          'relW %.2e relH %.2e relTh %.2e'%(relfro(model.W,W), relfro(model.H,H), relfro(model.lag_val,Th)),
          'rn oracle %.4e gpu %.4e'%(log[-1]['cg_rnorm'], st[-1]['cg_rnorm']), 'cgtol', 0.1*st[-1]['gnorm'])
    Jo = O.objective(p['Y'], p['lag_set'], W, H, Th, synth.HYPER); Jp = O.objective(p['Y'], p['lag_set'], model.W, model.H, model.lag_val, synth.HYPER)
    print('   J', Jo, Jp, abs(Jo-Jp)/Jo)
# same in float64 to see the "true" trajectory
p64 = synth.sparse_problem(n=1500, T=700, k=k, nlag=nlag, density=0.05, dtype=np.float64, seed=7)
m64 = synth.initial_model(p64['Y'], p64['lag_set'], k, seed=7)
W, H, Th = m64.W.copy(), m64.H.copy(), np.asfortranarray(m64.lag_val.copy())
log = O.train_port(p64['Y'], p64['lag_set'], W, H, Th, synth.HYPER, max_iter=4)
print('f64 oracle cg', [l['cg_iter'] for l in log])
