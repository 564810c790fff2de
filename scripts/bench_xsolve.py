#!/usr/bin/env python3
"""X-phase time with and without the AR term (diagnostic for the fused Hv kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))
import numpy as np
from trmf import session, synth
cfg = synth.CONFIGS['c3']
p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=np.float32, seed=0)
for lamAR in (50.0, 0.0):
    m = synth.initial_model(p['Y'], p['lag_set'], cfg['k'], seed=0)
    with session.Session(p['Y'], m, missing=True, lambdaI=0.5, lambdaAR=lamAR, lambdaLag=0.5, period_Lag=10**6) as s:
        s.run(8); st = s.stats(6)
    print('lambdaAR=%g: X %.3f ms  F %.3f ms  cg %s' % (lamAR, np.mean([x['ms_X'] for x in st]), np.mean([x['ms_F'] for x in st]), [x['cg_iter'] for x in st]))
