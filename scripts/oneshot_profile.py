#!/usr/bin/env python3
"""Wall time of the reference's own entry point, c_trmf_train (one-shot: upload, train, download), at a bench config.
    python scripts/oneshot_profile.py [config] [max_iter] [calls]
Prints one line per call (the first call of a process also pays the runtime's initialisation and the code-object load)
and, when the library exports trmf_last_train_profile, the call's own split."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))
import numpy as np
import trmf, trmf.trmf
from trmf import synth, session
from trmf.rf_util import PyMatrix

cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c3'
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 10
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = synth.CONFIGS[cfgname]
dt = np.dtype(cfg['dtype'])
if cfg.get('dense'):
    prob = synth.dense_problem(cfg['n'], cfg['T'], cfg['k'], cfg['lags'], dtype=dt, seed=0); hyper = dict(cfg['hyper']); missing = False
else:
    prob = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=dt, seed=0); hyper = dict(synth.HYPER); missing = True
pyY = PyMatrix(prob['Y'], dtype=dt)
lib = session.lib_for(dt)
for c in range(calls):
    m = synth.initial_model(prob['Y'], prob['lag_set'], cfg['k'], seed=0)
    t0 = time.perf_counter()
    trmf.trmf.get_clib().train(pyY, m.lag_set, m.pyW, m.pyH, m.pylag_val, warm_start=True, max_iter=max_iter, missing=missing, **hyper)
    dtc = time.perf_counter() - t0
    line = {'call': c, 'config': cfgname, 'max_iter': max_iter, 'wall_ms': 1e3 * dtc, 'H_checksum': float(np.abs(m.H).sum())}
    if hasattr(session, 'train_profile'):
        line['profile'] = session.train_profile(dt)
    print(json.dumps(line)); sys.stdout.flush()
