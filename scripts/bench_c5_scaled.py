#!/usr/bin/env python3
"""Config-5-like run scaled to one quick call: fp64, k=64, |L|=32 (BASELINE config 5 is 1M x 50k; here n x T is
passed on the command line, optionally the density).  Prints per-phase times of the last iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))
import numpy as np
from trmf import session, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
k, nlag = 64, 32
dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
p = synth.sparse_problem(n, T, k, nlag, dens, dtype=np.float64, seed=0)
m = synth.initial_model(p['Y'], p['lag_set'], k, seed=0)
with session.Session(p['Y'], m, missing=True, log_norms=False, **synth.HYPER) as s:
    s.run(6); st = s.stats(4); J = s.objective()
print('fp64 n=%d T=%d k=%d |L|=%d nnz=%d: F %.3f ms (kernel %.3f)  X %.3f ms  Theta %.3f ms  cg %s  J %.6e' % (
    n, T, k, nlag, p['Y'].nnz, np.mean([x['ms_F'] for x in st]), np.mean([x['ms_F_kernel'] for x in st]),
    np.mean([x['ms_X'] for x in st]), np.mean([x['ms_LV'] for x in st]), [x['cg_iter'] for x in st], J))
