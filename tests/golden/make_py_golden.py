#!/usr/bin/env python3
"""Golden vectors of the reference's PYTHON harness (python/trmf/trmf.py:82-346), captured by importing the
real package in the build container.  Writes tests/golden/py_harness.npz (data only: inputs and the
reference's outputs); tests/test_python_frontend.py replays them through this repo's own front end.

The reference package does not import under the installed SciPy / NumPy as it stands: it is
copied to a scratch directory OUTSIDE the repo, the two oracle/_ref libraries are dropped into its corelib/,
and the NumPy names it expects on the `scipy` module are aliased before import.  Nothing of it is stored here.

    python tests/golden/make_py_golden.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import scipy

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_PKG = '/root/reference/python/trmf'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'py_harness.npz')


def import_reference():
    scratch = tempfile.mkdtemp(prefix='trmf_ref_')
    dst = os.path.join(scratch, 'trmf')
    shutil.copytree(REF_PKG, dst)
    for name in ('trmf_float32.so', 'trmf_float64.so'):
        shutil.copy(os.path.join(ROOT, 'oracle', '_ref', name), os.path.join(dst, 'corelib', name))
    for name in dir(np):                                   # `import scipy as sp; sp.zeros(...)` style of the reference
        if not name.startswith('_') and not hasattr(scipy, name):
            setattr(scipy, name, getattr(np, name))
    scipy.random, scipy.rand, scipy.randn = np.random, np.random.rand, np.random.randn
    scipy.absolute, scipy.float32, scipy.float64 = np.absolute, np.float32, np.float64
    os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
    sys.path.insert(0, scratch)
    import trmf as ref
    return ref, scratch


def main():
    ref, scratch = import_reference()
    out = {}
    # --- Metrics.generate on a block with zeros, a constant series and a zero series
    rng = np.random.RandomState(5)
    true = np.abs(rng.randn(30, 7)) * [1, 2, 3, 4, 5, 0, 7] + [0, 0, 0, 0, 0, 0, 3]
    true[rng.rand(30, 7) < 0.1] = 0.0
    pred = true + 0.3 * rng.randn(30, 7)
    m = ref.Metrics.generate(true, pred)
    out['met_true'], out['met_pred'] = true, pred
    out['met_values'] = np.array([getattr(m, f) for f in m._fields]); out['met_fields'] = np.array(m._fields)
    # the same block with every true value non-zero: the reference's MAPE is only well defined then (it evaluates
    # np.divide(..., where=mask) without `out=`, so entries with a zero truth hold uninitialised memory)
    true2 = np.where(true == 0, 0.25, true)
    m2 = ref.Metrics.generate(true2, pred)
    out['met2_true'] = true2; out['met2_values'] = np.array([getattr(m2, f) for f in m2._fields])
    # --- syn_gen / initialize / latent_forecast / forecast / warm start, fp32 and fp64
    for tag, dt in (('f32', np.float32), ('f64', np.float64)):
        d = ref.Model.syn_gen(50, 9, 4, [1, 3, 7], seed=11, dtype=dt)
        for key in ('W', 'H', 'lag_val', 'lag_set', 'Y'):
            out['syn_%s_%s' % (tag, key)] = d[key]
        mod = ref.Model.initialize(d['Y'], [7, 1, 3], 4, seed=2)
        out['init_%s_W' % tag], out['init_%s_H' % tag], out['init_%s_lag_val' % tag] = mod.W.copy(), mod.H.copy(), mod.lag_val.copy()
        mod.W[:] = d['W']; mod.H[:] = d['H']; mod.lag_val[:] = d['lag_val']
        out['lat_%s' % tag] = mod.latent_forecast(6)
        Yn, Wn = mod.forecast(6, threshold=0.05)
        out['fc_%s_Y' % tag], out['fc_%s_W' % tag] = Yn, Wn
        Ybig = np.vstack([d['Y'], Yn]).astype(dt)
        warm = ref.Model.initialize(Ybig, [1, 3, 7], 4, seed=3, warm_start_model=mod)
        out['warm_%s_W' % tag] = warm.W.copy()
        # transform: fitted on Y, forecast is mapped back
        modt = ref.Model.initialize(d['Y'], [1, 3, 7], 4, seed=2, transform=True)
        out['tr_%s_a' % tag], out['tr_%s_b' % tag] = modt.transform.a, modt.transform.b
        out['tr_%s_pre' % tag] = modt.transform.preprocess(d['Y'])
        modt.W[:] = d['W']; modt.H[:] = d['H']; modt.lag_val[:] = d['lag_val']
        out['tr_%s_fc' % tag] = modt.forecast(3)[0]
    # --- rolling_validate + grid_search end to end through the reference's own CPU solver (fp64, dense, missing=0 and 1)
    d = ref.Model.syn_gen(160, 12, 3, [1, 2, 5], seed=4, dtype=np.float64)
    Y = np.abs(d['Y']) + 0.1
    out['rv_Y'] = Y
    for missing in (False, True):
        met = ref.rolling_validate(Y, [1, 2, 5], k=3, window_size=8, nr_windows=3, lambdaI=0.5, lambdaAR=50, lambdaLag=0.5,
                                   max_iter=4, missing=missing, threshold=0, threads=2, seed=0)
        out['rv_missing%d' % int(missing)] = np.array([getattr(met, f) for f in met._fields])
    met = ref.rolling_validate(Y, [1, 2, 5], k=3, window_size=8, nr_windows=3, lambdaI=0.5, lambdaAR=50, lambdaLag=0.5,
                               max_iter=4, missing=False, threshold=None, transform=True, threads=2, seed=0)
    out['rv_transform'] = np.array([getattr(met, f) for f in met._fields])
    results, best = ref.grid_search(Y, [1, 2, 5], {'lambdaI': [0.5, 5.0], 'lambdaAR': [5, 50]}, k=3, window_size=8, nr_windows=2,
                                    max_iter=3, missing=True, threshold=0, threads=2, seed=0)
    out['gs_best'] = np.array([getattr(best, f) for f in best._fields])
    out['gs_m_nd'] = np.array([r['metrics'].m_nd for r in results])
    out['gs_lambdaI'] = np.array([r['kws']['lambdaI'] for r in results]); out['gs_lambdaAR'] = np.array([r['kws']['lambdaAR'] for r in results])
    np.savez_compressed(OUT, **out)
    shutil.rmtree(scratch, ignore_errors=True)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
