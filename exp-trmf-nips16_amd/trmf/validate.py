"""Rolling-window validation and grid search on top of train() (reference trmf.py:303-346)."""
import itertools
import pickle

import numpy as np
import scipy.sparse as smat

from .metrics import Metrics
from .model import Model


def rolling_validate(Y, lag_set, k=40, window_size=24, nr_windows=7, lambdaI=0.5, lambdaAR=50, lambdaLag=0.5,
                     max_iter=20, missing=True, threshold=0, transform=None, threads=16, verbose=0, seed=0):
    """Train on a growing prefix, forecast the next window, warm-start the next fit (trmf.py:303-329)."""
    T, n = Y.shape
    assert T > nr_windows * window_size
    trueY = Y[-(nr_windows * window_size):, :]
    forecastY = np.zeros((nr_windows * window_size, n), dtype=Y.dtype, order='C')
    prev_model = None
    for i in range(nr_windows):
        trn_end = T - (nr_windows - i) * window_size
        Y_trn = Y[0:trn_end, :]
        if missing:
            Y_trn = smat.csr_matrix(Y_trn)
        model = Model.initialize(Y_trn, lag_set, k, seed=seed, warm_start_model=prev_model, transform=transform)
        from .trmf import train
        model = train(Y_trn, model, lambdaI=lambdaI, lambdaAR=lambdaAR, lambdaLag=lambdaLag,
                      max_iter=max_iter, missing=missing, threads=threads, verbose=verbose)
        model.forecast(window_size, Ynew=forecastY[i * window_size:(i + 1) * window_size, :], threshold=threshold)
        prev_model = model
    return Metrics.generate(trueY, forecastY, missing=missing)


def grid_search(Y, lag_set, grid_params, pkl_file=None, **kw_args):
    """Exhaustive grid over rolling_validate keyword arguments (trmf.py:331-346)."""
    results = []
    best = Metrics.default()
    keys = list(grid_params.keys())
    for values in itertools.product(*[grid_params[key] for key in keys]):
        kws = dict(kw_args)
        kws.update(zip(keys, values))
        metrics = rolling_validate(Y, lag_set, **kws)
        results.append({'kws': kws, 'metrics': metrics})
        if metrics.m_nd < best.m_nd:
            best = metrics
            print(metrics, dict(zip(keys, values)))
        if pkl_file is not None:
            with open(pkl_file, 'wb') as fh:
                pickle.dump(results, fh)
    return results, best
