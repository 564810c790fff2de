"""Rolling-window evaluation and hyper-parameter grid of the Python front end.

Contract (same call signatures and results as the reference's python/trmf/trmf.py:303-346): the last
``nr_windows * window_size`` timestamps of ``Y`` are forecast window by window; for window ``i`` the model is
trained on the prefix that ends where the window starts, warm-started from the previous window's model rolled
forward by the AR recursion, and its ``window_size``-step forecast is stored.  The result is ``Metrics`` of all
forecasts against the truth.

What is different here is where the data lives.  The reference re-wraps the whole growing prefix for every
window; on the GPU that would be a fresh upload of everything per window.  ``rolling_validate`` instead keeps ONE
HBM-resident session for all windows (``trmf.session.Session``): the first prefix is uploaded once, each later
window only appends its new timestamps (``Session.append_rows`` -> ``trmf_session_append_rows``: CSR rows
appended, CSC rebuilt on the device, W extended on the device by the same AR recursion, H and the lag weights
stay where they are).  A per-window ``NormalizedTransform`` (the paper scripts' ``transform=True``) rescales
every entry of the prefix: the session then keeps the RAW dense matrix and applies each window's refitted
coefficients on the device (``Session.set_transform`` -> ``trmf_session_set_series_transform``), so only 2n numbers
per window are uploaded.  ``resident=False`` forces a fresh upload per window through ``train``.
"""
import itertools
import pickle

import numpy as np
import scipy.sparse as smat

from .metrics import Metrics
from .model import Model


def _as_training_matrix(block, missing):
    """Observed-entries training takes a sparse matrix whose stored entries are the non-zeros of the block."""
    return smat.csr_matrix(block) if missing else block


def _train_windows_resident(Y, lag_set, k, cuts, seed, hyper, max_iter, missing, transform, verbose):
    """Yield the trained model of every window from one resident session.  With a transform (dense Y, full
    observation) the session holds the RAW matrix and applies each window's refitted coefficients on the device."""
    from .session import Session
    model = Model.initialize(Y[:cuts[0]], lag_set, k, seed=seed, transform=transform)
    with Session(_as_training_matrix(Y[:cuts[0]], missing), model, missing=missing, verbose=verbose,
                 log_norms=bool(verbose), timing=0, **hyper) as sess:      # (nobody reads per-phase times here: no phase events)
        if model.transform is not None:
            sess.set_transform(model.transform)
        sess.run(max_iter).download()
        yield model
        for prev_cut, cut in zip(cuts[:-1], cuts[1:]):
            # host-side model of the new size: same warm start the device applies, same RNG consumption as the
            # reference, and (if asked for) a transform refitted on the grown prefix
            model = Model.initialize(Y[:cut], lag_set, k, seed=seed, warm_start_model=model, transform=transform)
            sess.append_rows(_as_training_matrix(Y[prev_cut:cut], missing))
            if model.transform is not None:
                sess.set_transform(model.transform)
            sess.model = model
            sess.run(max_iter).download()
            yield model


def _train_windows_fresh(Y, lag_set, k, cuts, seed, hyper, max_iter, missing, transform, threads, verbose):
    """Yield the trained model of every window, each from its own upload (needed with a per-window transform)."""
    from .trmf import train
    model = None
    for cut in cuts:
        prefix = _as_training_matrix(Y[:cut], missing)
        model = Model.initialize(prefix, lag_set, k, seed=seed, warm_start_model=model, transform=transform)
        train(prefix, model, max_iter=max_iter, missing=missing, threads=threads, verbose=verbose, **hyper)
        yield model


def rolling_validate(Y, lag_set, k=40, window_size=24, nr_windows=7, lambdaI=0.5, lambdaAR=50, lambdaLag=0.5,
                     max_iter=20, missing=True, threshold=0, transform=None, threads=16, verbose=0, seed=0,
                     resident=True):
    T, n = Y.shape
    horizon = nr_windows * window_size
    assert T > horizon, 'series too short for {} windows of {}'.format(nr_windows, window_size)
    cuts = [T - horizon + i * window_size for i in range(nr_windows)]        # training prefix of window i = Y[:cuts[i]]
    hyper = dict(lambdaI=lambdaI, lambdaAR=lambdaAR, lambdaLag=lambdaLag)
    # resident: a NumPy Y, and a transform only where the device can apply it (dense full-observation training)
    if resident and isinstance(Y, np.ndarray) and (transform is None or (not missing and Y.dtype in (np.float32, np.float64))):
        models = _train_windows_resident(Y, lag_set, k, cuts, seed, hyper, max_iter, missing, transform, verbose)
    else:
        models = _train_windows_fresh(Y, lag_set, k, cuts, seed, hyper, max_iter, missing, transform, threads, verbose)
    forecasts = np.zeros((horizon, n), dtype=Y.dtype, order='C')
    for i, model in enumerate(models):
        model.forecast(window_size, Ynew=forecasts[i * window_size:(i + 1) * window_size], threshold=threshold)
    return Metrics.generate(Y[T - horizon:], forecasts, missing=missing)


def _grid_points(grid_params):
    names = list(grid_params)
    for combo in itertools.product(*(grid_params[name] for name in names)):
        yield dict(zip(names, combo))


def grid_search(Y, lag_set, grid_params, pkl_file=None, **kw_args):
    """Every combination of ``grid_params`` through ``rolling_validate``; returns (all results, best by m_nd).
    Each improvement is printed; with ``pkl_file`` the result list is re-written after every combination."""
    results, best = [], Metrics.default()
    for point in _grid_points(grid_params):
        settings = dict(kw_args, **point)
        score = rolling_validate(Y, lag_set, **settings)
        results.append({'kws': settings, 'metrics': score})
        if score.m_nd < best.m_nd:
            best = score
            print(score, point)
        if pkl_file is not None:
            with open(pkl_file, 'wb') as fh:
                pickle.dump(results, fh)
    return results, best
