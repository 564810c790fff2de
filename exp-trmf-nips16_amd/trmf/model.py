"""The TRMF model object of the Python front end (host side, pure NumPy -- no GPU code here).

API contract (names, signatures, defaults, array layouts, RNG call order, file names), pinned by
tests/test_python_frontend.py; the reference's counterpart is python/trmf/trmf.py:82-251:

* ``Model``: ``W`` (T x k, C order), ``H`` (n x k, C order), ``lag_val`` (|L| x k, F order) wrapped as
  ``PyMatrix`` views (``pyW``, ``pyH``, ``pylag_val``) that the C ABI updates in place; ``lag_set`` sorted uint32.
* ``Model.initialize`` draws ``rand`` W, ``rand`` H, ``randn`` lag_val, in that order, after ``np.random.seed(seed)``;
  a ``warm_start_model`` replaces them by the previous model rolled forward.
* ``Model.save`` / ``Model.load``: ``arrays.npz`` (W, H, lag_val, lag_set) + ``other.pkl`` ({'transform': ...}).
* ``NormalizedTransform``: per-series affine map ``y -> a*y + b`` to zero mean / unit variance.
"""
import os
import pickle

import numpy as np
import scipy.sparse as smat

from .rf_util import PyMatrix

_ARRAYS, _EXTRA = 'arrays.npz', 'other.pkl'


class NormalizedTransform(object):
    """z-score per series: ``preprocess`` applies ``a*y + b`` with a = 1/std, b = -mean/std; ``postprocess`` undoes it.
    Constant series (std = 0) keep a unit scale."""

    def __init__(self, Y):
        dense = Y.toarray() if smat.issparse(Y) else np.asarray(Y)
        center = dense.mean(axis=0, keepdims=True)
        spread = dense.std(axis=0, keepdims=True)
        spread = np.where(spread == 0, 1.0, spread)
        self.a = 1.0 / spread
        self.b = -self.a * center

    def _check(self, Y):
        assert Y.shape[1] == self.a.shape[1], 'series count differs from the fitted transform'

    def preprocess(self, Y):
        self._check(Y)
        return Y * self.a + self.b

    def postprocess(self, Y):
        self._check(Y)
        return (Y - self.b) / self.a


def _sorted_lags(lag_set):
    return np.array(sorted(lag_set), dtype=np.uint32)


def _empty_factors(m, n, k, nlag, dtype):
    """Zero-filled (W, H, lag_val) in the memory orders the C ABI requires (trmf.cpp:583-594)."""
    return (np.zeros((m, k), dtype=dtype, order='C'), np.zeros((n, k), dtype=dtype, order='C'),
            np.zeros((nlag, k), dtype=dtype, order='F'))


def _ar_rollout(W, first, last, lag_set, lag_val):
    """In place: W[i] = sum_l lag_val[l] * W[i - lag_set[l]] for i = first .. last-1, in time order."""
    back = lag_set.astype(np.int64)
    for i in range(first, last):
        np.sum(W[i - back] * lag_val, axis=0, out=W[i])


class Model(object):

    def __init__(self, pyW=None, pyH=None, pylag_val=None, lag_set=None, transform=None):
        self.pyW, self.pyH, self.pylag_val = pyW, pyH, pylag_val
        self.lag_set = lag_set
        self.transform = transform

    @classmethod
    def from_arrays(cls, W, H, lag_val, lag_set, transform=None, dtype=None):
        dtype = W.dtype if dtype is None else dtype
        return cls(PyMatrix(np.ascontiguousarray(W), dtype), PyMatrix(np.ascontiguousarray(H), dtype),
                   PyMatrix(np.asfortranarray(lag_val), dtype), lag_set=lag_set, transform=transform)

    # the arrays the C side reads and writes are the ones pinned inside the PyMatrix views
    @property
    def W(self):
        return self.pyW.py_buf['val']

    @property
    def H(self):
        return self.pyH.py_buf['val']

    @property
    def lag_val(self):
        return self.pylag_val.py_buf['val']

    @property
    def m(self):
        return self.W.shape[0]

    @property
    def n(self):
        return self.H.shape[0]

    @property
    def k(self):
        return self.W.shape[1]

    # ---- persistence ---------------------------------------------------------------------------
    def save(self, path_to_folder):
        os.makedirs(path_to_folder, exist_ok=True)
        np.savez(os.path.join(path_to_folder, _ARRAYS), W=self.W, H=self.H, lag_val=self.lag_val, lag_set=self.lag_set)
        with open(os.path.join(path_to_folder, _EXTRA), 'wb') as fh:
            pickle.dump({'transform': self.transform}, fh)

    @classmethod
    def load(cls, path_to_folder, dtype=None):
        assert os.path.isdir(path_to_folder), path_to_folder
        with np.load(os.path.join(path_to_folder, _ARRAYS)) as stored:
            parts = {name: stored[name] for name in ('W', 'H', 'lag_val', 'lag_set')}
        with open(os.path.join(path_to_folder, _EXTRA), 'rb') as fh:
            extra = pickle.load(fh)
        return cls.from_arrays(parts['W'], parts['H'], parts['lag_val'], parts['lag_set'],
                               transform=extra['transform'], dtype=dtype)

    # ---- forecasting -----------------------------------------------------------------------------
    def latent_forecast(self, window, Wnew=None):
        """W extended by ``window`` timestamps through the AR model; the first ``m`` rows are W itself."""
        total = self.m + window
        if Wnew is None:
            Wnew = np.zeros((total, self.k), dtype=self.W.dtype, order='C')
        assert Wnew.shape == (total, self.k) and Wnew.dtype == self.W.dtype and Wnew.flags.c_contiguous
        Wnew[:self.m] = self.W
        _ar_rollout(Wnew, self.m, total, self.lag_set, self.lag_val)
        return Wnew

    def forecast(self, window, Ynew=None, threshold=None):
        """(Ynew, Wnew): the next ``window`` rows of Y and of W; values below ``threshold`` are clipped before the
        transform (if any) is undone."""
        Wnew = self.latent_forecast(window)[self.m:]
        if Ynew is None:
            Ynew = np.zeros((window, self.n), dtype=self.W.dtype, order='C')
        Ynew[:] = Wnew.dot(self.H.T)
        if threshold is not None:
            np.maximum(Ynew, threshold, out=Ynew)
        if self.transform is not None:
            Ynew[:] = self.transform.postprocess(Ynew)
        return Ynew, Wnew

    # ---- construction ------------------------------------------------------------------------------
    @staticmethod
    def syn_gen(m, n, k, lag_set, seed=None, noise=0.01, dtype=np.float32):
        """A low-rank matrix whose temporal factor follows an AR model over ``lag_set``.  Random draws in this order:
        W, H, lag_val (standard normal), then the innovation noise of rows midx.. of W."""
        if seed is not None:
            np.random.seed(seed)
        lags = _sorted_lags(lag_set)
        W, H, theta = _empty_factors(m, n, k, len(lags), dtype)
        for target in (W, H, theta):
            target[:] = np.random.randn(*target.shape)
        theta = theta.dot(np.diag(1.0 / (np.abs(theta).sum(axis=0) + 0.1)))      # contractive columns
        start = int(lags.max())
        _ar_rollout(W, start, m, lags, theta)
        W[start:] += noise * np.random.randn(m - start, k)
        Y = np.zeros((m, n), dtype=dtype, order='C')
        Y[:] = W.dot(H.T)
        return {'W': W, 'H': H, 'lag_val': theta, 'lag_set': lags, 'Y': Y, 'k': k}

    @classmethod
    def initialize(cls, Y, lag_set, k, warm_start_model=None, seed=None, dtype=None, transform=None):
        if seed is not None:
            np.random.seed(seed)
        dtype = Y.dtype if dtype is None else dtype
        m, n = Y.shape
        lags = _sorted_lags(lag_set)
        W, H, theta = _empty_factors(m, n, k, len(lags), dtype)
        W[:] = np.random.rand(m, k)
        H[:] = np.random.rand(n, k)
        theta[:] = np.random.randn(len(lags), k)
        prev = warm_start_model
        if prev is not None:
            assert (prev.k, prev.n) == (k, n) and prev.m <= m and len(prev.lag_set) == len(lags)
            W[:] = prev.latent_forecast(m - prev.m)
            H[:] = prev.H
            theta[:] = prev.lag_val
            transform = prev.transform
        if transform is not None:                       # any request for a transform: fit a fresh one on this Y
            transform = NormalizedTransform(Y)
        return cls(PyMatrix(W, dtype), PyMatrix(H, dtype), PyMatrix(theta, dtype), lag_set=lags, transform=transform)

    def fit(self, Y, **kw):
        """``model.fit(Y, ...)`` is ``train(Y, model, ...)``."""
        from .trmf import train
        return train(Y, self, **kw)
