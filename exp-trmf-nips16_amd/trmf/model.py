"""Model container, normalisation transform and forecasting helpers (pure NumPy, no GPU).

Counterparts in the reference: NormalizedTransform python/trmf/trmf.py:82-96, Model :98-251.
"""
import os
import pickle
from os import path

import numpy as np
import scipy.sparse as smat

from .rf_util import PyMatrix


class NormalizedTransform(object):
    """Column-wise affine map to zero mean / unit variance (reference trmf.py:82-96)."""

    def __init__(self, Y):
        Yd = Y.toarray() if smat.issparse(Y) else np.asarray(Y)
        mean = np.asarray(Yd.mean(axis=0)).reshape(1, -1)
        std = np.asarray(Yd.std(axis=0)).reshape(1, -1)
        std[std == 0] = 1.0
        self.a = 1. / std
        self.b = -self.a * mean

    def preprocess(self, Y):
        assert Y.shape[1] == self.a.shape[1]
        return Y * self.a + self.b

    def postprocess(self, Y):
        assert Y.shape[1] == self.a.shape[1]
        return (Y - self.b) / self.a


class Model(object):
    """W (T x k, C order), H (n x k, C order), lag_val (|L| x k, F order), sorted uint32 lag_set."""

    def __init__(self, pyW=None, pyH=None, pylag_val=None, lag_set=None, transform=None):
        self.pyW = pyW
        self.pyH = pyH
        self.pylag_val = pylag_val
        self.lag_set = lag_set
        self.transform = transform

    k = property(lambda self: self.W.shape[1])
    m = property(lambda self: self.W.shape[0])
    n = property(lambda self: self.H.shape[0])
    W = property(lambda self: self.pyW.py_buf['val'])
    H = property(lambda self: self.pyH.py_buf['val'])
    lag_val = property(lambda self: self.pylag_val.py_buf['val'])

    # -- persistence: arrays.npz + other.pkl, same file names as the reference (trmf.py:131-168)
    @classmethod
    def load(cls, path_to_folder, dtype=None):
        assert path.isdir(path_to_folder)
        with open(path.join(path_to_folder, 'other.pkl'), 'rb') as fh:
            transform = pickle.load(fh)['transform']
        with np.load(path.join(path_to_folder, 'arrays.npz')) as npz:
            W, H, lag_val, lag_set = npz['W'], npz['H'], npz['lag_val'], npz['lag_set']
        if dtype is None:
            dtype = W.dtype
        return cls(pyW=PyMatrix(np.ascontiguousarray(W), dtype), pyH=PyMatrix(np.ascontiguousarray(H), dtype),
                   pylag_val=PyMatrix(np.asfortranarray(lag_val), dtype), lag_set=lag_set,
                   transform=transform)

    def save(self, path_to_folder):
        if not path.exists(path_to_folder):
            os.makedirs(path_to_folder)
        assert path.isdir(path_to_folder)
        with open(path.join(path_to_folder, 'arrays.npz'), 'wb') as fh:
            np.savez(fh, W=self.W, H=self.H, lag_val=self.lag_val, lag_set=self.lag_set)
        with open(path.join(path_to_folder, 'other.pkl'), 'wb') as fh:
            pickle.dump({'transform': self.transform}, fh)

    # -- forecasting (trmf.py:170-193)
    def latent_forecast(self, window, Wnew=None):
        if Wnew is None:
            Wnew = np.zeros((self.m + window, self.k), dtype=self.W.dtype, order='C')
        else:
            assert Wnew.shape == (self.m + window, self.k)
            assert Wnew.dtype == self.W.dtype and Wnew.flags['C_CONTIGUOUS']
        Wnew[:self.m, :] = self.W
        lags = self.lag_set.astype(np.int64)
        for i in range(self.m, self.m + window):
            Wnew[i, :] = (Wnew[i - lags, :] * self.lag_val).sum(axis=0)
        return Wnew

    def forecast(self, window, Ynew=None, threshold=None):
        Wnew = self.latent_forecast(window)[self.m:, :]
        if Ynew is None:
            Ynew = np.zeros((window, self.n), dtype=self.W.dtype, order='C')
        Ynew[:] = Wnew.dot(self.H.T)
        if threshold is not None:
            Ynew[Ynew < threshold] = threshold
        if self.transform is not None:
            Ynew[:] = self.transform.postprocess(Ynew)
        return Ynew, Wnew

    # -- synthetic data (trmf.py:195-220)
    @staticmethod
    def syn_gen(m, n, k, lag_set, seed=None, noise=0.01, dtype=np.float32):
        if seed is not None:
            np.random.seed(seed)
        lag_set = np.array(sorted(lag_set), dtype=np.uint32)
        midx = int(lag_set.max())
        W = np.zeros((m, k), dtype=dtype, order='C')
        H = np.zeros((n, k), dtype=dtype, order='C')
        lag_val = np.zeros((len(lag_set), k), dtype=dtype, order='F')
        W[:] = np.random.randn(m, k)
        H[:] = np.random.randn(n, k)
        lag_val[:] = np.random.randn(len(lag_set), k)
        lag_val = lag_val.dot(np.diag(1. / (np.absolute(lag_val).sum(axis=0) + 0.1)))
        lags = lag_set.astype(np.int64)
        for i in range(midx, m):
            W[i, :] = (W[i - lags, :] * lag_val).sum(axis=0)
        W[midx:, :] += noise * np.random.randn(m - midx, k)
        Y = np.zeros((m, n), dtype=dtype, order='C')
        Y[:] = W.dot(H.T)
        return {'W': W, 'H': H, 'lag_val': lag_val, 'lag_set': lag_set, 'Y': Y, 'k': k}

    # -- random start / warm start (trmf.py:222-251)
    @classmethod
    def initialize(cls, Y, lag_set, k, warm_start_model=None, seed=None, dtype=None, transform=None):
        if seed is not None:
            np.random.seed(seed)
        if dtype is None:
            dtype = Y.dtype
        m, n = Y.shape
        lag_set = np.array(sorted(lag_set), dtype=np.uint32)
        W = np.zeros((m, k), dtype=dtype, order='C')
        H = np.zeros((n, k), dtype=dtype, order='C')
        lag_val = np.zeros((len(lag_set), k), dtype=dtype, order='F')
        W[:] = np.random.rand(m, k)
        H[:] = np.random.rand(n, k)
        lag_val[:] = np.random.randn(len(lag_set), k)
        if warm_start_model is not None:
            prev = warm_start_model
            assert prev.k == k and prev.n == n and prev.m <= m
            assert len(lag_set) == len(prev.lag_set)
            W[:] = prev.latent_forecast(m - prev.m)
            H[:] = prev.H
            lag_val[:] = prev.lag_val
            transform = prev.transform
        if transform is not None:          # any truthy/non-None request -> fit a fresh transform on Y
            transform = NormalizedTransform(Y)
        return cls(pyW=PyMatrix(W, dtype), pyH=PyMatrix(H, dtype), pylag_val=PyMatrix(lag_val, dtype),
                   lag_set=lag_set, transform=transform)

    def fit(self, Y, **kw):
        """Convenience: ``model.fit(Y, ...)`` == ``train(Y, model, ...)``."""
        from .trmf import train
        return train(Y, self, **kw)
