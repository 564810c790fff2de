"""Forecast accuracy summary of the rolling-window harness.

API contract (pinned by tests/test_python_frontend.py): a seven-field named tuple
``(nd, mase, nrmse, m_nd, m_mase, m_nrmse, mape)``, ``Metrics.default()`` = all 1e10,
``Metrics.generate(trueY, forecastY)``; the same quantities the reference reports
(python/trmf/trmf.py:266-301).

Every error measure here is a ratio "size of the forecast error / scale of the series".  It is evaluated
twice: pooled over the whole window x series block, and per series (column) and then averaged over the
series for which the ratio is finite (the ``m_`` fields).  MAPE is the mean relative error over the
entries whose true value is non-zero.  (Deliberate difference: the reference evaluates
``np.divide(|err|, |true|, where=true != 0)`` without an ``out`` array, trmf.py:300, so entries with a zero truth
contribute uninitialised memory to its mean; here they are left out, which is what the mask is for.  With no
zero truths the two agree exactly -- tests/test_python_frontend.py.)
"""
import collections

import numpy as np

_FIELDS = ('nd', 'mase', 'nrmse', 'm_nd', 'm_mase', 'm_nrmse', 'mape')


def _mean_of_finite(values):
    keep = values[np.isfinite(values)]
    assert keep.size > 0, 'no series with a finite score'
    return keep.mean()


# name -> (size of the error, scale of the truth); both take (array, axis) with axis None = pooled, 0 = per series
_MEASURES = {
    'nd': (lambda err, ax: np.abs(err).sum(axis=ax),
           lambda true, ax: np.abs(true).sum(axis=ax)),
    'nrmse': (lambda err, ax: np.sqrt(np.square(err).mean(axis=ax)),
              lambda true, ax: np.abs(true).mean(axis=ax)),
    'mase': (lambda err, ax: np.abs(err).mean(axis=ax),
             lambda true, ax: np.abs(np.diff(true, axis=0)).mean(axis=ax)),     # naive one-step forecast
}


class Metrics(collections.namedtuple('Metrics', _FIELDS)):
    __slots__ = ()

    def __str__(self):
        return ' '.join('{}={:.4g}'.format(name, value) for name, value in zip(self._fields, self))

    @classmethod
    def default(cls):
        """Worse than any real result: the start value of a search for the best setting."""
        return cls._make([1e10] * len(_FIELDS))

    @classmethod
    def generate(cls, trueY, forecastY, missing=True):
        truth = np.asarray(trueY)
        err = np.asarray(forecastY) - truth
        scores = {}
        with np.errstate(divide='ignore', invalid='ignore'):
            for name, (size, scale) in _MEASURES.items():
                scores[name] = size(err, None) / scale(truth, None)
                scores['m_' + name] = _mean_of_finite(size(err, 0) / scale(truth, 0))
            observed = truth != 0
            scores['mape'] = _mean_of_finite(np.abs(err[observed]) / np.abs(truth[observed]))
        return cls(**scores)
