"""Forecast accuracy metrics (reference python/trmf/trmf.py:266-301)."""
import collections

import numpy as np


class Metrics(collections.namedtuple('Metrics', ['nd', 'mase', 'nrmse', 'm_nd', 'm_mase', 'm_nrmse', 'mape'])):
    """Forecast accuracy summary (trmf.py:266-301)."""
    __slots__ = ()

    def __str__(self):
        return ' '.join('{}={:.4g}'.format(key, getattr(self, key)) for key in self._fields)

    @classmethod
    def default(cls):
        return cls(*([1e10] * 7))

    @classmethod
    def generate(cls, trueY, forecastY, missing=True):
        nz_mask = trueY != 0
        diff = forecastY - trueY
        abs_true = np.absolute(trueY)
        abs_diff = np.absolute(diff)

        def finite_mean(x):
            x = x[np.isfinite(x)]
            assert len(x) != 0
            return x.mean()

        with np.errstate(divide='ignore', invalid='ignore'):
            nrmse = np.sqrt((diff ** 2).mean()) / abs_true.mean()
            m_nrmse = finite_mean(np.sqrt((diff ** 2).mean(axis=0)) / abs_true.mean(axis=0))
            nd = abs_diff.sum() / abs_true.sum()
            m_nd = finite_mean(abs_diff.sum(axis=0) / abs_true.sum(axis=0))
            baseline = np.absolute(trueY[1:, :] - trueY[:-1, :])
            mase = abs_diff.mean() / baseline.mean()
            m_mase = finite_mean(abs_diff.mean(axis=0) / baseline.mean(axis=0))
            ratio = np.full(abs_diff.shape, np.nan, dtype=np.float64)
            np.divide(abs_diff, abs_true, out=ratio, where=nz_mask)
            mape = finite_mean(ratio)
        return cls(nd=nd, mase=mase, nrmse=nrmse, m_nd=m_nd, m_mase=m_mase, m_nrmse=m_nrmse, mape=mape)


