"""Python front end of the MI355X TRMF solver.

Same public names and argument meaning as the reference's ``python/trmf/trmf.py`` (own code):

=====================  ===============================================  =====================
name                   what                                             reference
=====================  ===============================================  =====================
``corelib``            loads trmf_float32/64.so, dtype dispatch         trmf.py:19-75
``NormalizedTransform``  per-series z-score                             trmf.py:82-96
``Model``              factors + lag weights + forecasting helpers      trmf.py:98-251
``train``              one ALS run through ``c_trmf_train``             trmf.py:253-264
``Metrics``            ND / MASE / NRMSE / MAPE                         trmf.py:266-301
``rolling_validate``   rolling-window forecast evaluation               trmf.py:303-329
``grid_search``        hyper-parameter grid over rolling_validate       trmf.py:331-346
=====================  ===============================================  =====================

``fit`` is an alias of ``train`` (the reference has no ``fit``; BASELINE.json's north star names
it).  The compute runs on the GPU behind ``c_trmf_train``; there is no CPU implementation in this
package.
"""
from ._corelib import corelib, corelib_path, get_clib, soname   # noqa: F401
from .metrics import Metrics                                     # noqa: F401
from .model import Model, NormalizedTransform                    # noqa: F401
from .rf_util import PyMatrix
from .validate import grid_search, rolling_validate              # noqa: F401


def train(Y, model, lambdaI=0.1, lambdaAR=0.1, lambdaLag=0.1,
          max_iter=10, period_W=1, period_H=1, period_Lag=2,
          threads=1, missing=False, verbose=0):
    """Run ``max_iter`` ALS iterations on the GPU, updating ``model`` in place (trmf.py:253-264)."""
    if model.transform is not None:
        Y = model.transform.preprocess(Y)
    get_clib().train(PyMatrix(Y, dtype=model.W.dtype), model.lag_set,
                     model.pyW, model.pyH, model.pylag_val, warm_start=True,
                     lambdaI=lambdaI, lambdaAR=lambdaAR, lambdaLag=lambdaLag,
                     max_iter=max_iter, period_W=period_W, period_H=period_H, period_Lag=period_Lag,
                     threads=threads, missing=missing, verbose=verbose)
    return model


fit = train


