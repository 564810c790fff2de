"""ctypes binding of trmf_float32.so / trmf_float64.so (the GPU libraries) and dtype dispatch.

Counterpart of the ``corelib`` class of the reference (python/trmf/trmf.py:19-80).
"""
from ctypes import POINTER, byref, c_double, c_int32, c_uint32
from os import path

import numpy as np

from .rf_util import PyMatrix, fillprototype, load_dynamic_library


class corelib(object):
    """ctypes binding of the two element-type libraries (reference trmf.py:19-75)."""

    def __init__(self, dirname, soname, forced_rebuild=False):
        self.clib_float32 = load_dynamic_library(dirname, soname + '_float32', forced_rebuild=forced_rebuild)
        self.clib_float64 = load_dynamic_library(dirname, soname + '_float64', forced_rebuild=forced_rebuild)
        arg_list = [
            POINTER(PyMatrix),  # Y
            POINTER(c_uint32),  # lag_set
            c_uint32,           # lag_size
            POINTER(PyMatrix),  # W
            POINTER(PyMatrix),  # H
            POINTER(PyMatrix),  # lag_val
            c_int32,            # warm_start
            c_double,           # lambdaI
            c_double,           # lambdaAR
            c_double,           # lambdaLag
            c_int32,            # max_iter
            c_int32,            # period_W
            c_int32,            # period_H
            c_int32,            # period_Lag
            c_int32,            # threads
            c_int32,            # missing
            c_int32,            # verbose
        ]
        fillprototype(self.clib_float32.c_trmf_train, None, arg_list)
        fillprototype(self.clib_float64.c_trmf_train, None, arg_list)

    def lib_for(self, dtype):
        return self.clib_float64 if np.dtype(dtype) == np.float64 else self.clib_float32

    def train(self, pyY, lag_set, pyW, pyH, pylag_val, warm_start=True,
              lambdaI=0.1, lambdaAR=0.1, lambdaLag=0.1, max_iter=10,
              period_W=1, period_H=1, period_Lag=2, threads=1, missing=False, verbose=0):
        clib = self.lib_for(pyY.dtype)
        if verbose != 0:
            print('perform float64 computation' if clib is self.clib_float64
                  else 'perform float32 computation')
        clib.c_trmf_train(
            byref(pyY), lag_set.ctypes.data_as(POINTER(c_uint32)), len(lag_set),
            byref(pyW), byref(pyH), byref(pylag_val), c_int32(int(warm_start)),
            lambdaI, lambdaAR, lambdaLag, max_iter, period_W, period_H, period_Lag,
            threads, int(missing), verbose)


# TRMF_CORELIB_DIR: load the library pair from another directory (the sanitizer build of `make asan`)
import os as _os
corelib_path = _os.environ.get('TRMF_CORELIB_DIR') or path.join(path.dirname(path.abspath(__file__)), 'corelib/')
soname = 'trmf'
_clib = None


def get_clib(forced_rebuild=False):
    """The library pair is loaded on first use (the reference loads at import time; deferring
    keeps ``import trmf`` usable for the pure-Python helpers on a machine without the build)."""
    global _clib
    if _clib is None or forced_rebuild:
        _clib = corelib(corelib_path, soname, forced_rebuild=forced_rebuild)
    return _clib


