"""HBM-resident solver sessions: ctypes wrapper over section 2 of include/trmf_abi.h.

A session uploads Y and the factors once, runs ALS iterations asynchronously on the GPU and
exposes per-iteration statistics (norms, CG counts, HIP-event phase times).  ``trmf.train`` /
``c_trmf_train`` are the stateless one-shot form of the same loop.  No reference counterpart
(SURVEY.md 8(f) rank 4).
"""
import ctypes
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int32, c_uint32, c_uint64, c_void_p

import numpy as np

from ._corelib import get_clib
from .rf_util import PyMatrix


class TrmfIterStats(ctypes.Structure):
    _fields_ = [('normF', c_double), ('normX', c_double), ('normLV', c_double),
                ('f', c_double), ('fnew', c_double), ('actred', c_double), ('prered', c_double),
                ('gnorm', c_double), ('cg_rnorm', c_double),
                ('cg_iter', c_int32), ('accepted', c_int32),
                ('ms_F', c_float), ('ms_X', c_float), ('ms_LV', c_float), ('ms_F_kernel', c_float),
                ('delta', c_double), ('cg_rnorm_direct', c_double), ('ms_X_gram', c_float), ('reserved_', c_float)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class TrmfTrainProfile(ctypes.Structure):
    """Split of the last c_trmf_train call of this process (include/trmf_abi.h)."""
    _fields_ = [('total_s', c_double), ('setup_s', c_double), ('upload_s', c_double), ('compute_s', c_double),
                ('download_s', c_double), ('teardown_s', c_double), ('bytes_h2d', c_double), ('bytes_d2h', c_double),
                ('iters', c_int32), ('device_mallocs', c_int32), ('pool_reused', c_int32), ('failed', c_int32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


ALLGATHERV_FN = ctypes.CFUNCTYPE(c_int32, c_void_p, POINTER(c_uint64), c_int32, c_void_p)
_prototyped = set()


def bind(lib):
    """Attach argtypes/restypes of the build-owned entry points to a loaded library (idempotent)."""
    if id(lib) in _prototyped:
        return lib
    P = POINTER(PyMatrix)
    lib.trmf_sizeof_real.restype = c_int32
    lib.trmf_device_count.restype = c_int32
    lib.trmf_set_device.argtypes = [c_int32]; lib.trmf_set_device.restype = c_int32
    lib.trmf_last_error.restype = c_char_p
    lib.trmf_device_free_bytes.restype = ctypes.c_int64
    lib.trmf_last_train_profile.argtypes = [POINTER(TrmfTrainProfile)]; lib.trmf_last_train_profile.restype = c_int32
    lib.trmf_release_cached.restype = c_int32
    lib.trmf_session_create.argtypes = [P, POINTER(c_uint32), c_uint32, P, P, P, c_double, c_double, c_double,
                                        c_int32, c_int32, c_int32, c_int32, c_int32]
    lib.trmf_session_create.restype = c_void_p
    lib.trmf_session_run.argtypes = [c_void_p, c_int32]; lib.trmf_session_run.restype = c_int32
    lib.trmf_session_log_norms.argtypes = [c_void_p, c_int32]; lib.trmf_session_log_norms.restype = c_int32
    lib.trmf_session_set_timing.argtypes = [c_void_p, c_int32]; lib.trmf_session_set_timing.restype = c_int32
    lib.trmf_session_sync.argtypes = [c_void_p]; lib.trmf_session_sync.restype = c_int32
    if hasattr(lib, 'trmf_session_mark'):      # (absent from libraries built before round 6: scripts/ab_builds.sh loads those side by side)
        lib.trmf_session_mark.argtypes = [c_void_p]; lib.trmf_session_mark.restype = c_int32
        lib.trmf_session_rewind.argtypes = [c_void_p]; lib.trmf_session_rewind.restype = c_int32
    lib.trmf_session_append_rows.argtypes = [c_void_p, P]; lib.trmf_session_append_rows.restype = c_int32
    lib.trmf_session_rows.argtypes = [c_void_p]; lib.trmf_session_rows.restype = c_int32
    lib.trmf_session_set_series_transform.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.trmf_session_set_series_transform.restype = c_int32
    lib.trmf_session_download.argtypes = [c_void_p, P, P, P]; lib.trmf_session_download.restype = c_int32
    lib.trmf_session_stats.argtypes = [c_void_p, POINTER(TrmfIterStats), c_int32]
    lib.trmf_session_stats.restype = c_int32
    lib.trmf_session_objective.argtypes = [c_void_p]; lib.trmf_session_objective.restype = c_double
    lib.trmf_session_fsolve_bytes.argtypes = [c_void_p]; lib.trmf_session_fsolve_bytes.restype = c_double
    lib.trmf_session_describe.argtypes = [c_void_p, ctypes.c_char_p, c_int32]; lib.trmf_session_describe.restype = c_int32
    lib.trmf_session_destroy.argtypes = [c_void_p]; lib.trmf_session_destroy.restype = None
    lib.trmf_dist_get_unique_id.argtypes = [c_void_p]; lib.trmf_dist_get_unique_id.restype = c_int32
    lib.trmf_dist_init.argtypes = [c_int32, c_int32, c_void_p]; lib.trmf_dist_init.restype = c_int32
    lib.trmf_dist_init_callback.argtypes = [c_int32, c_int32, ALLGATHERV_FN, c_void_p]
    lib.trmf_dist_init_callback.restype = c_int32
    lib.trmf_dist_init_solo.argtypes = [c_int32, c_int32]; lib.trmf_dist_init_solo.restype = c_int32
    lib.trmf_dist_rank.restype = c_int32
    lib.trmf_dist_world.restype = c_int32
    lib.trmf_dist_finalize.restype = None
    lib.trmf_partition_by_nnz.argtypes = [c_uint64, POINTER(c_uint64), c_int32, POINTER(c_uint64)]
    lib.trmf_partition_by_nnz.restype = c_int32
    _prototyped.add(id(lib))
    return lib


def lib_for(dtype):
    return bind(get_clib().lib_for(dtype))


def train_profile(dtype):
    """Split of the last ``c_trmf_train`` call made through the library of ``dtype`` (None if there was none)."""
    prof = TrmfTrainProfile()
    return prof.as_dict() if lib_for(dtype).trmf_last_train_profile(byref(prof)) == 0 else None


class Session(object):
    """``Session(Y, model, **hyper).run(iters)``; the model's arrays are refreshed by ``download()``."""

    def __init__(self, Y, model, lambdaI=0.1, lambdaAR=0.1, lambdaLag=0.1,
                 period_W=1, period_H=1, period_Lag=2, missing=True, verbose=0, log_norms=True, timing=1):
        self.model = model
        self.lib = lib_for(model.W.dtype)
        self.pyY = Y if isinstance(Y, PyMatrix) else PyMatrix(Y, dtype=model.W.dtype)
        self.handle = self.lib.trmf_session_create(
            byref(self.pyY), model.lag_set.ctypes.data_as(POINTER(c_uint32)), len(model.lag_set),
            byref(model.pyW), byref(model.pyH), byref(model.pylag_val),
            lambdaI, lambdaAR, lambdaLag, period_W, period_H, period_Lag, int(missing), verbose)
        if not self.handle:
            raise RuntimeError('trmf_session_create failed: ' + self.lib.trmf_last_error().decode())
        if not log_norms:       # the reference computes the ||.||^2 log lines only under verbose
            self.lib.trmf_session_log_norms(self.handle, 0)
        if timing != 1:         # phase events (ms_* of stats()) on every `timing`-th iteration only / never (0): they cost ~25 us apiece
            self._check(self.lib.trmf_session_set_timing(self.handle, int(timing)), 'trmf_session_set_timing')

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError('{} failed: {}'.format(what, self.lib.trmf_last_error().decode()))
        return rc

    def run(self, iters=1):
        self._check(self.lib.trmf_session_run(self.handle, iters), 'trmf_session_run')
        return self

    def sync(self):
        self._check(self.lib.trmf_session_sync(self.handle), 'trmf_session_sync')
        return self

    def mark(self):
        """Checkpoint (W, H, lag_val, iteration counter) on the device; ``rewind()`` returns to it."""
        self._check(self.lib.trmf_session_mark(self.handle), 'trmf_session_mark')
        return self

    def rewind(self):
        self._check(self.lib.trmf_session_rewind(self.handle), 'trmf_session_rewind')
        return self

    def append_rows(self, Ynew):
        """Append the timestamps of ``Ynew`` (Tn x n, same storage class as the session's Y) to the resident problem;
        W is extended on the device by the AR recursion, H and the lag weights are kept.  ``self.model`` must be
        replaced by a model with ``rows()`` timestamps before the next ``download()``."""
        block = Ynew if isinstance(Ynew, PyMatrix) else PyMatrix(Ynew, dtype=self.model.W.dtype)
        self._check(self.lib.trmf_session_append_rows(self.handle, byref(block)), 'trmf_session_append_rows')
        return self

    def set_transform(self, transform):
        """Train on ``transform.preprocess`` of the raw dense Y held by the session (``None``: the raw values); the
        coefficients (``a``, ``b`` of a ``NormalizedTransform``, which have the dtype of the Y they were fitted on)
        are applied on the device in the session's element type, as NumPy evaluates ``Y * a + b``."""
        if transform is None:
            rc = self.lib.trmf_session_set_series_transform(self.handle, None, None)
        else:
            dt = self.model.W.dtype
            if np.asarray(transform.a).dtype != dt:
                raise TypeError('transform coefficients are {}, the session trains in {}'.format(np.asarray(transform.a).dtype, dt))
            a = np.ascontiguousarray(np.asarray(transform.a).ravel())
            b = np.ascontiguousarray(np.asarray(transform.b, dtype=dt).ravel())
            rc = self.lib.trmf_session_set_series_transform(self.handle, a.ctypes.data, b.ctypes.data)
        self._check(rc, 'trmf_session_set_series_transform')
        return self

    def rows(self):
        return self._check(self.lib.trmf_session_rows(self.handle), 'trmf_session_rows')

    def download(self):
        m = self.model
        self._check(self.lib.trmf_session_download(self.handle, byref(m.pyW), byref(m.pyH), byref(m.pylag_val)),
                    'trmf_session_download')
        return m

    def stats(self, last=64):
        buf = (TrmfIterStats * last)()
        cnt = self._check(self.lib.trmf_session_stats(self.handle, buf, last), 'trmf_session_stats')
        return [buf[i].as_dict() for i in range(cnt)]

    def objective(self):
        return self.lib.trmf_session_objective(self.handle)

    def fsolve_bytes(self):
        return self.lib.trmf_session_fsolve_bytes(self.handle)

    def describe(self):
        """One line: which phases are sharded, the form of the X-solve the measure-once rule chose (with the measured
        X-phase time of every candidate) and whether the peer-to-peer transport is available."""
        buf = ctypes.create_string_buffer(1024)
        self._check(self.lib.trmf_session_describe(self.handle, buf, len(buf)), 'trmf_session_describe')
        return buf.value.decode()

    def close(self):
        if self.handle:
            self.lib.trmf_session_destroy(self.handle)
            self.handle = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def partition_by_nnz(ptr, world, dtype=np.float32):
    """Row partition used by the sharded solver (host logic of the library, for tests/tools)."""
    ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
    bounds = np.zeros(world + 1, dtype=np.uint64)
    lib = lib_for(dtype)
    rc = lib.trmf_partition_by_nnz(len(ptr) - 1, ptr.ctypes.data_as(POINTER(c_uint64)), world,
                                   bounds.ctypes.data_as(POINTER(c_uint64)))
    if rc != 0:
        raise RuntimeError(lib.trmf_last_error().decode())
    return bounds
