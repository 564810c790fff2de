"""ctypes access to the oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (exp-trmf-nips16_amd/trmf) never does.

  port(dtype)  -> oracle/libtrmf_oracle_f{32,64}.so   our C restatement (travels to the GPU box)
  ref(dtype)   -> oracle/_ref/trmf_float{32,64}.so    the real reference, built by `make -C oracle ref`
                  (None when it has not been built)
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, byref, c_double, c_int32, c_uint32, c_uint64, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class OracleMatrix(Structure):      # same layout as the reference's PyMatrix (80 bytes)
    _fields_ = [('rows', c_uint64), ('cols', c_uint64), ('nnz', c_uint64),
                ('row_ptr', c_void_p), ('col_ptr', c_void_p), ('row_idx', c_void_p), ('col_idx', c_void_p),
                ('val', c_void_p), ('val_t', c_void_p), ('type', c_int32)]


class OracleXStats(Structure):
    _fields_ = [('cg_iter', c_int32), ('accepted', c_int32), ('f', c_double), ('fnew', c_double),
                ('actred', c_double), ('prered', c_double), ('gnorm', c_double), ('cg_rnorm', c_double)]


class OracleIterLog(Structure):
    _fields_ = [('normF', c_double), ('normX', c_double), ('normLV', c_double), ('x', OracleXStats)]


def _suffix(dtype):
    return 'f64' if np.dtype(dtype) == np.float64 else 'f32'


_cache = {}


def build_port():
    subprocess.run(['make', '-C', HERE, 'port'], check=True, stdout=subprocess.DEVNULL)


def port(dtype=np.float32):
    key = ('port', _suffix(dtype))
    if key not in _cache:
        path = os.path.join(HERE, 'libtrmf_oracle_{}.so'.format(_suffix(dtype)))
        if not os.path.exists(path):
            build_port()
        lib = ctypes.CDLL(path)
        lib.oracle_xfun_sparse.restype = c_double
        lib.oracle_objective_sparse.restype = c_double
        _cache[key] = lib
    return _cache[key]


def ref(dtype=np.float32):
    key = ('ref', _suffix(dtype))
    if key not in _cache:
        name = 'trmf_float64.so' if np.dtype(dtype) == np.float64 else 'trmf_float32.so'
        path = os.path.join(HERE, '_ref', name)
        os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')   # OpenBLAS' own pool fights OpenMP
        _cache[key] = ctypes.CDLL(path) if os.path.exists(path) else None
    return _cache[key]


class Mat(object):
    """Keeps NumPy buffers alive and exposes them as an OracleMatrix."""

    def __init__(self, A, dtype):
        import scipy.sparse as smat
        self.m = OracleMatrix()
        self.buf = {}
        self.m.rows, self.m.cols = A.shape
        if smat.issparse(A):
            csr, csc = smat.csr_matrix(A), smat.csc_matrix(A)
            self.buf = dict(row_ptr=csr.indptr.astype(np.uint64), col_idx=csr.indices.astype(np.uint32),
                            val_t=csr.data.astype(dtype), col_ptr=csc.indptr.astype(np.uint64),
                            row_idx=csc.indices.astype(np.uint32), val=csc.data.astype(dtype))
            self.m.type, self.m.nnz = 3, int(csr.indptr[-1])
        else:
            assert A.dtype == np.dtype(dtype)
            self.buf = dict(val=A)      # in place: outputs are written into A
            self.m.type = 2 if (A.flags.f_contiguous and not A.flags.c_contiguous) or \
                (A.flags.f_contiguous and A.shape[0] != 1 and A.shape[1] == 1) else 1
            if A.flags.f_contiguous and A.flags.c_contiguous:
                self.m.type = 2          # mirrors rf_util.py:122-126 (f_contiguous tested first)
            self.m.nnz = A.shape[0] * A.shape[1]
        for name, arr in self.buf.items():
            setattr(self.m, name, arr.ctypes.data)

    def ref(self):
        return byref(self.m)


def _train_args(Y, lag_set, W, H, theta, hyper, max_iter, periods, threads, missing, verbose, warm_start=1):
    dtype = W.dtype
    mats = (Mat(Y, dtype), Mat(W, dtype), Mat(H, dtype), Mat(theta, dtype))
    lag_set = np.ascontiguousarray(lag_set, dtype=np.uint32)
    args = [mats[0].ref(), lag_set.ctypes.data_as(POINTER(c_uint32)), c_uint32(len(lag_set)),
            mats[1].ref(), mats[2].ref(), mats[3].ref(), c_int32(int(warm_start)),
            c_double(hyper['lambdaI']), c_double(hyper['lambdaAR']), c_double(hyper['lambdaLag']),
            c_int32(max_iter), c_int32(periods[0]), c_int32(periods[1]), c_int32(periods[2]),
            c_int32(threads), c_int32(int(missing)), c_int32(verbose)]
    return args, (mats, lag_set)


def train_ref(Y, lag_set, W, H, theta, hyper, max_iter=10, periods=(1, 1, 2), threads=4, missing=True, verbose=0, warm_start=1):
    """Run the REAL reference c_trmf_train in place on (W, H, theta)."""
    lib = ref(W.dtype)
    if lib is None:
        raise RuntimeError('oracle/_ref not built (run `make -C oracle ref` in the build container)')
    args, keep = _train_args(Y, lag_set, W, H, theta, hyper, max_iter, periods, threads, missing, verbose, warm_start)
    lib.c_trmf_train.restype = None
    lib.c_trmf_train(*args)
    return W, H, theta


def train_port(Y, lag_set, W, H, theta, hyper, max_iter=10, periods=(1, 1, 2), threads=4, missing=True, verbose=0):
    """Run the C restatement in place on (W, H, theta); returns the per-iteration log."""
    lib = port(W.dtype)
    args, keep = _train_args(Y, lag_set, W, H, theta, hyper, max_iter, periods, threads, missing, verbose)
    log = (OracleIterLog * max(max_iter, 1))()
    lib.oracle_trmf_train_log.restype = None
    lib.oracle_trmf_train_log(*args, log)
    return [dict(normF=l.normF, normX=l.normX, normLV=l.normLV, cg_iter=l.x.cg_iter, accepted=l.x.accepted,
                 f=l.x.f, fnew=l.x.fnew, gnorm=l.x.gnorm, cg_rnorm=l.x.cg_rnorm) for l in log[:max_iter]]


def objective(Y, lag_set, W, H, theta, hyper):
    """Global objective J in fp64 on factors of any dtype (parity gate of SURVEY.md 8(d))."""
    import scipy.sparse as smat
    lib = port(np.float64)
    csr = smat.csr_matrix(Y)
    ptr = csr.indptr.astype(np.uint64); idx = csr.indices.astype(np.uint32); val = csr.data.astype(np.float64)
    W64 = np.ascontiguousarray(W, dtype=np.float64); H64 = np.ascontiguousarray(H, dtype=np.float64)
    th = np.asfortranarray(theta, dtype=np.float64)
    lag_set = np.ascontiguousarray(lag_set, dtype=np.uint32)
    return lib.oracle_objective_sparse(
        c_uint64(W.shape[0]), c_uint64(H.shape[0]), ptr.ctypes.data_as(c_void_p), idx.ctypes.data_as(c_void_p),
        val.ctypes.data_as(c_void_p), W64.ctypes.data_as(c_void_p), H64.ctypes.data_as(c_void_p),
        c_uint64(W.shape[1]), lag_set.ctypes.data_as(c_void_p), c_uint32(len(lag_set)),
        th.ctypes.data_as(c_void_p), c_double(hyper['lambdaI']), c_double(hyper['lambdaAR']))


def fsolve_port(Yt_csr, X, F, lam, threads=4):
    """One F-solve over the rows of the CSR matrix Yt (items x timestamps) in place on F."""
    lib = port(F.dtype)
    ptr = Yt_csr.indptr.astype(np.uint64); idx = Yt_csr.indices.astype(np.uint32); val = Yt_csr.data.astype(F.dtype)
    lib.oracle_fsolve_sparse(c_uint64(Yt_csr.shape[0]), ptr.ctypes.data_as(c_void_p), idx.ctypes.data_as(c_void_p),
                             val.ctypes.data_as(c_void_p), X.ctypes.data_as(c_void_p), c_uint64(X.shape[1]),
                             c_double(lam), F.ctypes.data_as(c_void_p), c_int32(threads))
    return F
