"""ctypes boundary helpers for the MI355X TRMF solver.

Mirrors the names and semantics of the reference's ``python/trmf/rf_util.py`` so that code written
against ``trmf.rf_util.PyMatrix`` keeps working:

* ``PyMatrix``                -- reference ``rf_util.py:35-130`` / C struct ``rf_matrix.h:3398-3416``
* ``load_dynamic_library``    -- reference ``rf_util.py:19-32`` (here: builds with hipcc via the
  package Makefile instead of the reference's BLAS-probing make; raises if that fails -- there is
  NO CPU fallback: without the HIP library the product path does not run).
"""
import ctypes
import glob
import os
import subprocess
from ctypes import POINTER, c_int32, c_uint32, c_uint64, c_void_p

import numpy as np
import scipy.sparse as smat


def fillprototype(f, restype, argtypes):
    f.restype = restype
    f.argtypes = argtypes


def load_dynamic_library(dirname, soname, forced_rebuild=False):
    """Find ``<dirname>/<soname>*.so`` and CDLL it; try one rebuild if it is missing."""
    pattern = os.path.join(dirname, soname) + '*.so'

    def _find():
        hits = sorted(glob.glob(pattern))
        return hits[0] if hits else None

    path = None if forced_rebuild else _find()
    if path is None:
        pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(dirname.rstrip('/'))))
        try:
            subprocess.run(['make', '-C', pkg_root, 'lib'], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
        except Exception as exc:  # noqa: BLE001 - mirror the reference's "cannot be built" error
            raise Exception('{} library cannot be found and built.'.format(soname)) from exc
        path = _find()
    if path is None:
        raise Exception('{} library cannot be found and built.'.format(soname))
    return ctypes.CDLL(path)


class PyMatrix(ctypes.Structure):
    """NumPy / SciPy matrix exposed to C as non-owning views (sizeof == 80).

    Sparse input always carries BOTH orientations: CSR = (row_ptr, col_idx, val_t) and
    CSC = (col_ptr, row_idx, val); dense input carries only ``val``.  The arrays are pinned in
    ``self.py_buf`` for the lifetime of the object.
    """
    DENSE_ROWMAJOR = 1
    DENSE_COLMAJOR = 2
    SPARSE = 3
    EYE = 4

    _fields_ = [
        ('rows', c_uint64),
        ('cols', c_uint64),
        ('nnz', c_uint64),
        ('row_ptr', POINTER(c_uint64)),
        ('col_ptr', POINTER(c_uint64)),
        ('row_idx', POINTER(c_uint32)),
        ('col_idx', POINTER(c_uint32)),
        ('val', c_void_p),
        ('val_t', c_void_p),
        ('type', c_int32),
    ]

    def __init__(self, A, dtype=np.float32):
        super().__init__()
        if A is None:
            return
        self.rows, self.cols = int(A.shape[0]), int(A.shape[1])
        self.dtype = np.dtype(dtype).type
        self.py_buf = buf = {}

        if smat.issparse(A):
            if isinstance(A, smat.coo_matrix):
                # duplicates are kept as separate entries by the reference's coo path
                # (rf_util.py:98-118): stable sort by (row, col) without summing.
                csr = self._coo_to_compressed(A.row, A.col, A.data, A.shape[0], A.shape[1])
                csc = self._coo_to_compressed(A.col, A.row, A.data, A.shape[1], A.shape[0])
                buf['row_ptr'], buf['col_idx'], buf['val_t'] = csr
                buf['col_ptr'], buf['row_idx'], buf['val'] = csc
                self.nnz = int(A.data.shape[0])
            else:
                Acsr = smat.csr_matrix(A)
                Acsc = smat.csc_matrix(A)
                buf['row_ptr'] = Acsr.indptr.astype(np.uint64)
                buf['col_idx'] = Acsr.indices.astype(np.uint32)
                buf['val_t'] = Acsr.data.astype(dtype)
                buf['col_ptr'] = Acsc.indptr.astype(np.uint64)
                buf['row_idx'] = Acsc.indices.astype(np.uint32)
                buf['val'] = Acsc.data.astype(dtype)
                self.nnz = int(Acsr.indptr[-1])
            self.type = PyMatrix.SPARSE
        elif isinstance(A, np.ndarray):
            buf['val'] = A.astype(dtype)          # keeps the memory order of A ('K')
            # f_contiguous is tested first, exactly like the reference (quirk Q2: a (T,1) array
            # is tagged column-major).
            self.type = (PyMatrix.DENSE_COLMAJOR if buf['val'].flags.f_contiguous
                         else PyMatrix.DENSE_ROWMAJOR)
            self.nnz = int(A.shape[0] * A.shape[1])
        else:
            raise TypeError('PyMatrix: unsupported matrix type {}'.format(type(A)))

        ctype_of = dict(PyMatrix._fields_)
        for name, arr in buf.items():
            setattr(self, name, arr.ctypes.data_as(ctype_of[name]))

    def _coo_to_compressed(self, major, minor, data, n_major, n_minor):
        counts = np.bincount(np.asarray(major, dtype=np.int64), minlength=n_major)
        indptr = np.zeros(n_major + 1, dtype=np.uint64)
        np.cumsum(counts, out=indptr[1:])
        order = np.argsort(np.asarray(major, dtype=np.int64) * n_minor
                           + np.asarray(minor, dtype=np.int64), kind='stable')
        return (indptr, np.asarray(minor)[order].astype(np.uint32),
                np.asarray(data)[order].astype(self.dtype))

    @classmethod
    def identity(cls, size, dtype=np.float32):
        eye = cls(None)
        eye.rows = eye.cols = eye.nnz = int(size)
        eye.dtype = np.dtype(dtype).type
        eye.type = PyMatrix.EYE
        eye.py_buf = {}
        return eye
