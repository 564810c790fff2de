"""CPU: the C-ABI libraries load, export every symbol include/trmf_abi.h declares, agree with the
reference's PyMatrix layout, validate arguments like the reference, and fail LOUDLY (no silent CPU
fallback) when no GPU is present.  No compute is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as smat

from helpers import make_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'trmf_abi.h')


def declared_symbols():
    text = open(HEADER).read()
    # declarations look like:  TRMF_API <return type> name(args);   (the #define line is skipped)
    return sorted(set(re.findall(r'^TRMF_API\s[^;(]*?\b(\w+)\s*\(', text, flags=re.M)))


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_library_exports_every_declared_symbol(dtype):
    from trmf import session
    lib = session.lib_for(dtype)
    names = declared_symbols()
    assert 'c_trmf_train' in names and len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert lib.trmf_sizeof_real() == np.dtype(dtype).itemsize


def test_library_names_follow_reference_glob():
    # reference loader: glob(<pkg>/corelib/trmf_float32*.so) (trmf.py:21-22, rf_util.py:23)
    import glob
    from trmf._corelib import corelib_path
    assert glob.glob(os.path.join(corelib_path, 'trmf_float32*.so'))
    assert glob.glob(os.path.join(corelib_path, 'trmf_float64*.so'))


def test_pymatrix_layout_matches_reference_struct():
    from trmf.rf_util import PyMatrix
    # rf_matrix.h:3407-3415 with natural alignment
    assert ctypes.sizeof(PyMatrix) == 80
    expect = dict(rows=0, cols=8, nnz=16, row_ptr=24, col_ptr=32, row_idx=40, col_idx=48, val=56, val_t=64, type=72)
    for name, off in expect.items():
        assert getattr(PyMatrix, name).offset == off, name


def test_pymatrix_conversions():
    from trmf.rf_util import PyMatrix
    rng = np.random.RandomState(0)
    A = smat.random(7, 5, density=0.4, random_state=rng, format='csr', dtype=np.float64)
    for fmt in (A, A.tocsc(), A.tocoo()):
        m = PyMatrix(fmt, dtype=np.float32)
        assert m.type == PyMatrix.SPARSE and m.nnz == A.nnz and (m.rows, m.cols) == (7, 5)
        assert m.py_buf['row_ptr'].dtype == np.uint64 and m.py_buf['col_idx'].dtype == np.uint32
        assert m.py_buf['val'].dtype == np.float32 and m.py_buf['val_t'].dtype == np.float32
        csr = smat.csr_matrix((m.py_buf['val_t'], m.py_buf['col_idx'], m.py_buf['row_ptr'].astype(np.int64)), shape=(7, 5))
        csc = smat.csc_matrix((m.py_buf['val'], m.py_buf['row_idx'], m.py_buf['col_ptr'].astype(np.int64)), shape=(7, 5))
        assert np.allclose(csr.toarray(), A.toarray()) and np.allclose(csc.toarray(), A.toarray())
    C = np.zeros((4, 3), order='C'); F = np.zeros((4, 3), order='F')
    assert PyMatrix(C, np.float64).type == PyMatrix.DENSE_ROWMAJOR
    assert PyMatrix(F, np.float64).type == PyMatrix.DENSE_COLMAJOR
    assert PyMatrix(np.zeros((4, 1)), np.float64).type == PyMatrix.DENSE_COLMAJOR     # quirk Q2


def test_partition_by_nnz():
    from trmf.session import partition_by_nnz
    rng = np.random.RandomState(1)
    counts = rng.poisson(20, size=1000)
    counts[100:140] = 0
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    for world in (1, 2, 3, 8):
        b = partition_by_nnz(ptr, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b.astype(np.int64)) >= 0)
        loads = [int(ptr[b[r + 1]] - ptr[b[r]]) for r in range(world)]
        assert max(loads) - min(loads) <= 2 * counts.max()
    # all-empty matrix: even row split
    b = partition_by_nnz(np.zeros(11, dtype=np.uint64), 4)
    assert b.tolist() == [0, 2, 5, 7, 10]


def _tiny():
    rng = np.random.RandomState(0)
    Y = smat.random(30, 20, density=0.3, random_state=rng, format='csr', dtype=np.float32)
    W0 = rng.rand(30, 4).astype(np.float32); H0 = rng.rand(20, 4).astype(np.float32)
    Th0 = np.asfortranarray(rng.randn(2, 4).astype(np.float32))
    return Y, W0, H0, Th0


def test_dimension_errors_leave_outputs_untouched(capfd):
    import trmf
    Y, W0, H0, Th0 = _tiny()
    model = make_model(W0[:-1], H0, Th0, [1, 2])                 # W has the wrong row count
    trmf.train(Y, model, missing=True, max_iter=2)
    err = capfd.readouterr().err
    assert '[ERR MSG]: Y.rows (30) != W.rows (29)' in err         # trmf.cpp:563-566
    assert np.array_equal(model.W, W0[:-1]) and np.array_equal(model.H, H0)


def test_cold_start_is_a_noop_like_the_reference():
    # warm_start=0 never updates the caller's arrays
    from trmf import session
    from trmf.rf_util import PyMatrix
    from ctypes import POINTER, byref, c_uint32
    Y, W0, H0, Th0 = _tiny()
    model = make_model(W0, H0, Th0, [1, 2])
    lib = session.lib_for(np.float32)
    pyY = PyMatrix(Y, np.float32)
    lib.c_trmf_train(byref(pyY), model.lag_set.ctypes.data_as(POINTER(c_uint32)), 2, byref(model.pyW),
                     byref(model.pyH), byref(model.pylag_val), 0, 0.5, 50.0, 0.5, 2, 1, 1, 2, 1, 1, 0)
    assert np.array_equal(model.W, W0) and np.array_equal(model.H, H0) and np.array_equal(model.lag_val, Th0)


def test_no_gpu_means_loud_failure_not_cpu_fallback(capfd, have_gpu):
    if have_gpu:
        pytest.skip('a GPU is present')
    import trmf
    from trmf import session
    Y, W0, H0, Th0 = _tiny()
    model = make_model(W0, H0, Th0, [1, 2])
    trmf.train(Y, model, missing=True, max_iter=2, lambdaI=0.5, lambdaAR=50, lambdaLag=0.5)
    err = capfd.readouterr().err
    assert 'no HIP device' in err
    assert np.array_equal(model.W, W0) and np.array_equal(model.H, H0)      # nothing computed on the host
    with pytest.raises(RuntimeError, match='no HIP device'):
        session.Session(Y, model, missing=True)


def test_cold_start_is_silent_and_never_writes(capfd):
    """warm_start == 0 (quirk Q1 of the reference): trmf_initialization rebuilds W, H and lag_val with matching shapes
    before check_dimension runs (trmf.cpp:547-558, 719-722), so the reference can print no "[ERR MSG]" line for the
    caller's shapes and never updates the caller's arrays.  Same here: silent on stderr, nothing written."""
    from ctypes import POINTER, byref, c_uint32
    from trmf import session
    from trmf.rf_util import PyMatrix
    lib = session.lib_for(np.float32)
    Y = smat.random(30, 20, density=0.3, random_state=np.random.RandomState(0), format='csr', dtype=np.float32)
    W = np.ones((29, 4), np.float32); H = np.ones((20, 4), np.float32); Th = np.ones((2, 4), np.float32, order='F')   # W has a row too few
    lags = np.array([1, 2], dtype=np.uint32)
    pyY, pyW, pyH, pyT = PyMatrix(Y), PyMatrix(W), PyMatrix(H), PyMatrix(Th)
    get = lambda m: m.py_buf['val'].copy()
    before = [get(pyW), get(pyH), get(pyT)]
    lib.c_trmf_train(byref(pyY), lags.ctypes.data_as(POINTER(c_uint32)), 2, byref(pyW), byref(pyH), byref(pyT), 0,
                     0.5, 50.0, 0.5, 2, 1, 1, 2, 1, 1, 0)
    err = capfd.readouterr().err
    assert 'Y.rows' not in err and 'W.rows' not in err           # nothing about the CALLER's shapes (a box without a GPU says so)
    assert all(np.array_equal(a, get(m)) for a, m in zip(before, (pyW, pyH, pyT)))


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['py_cold_f64', 'py_cold_f32'])
def test_cold_start_trains_the_references_private_model(name):
    """warm_start == 0 against a capture of the reference (tests/golden/make_golden.py: make_cold_case): the reference draws a
    private model from std::mt19937(0) (trmf.cpp:547-558), trains it, prints the ">> iter i F|X|LV" norms of THAT run under
    verbose and leaves the caller's arrays untouched.  Same generator calls here => the same starting point => the same lines
    (printed with %g: 6 digits), and the caller's arrays bit-for-bit unchanged."""
    import os
    import re
    from ctypes import POINTER, byref, c_uint32
    from helpers import GOLDEN_DIR, capture_fds
    from trmf import session
    from trmf.rf_util import PyMatrix
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    T, n = [int(v) for v in z['shape']]
    Y = smat.csr_matrix((z['Y_data'], z['Y_indices'], z['Y_indptr']), shape=(T, n))
    dtype = z['W0'].dtype
    model = make_model(z['W0'], z['H0'], z['Th0'], z['lag_set'])
    lib = session.lib_for(dtype)
    pyY = PyMatrix(Y, dtype)
    iters = int(z['max_iter'])
    with capture_fds() as cap:
        lib.c_trmf_train(byref(pyY), model.lag_set.ctypes.data_as(POINTER(c_uint32)), len(model.lag_set), byref(model.pyW),
                         byref(model.pyH), byref(model.pylag_val), 0, float(z['lambdaI']), float(z['lambdaAR']), float(z['lambdaLag']),
                         iters, 1, 1, 2, 1, 1, 1)
    assert np.array_equal(model.W, z['W0']) and np.array_equal(model.H, z['H0']) and np.array_equal(model.lag_val, z['Th0'])
    got = {'F': np.full(iters, -1.0), 'X': np.full(iters, -1.0), 'LV': np.full(iters, -1.0)}
    for line in cap.err:
        m = re.match(r'>> iter (\d+) (F|X|LV) (\S+)$', line.strip())
        if m:
            got[m.group(2)][int(m.group(1)) - 1] = float(m.group(3))
    tol = 2e-5 if dtype == np.float64 else 2e-4
    for key, ref in (('F', z['normF']), ('X', z['normX']), ('LV', z['normLV'])):
        assert np.all((ref < 0) == (got[key] < 0)), (key, got[key], ref)
        assert np.allclose(got[key], ref, rtol=tol), (key, got[key], ref)


def test_rank_one_quirk_q2_is_reproduced(capfd):
    """a (T, 1) NumPy array is both C- and F-contiguous and PyMatrix tests f_contiguous first, so
    W and H of a rank-1 model arrive tagged column-major; the reference's dimension check rejects them and returns
    without touching anything (trmf.cpp:583-586).  Same here, before any device is needed."""
    import trmf
    Y = smat.random(25, 12, density=0.4, random_state=np.random.RandomState(1), format='csr', dtype=np.float32)
    m = trmf.Model.initialize(Y, [1, 2], 1, seed=0)
    assert m.pyW.type == 2 and m.pyH.type == 2
    W0, H0 = m.W.copy(), m.H.copy()
    trmf.train(Y, m, max_iter=2, missing=True)
    err = capfd.readouterr().err
    assert 'W should be rowmajored' in err and 'H should be rowmajored' in err
    assert np.array_equal(m.W, W0) and np.array_equal(m.H, H0)
