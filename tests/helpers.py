"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import os

import numpy as np
import scipy.sparse as smat

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_names():
    # solver cases only; py_*.npz hold the Python-harness captures (tests/golden/make_py_golden.py)
    return sorted(name for name in (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))
                  if not name.startswith('py_'))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    T, n = [int(v) for v in z['shape']]
    if 'Y_dense' in z.files:
        Y = np.asarray(z['Y_dense'], order=str(z['Y_order']))
    else:
        Y = smat.csr_matrix((z['Y_data'], z['Y_indices'], z['Y_indptr']), shape=(T, n))
    g = {k: z[k] for k in z.files}
    g['missing'] = bool(int(z['missing'])) if 'missing' in z.files else True
    g['Y'] = Y
    g['hyper'] = dict(lambdaI=float(z['lambdaI']), lambdaAR=float(z['lambdaAR']), lambdaLag=float(z['lambdaLag']))
    g['max_iter'] = int(z['max_iter'])
    g['dtype'] = z['W0'].dtype
    return g


def relmax(a, ref):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(ref, dtype=np.float64)).max()
                 / max(np.abs(ref).max(), 1e-300))


def relfro(a, ref):
    a = np.asarray(a, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-300))


def make_model(W0, H0, Th0, lag_set):
    """A trmf.Model over copies of the given initial factors."""
    from trmf import Model
    from trmf.rf_util import PyMatrix
    dt = W0.dtype
    return Model(pyW=PyMatrix(np.ascontiguousarray(W0.copy()), dt), pyH=PyMatrix(np.ascontiguousarray(H0.copy()), dt),
                 pylag_val=PyMatrix(np.asfortranarray(Th0.copy()), dt), lag_set=np.array(lag_set, dtype=np.uint32))


# parity tolerances (SURVEY.md 8(d) "Parity gates"), by element type
TOL = {
    'float64': dict(factor=1e-6, objective=1e-8),
    'float32': dict(factor=1e-3, objective=1e-5),
}


class capture_fds(object):
    """Capture what C code writes to fd 1 and fd 2 (the libraries print their verbose lines with fprintf)."""

    def __enter__(self):
        import os
        import sys
        import tempfile
        sys.stdout.flush(); sys.stderr.flush()
        self.saved = [os.dup(1), os.dup(2)]
        self.tmp = [tempfile.TemporaryFile(), tempfile.TemporaryFile()]
        os.dup2(self.tmp[0].fileno(), 1); os.dup2(self.tmp[1].fileno(), 2)
        return self

    def __exit__(self, *exc):
        import ctypes
        import os
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved[0], 1); os.dup2(self.saved[1], 2)
        self.out, self.err = [], []
        for t, dst in zip(self.tmp, (self.out, self.err)):
            t.seek(0); dst.extend(t.read().decode().splitlines()); t.close()
        for fd in self.saved:
            os.close(fd)


def evidence(line):
    """Print a measured line of a test and, on the GPU box, append it to gpurun_out/test_evidence.txt -- merged back by gpurun
    and copied to profiles/ at round end (the numbers the tests print must be on record, not only in a scratch log)."""
    print(line)
    root = os.environ.get('GRAFT_REPO_ROOT')
    if root and os.path.isdir(os.path.join(root, 'gpurun_out')):
        with open(os.path.join(root, 'gpurun_out', 'test_evidence.txt'), 'a') as fh:
            fh.write(line + '\n')
