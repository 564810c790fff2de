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


# ---- the fp32 noise floor as a MEASURED yardstick ------------------------------------------------------------
# A CG truncated at ||r|| <= 0.1 ||g|| amplifies last-bit differences from iteration to iteration; in fp32 the reference's own
# build is itself that far from its line-by-line restatement (its BLAS dot products are order dependent).  Where an fp32 run misses
# the direct gates it may instead be no farther from the fp64 trajectory than the reference-side fp32 runs are -- measured in the
# test, on the same inputs: the fp32 restatement and, when oracle/_ref travelled with the tree, the reference's fp32 build on two
# OpenMP thread counts.  No whitelist of cases.
def fp32_noise_yardstick(Y, lag_set, W0, H0, Th0, hyper, iters, periods=(1, 1, 2), threads=None, with_ref=True):
    import oracle_py as O
    ncpu = os.cpu_count() or 8
    threads = threads or (ncpu if Y.nnz > 2000000 else min(8, ncpu))      # small problems: OpenMP region entry on 256 threads dominates
    Y64 = Y.astype(np.float64)
    W64, H64 = W0.astype(np.float64), H0.astype(np.float64)
    T64 = np.asfortranarray(Th0.astype(np.float64))
    O.train_port(Y64, lag_set, W64, H64, T64, hyper, max_iter=iters, periods=periods, threads=threads)
    J64 = O.objective(Y64, lag_set, W64, H64, T64, hyper)
    runs = []

    def dist(name, W, H, Th):
        J = O.objective(Y64, lag_set, W, H, Th, hyper)
        runs.append(dict(name=name, W=relfro(W, W64), H=relfro(H, H64), Th=relfro(Th, T64), J=abs(J - J64) / abs(J64)))

    Y32 = Y.astype(np.float32)
    W, H, Th = W0.astype(np.float32), H0.astype(np.float32), np.asfortranarray(Th0.astype(np.float32))
    O.train_port(Y32, lag_set, W, H, Th, hyper, max_iter=iters, periods=periods, threads=threads)
    dist('restatement fp32', W, H, Th)
    if with_ref and O.ref(np.float32) is not None:
        for t in sorted({min(64, ncpu) if Y.nnz > 2000000 else min(16, ncpu), min(8, ncpu)}):
            W, H, Th = W0.astype(np.float32), H0.astype(np.float32), np.asfortranarray(Th0.astype(np.float32))
            O.train_ref(Y32, lag_set, W, H, Th, hyper, max_iter=iters, periods=periods, threads=t)
            dist('reference fp32 build, %d threads' % t, W, H, Th)
    yard = {key: max(r[key] for r in runs) for key in ('W', 'H', 'Th', 'J')}
    return dict(W64=W64, H64=H64, T64=T64, J64=J64, Y64=Y64, runs=runs, yard=yard)


def assert_within_fp32_noise(model, ys, lag_set, hyper, c=2.0, what=''):
    """The GPU's fp32 factors / objective are no farther from the fp64 trajectory than c x the reference side's own fp32 runs."""
    import oracle_py as O
    Jg = O.objective(ys['Y64'], lag_set, model.W, model.H, model.lag_val, hyper)
    got = dict(W=relfro(model.W, ys['W64']), H=relfro(model.H, ys['H64']), Th=relfro(model.lag_val, ys['T64']), J=abs(Jg - ys['J64']) / abs(ys['J64']))
    evidence('%sfp32 noise floor, distance to the fp64 trajectory: GPU W %.2e H %.2e Theta %.2e J %.2e; reference side (max of %s): W %.2e H %.2e Theta %.2e J %.2e' % (
        what + ': ' if what else '', got['W'], got['H'], got['Th'], got['J'], ', '.join(r['name'] for r in ys['runs']),
        ys['yard']['W'], ys['yard']['H'], ys['yard']['Th'], ys['yard']['J']))
    for key in ('W', 'H', 'Th', 'J'):
        assert got[key] <= c * ys['yard'][key] + 1e-7, (key, got[key], ys['yard'][key])
    return got
