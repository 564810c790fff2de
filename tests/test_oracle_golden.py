"""CPU: the C restatement (oracle/trmf_oracle.c) against the golden vectors captured from the real
reference (tests/golden/make_golden.py), and -- when oracle/_ref exists -- against the reference
itself on fresh seeds.  This is what pins the oracle."""
import numpy as np
import pytest

import oracle_py as O
from helpers import golden_names, load_golden, relfro, relmax


@pytest.mark.parametrize('name', golden_names())
def test_port_matches_golden(name):
    g = load_golden(name)
    W, H, Th = g['W0'].copy(), g['H0'].copy(), np.asfortranarray(g['Th0'].copy())
    log = O.train_port(g['Y'], g['lag_set'], W, H, Th, g['hyper'], max_iter=g['max_iter'], missing=g['missing'])
    f64 = g['dtype'] == np.float64
    tol = 1e-9 if f64 else 2e-4
    assert relmax(W, g['W']) < tol and relmax(H, g['H']) < tol and relmax(Th, g['Th']) < tol * 10
    # same truncated-CG trajectory as the reference
    assert [l['cg_iter'] for l in log] == g['cg_iter'].tolist()
    # the reference's own printed observables (%g = 6 significant digits)
    for key, field in (('normF', 'normF'), ('normX', 'normX'), ('normLV', 'normLV')):
        ref = g[key]; got = np.array([l[field] for l in log])
        mask = ref >= 0
        assert np.allclose(got[mask], ref[mask], rtol=2e-5)
        assert np.all(got[~mask] == -1)
    if g['missing']:
        J = O.objective(g['Y'], g['lag_set'], W, H, Th, g['hyper'])
        assert abs(J - float(g['objective'])) / float(g['objective']) < (1e-10 if f64 else 1e-5)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_port_matches_reference_build(dtype):
    if O.ref(dtype) is None:
        pytest.skip('oracle/_ref not built here (reference sources only exist in the build container)')
    from trmf import synth
    p = synth.sparse_problem(n=350, T=260, k=12, nlag=5, density=0.07, dtype=dtype, seed=11)
    m = synth.initial_model(p['Y'], p['lag_set'], 12, seed=11)
    W1, H1, T1 = m.W.copy(), m.H.copy(), np.asfortranarray(m.lag_val.copy())
    W2, H2, T2 = m.W.copy(), m.H.copy(), np.asfortranarray(m.lag_val.copy())
    O.train_ref(p['Y'], p['lag_set'], W1, H1, T1, synth.HYPER, max_iter=5)
    O.train_port(p['Y'], p['lag_set'], W2, H2, T2, synth.HYPER, max_iter=5)
    tol = 1e-9 if dtype == np.float64 else 2e-4
    assert relfro(W2, W1) < tol and relfro(H2, H1) < tol and relfro(T2, T1) < tol * 10


def test_fsolve_is_ridge_regression():
    """Known answer: the F-solve equals the closed-form ridge solution per item (fp64)."""
    import scipy.sparse as smat
    rng = np.random.RandomState(0)
    T, n, k, lam = 40, 25, 6, 0.3
    X = rng.randn(T, k)
    Yt = smat.random(n, T, density=0.4, random_state=rng, format='csr', dtype=np.float64)
    F = rng.randn(n, k)
    F0 = F.copy()
    empty = 7
    Yt = Yt.tolil(); Yt[empty, :] = 0; Yt = smat.csr_matrix(Yt); Yt.eliminate_zeros()
    O.fsolve_port(Yt, X, F, lam)
    for i in range(n):
        cols = Yt.indices[Yt.indptr[i]:Yt.indptr[i + 1]]
        if len(cols) == 0:
            assert np.array_equal(F[i], F0[i])          # trmf.cpp:374
            continue
        P = X[cols]; y = Yt.data[Yt.indptr[i]:Yt.indptr[i + 1]]
        ref = np.linalg.solve(P.T @ P + lam * np.eye(k), P.T @ y)
        assert np.allclose(F[i], ref, rtol=1e-9, atol=1e-12)
