"""CPU: the pure-Python front end mirrors the reference's harness semantics (trmf.py:82-346)."""
import numpy as np
import scipy.sparse as smat

import trmf
from trmf.model import NormalizedTransform


def test_initialize_semantics():
    Y = smat.random(50, 30, density=0.2, random_state=np.random.RandomState(0), format='csr', dtype=np.float32)
    m = trmf.Model.initialize(Y, [5, 1, 2], 6, seed=0)
    assert m.lag_set.dtype == np.uint32 and m.lag_set.tolist() == [1, 2, 5]        # sorted (trmf.py:229)
    assert m.W.shape == (50, 6) and m.W.flags.c_contiguous and m.W.dtype == np.float32
    assert m.H.shape == (30, 6) and m.lag_val.shape == (3, 6) and m.lag_val.flags.f_contiguous
    assert m.pyW.type == 1 and m.pyH.type == 1 and m.pylag_val.type == 2
    # same RNG stream as the reference: rand(W), rand(H), randn(Theta) after np.random.seed(seed)
    np.random.seed(0)
    W = np.random.rand(50, 6); H = np.random.rand(30, 6); Th = np.random.randn(3, 6)
    assert np.allclose(m.W, W.astype(np.float32)) and np.allclose(m.H, H.astype(np.float32))
    assert np.allclose(m.lag_val, Th.astype(np.float32))
    assert m.k == 6 and m.m == 50 and m.n == 30


def test_forecast_and_warm_start():
    d = trmf.Model.syn_gen(60, 12, 3, [1, 2, 4], seed=0, dtype=np.float64)
    m = trmf.Model.initialize(d['Y'], d['lag_set'], 3, seed=1)
    m.W[:] = d['W']; m.H[:] = d['H']; m.lag_val[:] = d['lag_val']
    Wn = m.latent_forecast(3)
    assert Wn.shape == (63, 3) and np.array_equal(Wn[:60], d['W'])
    for i in range(60, 63):
        assert np.allclose(Wn[i], (Wn[i - d['lag_set'].astype(int)] * d['lag_val']).sum(axis=0))
    Yn, Wt = m.forecast(3)
    assert np.allclose(Yn, Wn[60:] @ d['H'].T)
    assert np.all(m.forecast(3, threshold=0.0)[0] >= 0)
    Ybig = np.vstack([d['Y'], Yn])
    m2 = trmf.Model.initialize(Ybig, d['lag_set'], 3, warm_start_model=m)
    assert np.allclose(m2.W, Wn) and np.array_equal(m2.H, m.H) and np.array_equal(m2.lag_val, m.lag_val)


def test_save_load_roundtrip(tmp_path):
    d = trmf.Model.syn_gen(20, 8, 2, [1, 3], seed=0, dtype=np.float32)
    m = trmf.Model.initialize(d['Y'], d['lag_set'], 2, seed=0, transform=True)
    m.save(str(tmp_path / 'mdl'))
    assert (tmp_path / 'mdl' / 'arrays.npz').exists() and (tmp_path / 'mdl' / 'other.pkl').exists()
    m2 = trmf.Model.load(str(tmp_path / 'mdl'))
    assert np.array_equal(m2.W, m.W) and np.array_equal(m2.H, m.H) and np.array_equal(m2.lag_val, m.lag_val)
    assert m2.pylag_val.type == 2 and np.array_equal(m2.lag_set, m.lag_set)
    assert np.allclose(m2.transform.a, m.transform.a)


def test_normalized_transform_roundtrip():
    rng = np.random.RandomState(0)
    Y = rng.randn(40, 5) * [1, 2, 3, 0, 5] + 7
    t = NormalizedTransform(Y)
    Z = t.preprocess(Y)
    assert np.allclose(Z[:, [0, 1, 2, 4]].mean(axis=0), 0) and np.allclose(Z[:, [0, 1, 2, 4]].std(axis=0), 1)
    assert np.allclose(t.postprocess(Z), Y)


def test_metrics_known_values():
    true = np.array([[1.0, 2.0], [2.0, 0.0], [4.0, 2.0]])
    pred = true + np.array([[0.5, -0.5], [0.5, 0.5], [-1.0, 0.0]])
    m = trmf.Metrics.generate(true, pred)
    assert np.isclose(m.nd, 3.0 / 11.0)
    assert np.isclose(m.nrmse, np.sqrt((0.25 * 4 + 1.0) / 6) / (11.0 / 6))
    assert np.isclose(m.mase, (3.0 / 6) / ((1 + 2 + 2 + 2) / 4))
    assert np.isclose(m.mape, np.mean([0.5, 0.25, 0.25, 0.25, 0.0]))       # zero entry of trueY excluded
    assert str(trmf.Metrics.default()).startswith('nd=1e+10')


def test_public_names():
    for name in ('Model', 'Metrics', 'train', 'fit', 'rolling_validate', 'grid_search'):
        assert hasattr(trmf, name)
    assert trmf.fit is trmf.train


# ---- replay of tests/golden/py_harness.npz: outputs of the REFERENCE's Python harness (captured by
# ---- tests/golden/make_py_golden.py from the real package) through this repo's own front end --------------------
import os

import pytest

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'py_harness.npz'))


def test_metrics_match_reference_capture():
    m = trmf.Metrics.generate(GOLD['met_true'], GOLD['met_pred'])
    assert list(m._fields) == list(GOLD['met_fields'])
    # zero truths present: every field but MAPE is defined by the reference (its MAPE reads uninitialised memory at
    # those entries -- np.divide(where=) without out=, trmf.py:300; here they are excluded from the mean)
    assert np.allclose(np.array(m)[:6], GOLD['met_values'][:6], rtol=1e-12, atol=0)
    nz = GOLD['met_true'] != 0
    assert np.isclose(m.mape, np.mean(np.abs(GOLD['met_pred'] - GOLD['met_true'])[nz] / np.abs(GOLD['met_true'])[nz]))
    m2 = trmf.Metrics.generate(GOLD['met2_true'], GOLD['met_pred'])
    assert np.allclose(np.array(m2), GOLD['met2_values'], rtol=1e-12, atol=0)          # no zero truths: all seven agree


@pytest.mark.parametrize('tag,dt', [('f32', np.float32), ('f64', np.float64)])
def test_model_helpers_match_reference_capture(tag, dt):
    d = trmf.Model.syn_gen(50, 9, 4, [1, 3, 7], seed=11, dtype=dt)
    for key in ('W', 'H', 'lag_val', 'lag_set', 'Y'):
        assert d[key].dtype == GOLD['syn_%s_%s' % (tag, key)].dtype
        assert np.array_equal(d[key], GOLD['syn_%s_%s' % (tag, key)]), key          # same RNG stream, same arithmetic
    mod = trmf.Model.initialize(d['Y'], [7, 1, 3], 4, seed=2)
    for key in ('W', 'H', 'lag_val'):
        assert np.array_equal(getattr(mod, key), GOLD['init_%s_%s' % (tag, key)]), key
    mod.W[:] = d['W']; mod.H[:] = d['H']; mod.lag_val[:] = d['lag_val']
    assert np.array_equal(mod.latent_forecast(6), GOLD['lat_%s' % tag])
    Yn, Wn = mod.forecast(6, threshold=0.05)
    assert np.array_equal(Yn, GOLD['fc_%s_Y' % tag]) and np.array_equal(Wn, GOLD['fc_%s_W' % tag])
    warm = trmf.Model.initialize(np.vstack([d['Y'], Yn]).astype(dt), [1, 3, 7], 4, seed=3, warm_start_model=mod)
    assert np.array_equal(warm.W, GOLD['warm_%s_W' % tag])
    modt = trmf.Model.initialize(d['Y'], [1, 3, 7], 4, seed=2, transform=True)
    assert np.array_equal(modt.transform.a, GOLD['tr_%s_a' % tag]) and np.array_equal(modt.transform.b, GOLD['tr_%s_b' % tag])
    assert np.array_equal(modt.transform.preprocess(d['Y']), GOLD['tr_%s_pre' % tag])
    modt.W[:] = d['W']; modt.H[:] = d['H']; modt.lag_val[:] = d['lag_val']
    assert np.allclose(modt.forecast(3)[0], GOLD['tr_%s_fc' % tag], rtol=1e-6 if dt == np.float32 else 1e-13)


def _fields(metrics):
    return np.array([getattr(metrics, f) for f in metrics._fields])


@pytest.mark.gpu
@pytest.mark.parametrize('resident', [True, False])
def test_rolling_validate_on_gpu_matches_reference_harness(resident):
    """rolling_validate end to end (GPU solver behind it) against the metrics the REFERENCE harness produced with its
    own CPU solver on the same data (fp64): observed-entries and full-observation training, resident session and
    per-window uploads, and the per-window transform."""
    Y = GOLD['rv_Y']
    kw = dict(k=3, window_size=8, nr_windows=3, lambdaI=0.5, lambdaAR=50, lambdaLag=0.5, max_iter=4, threads=2, seed=0)
    for missing in (False, True):
        got = trmf.rolling_validate(Y, [1, 2, 5], missing=missing, threshold=0, resident=resident, **kw)
        assert np.allclose(_fields(got), GOLD['rv_missing%d' % int(missing)], rtol=1e-7), (missing, resident)
    got = trmf.rolling_validate(Y, [1, 2, 5], missing=False, threshold=None, transform=True, resident=resident, **kw)
    assert np.allclose(_fields(got), GOLD['rv_transform'], rtol=1e-7)


@pytest.mark.gpu
def test_grid_search_on_gpu_matches_reference_harness(capsys):
    results, best = trmf.grid_search(GOLD['rv_Y'], [1, 2, 5], {'lambdaI': [0.5, 5.0], 'lambdaAR': [5, 50]}, k=3, window_size=8,
                                     nr_windows=2, max_iter=3, missing=True, threshold=0, threads=2, seed=0)
    assert [r['kws']['lambdaI'] for r in results] == GOLD['gs_lambdaI'].tolist()
    assert [r['kws']['lambdaAR'] for r in results] == GOLD['gs_lambdaAR'].tolist()
    assert np.allclose([r['metrics'].m_nd for r in results], GOLD['gs_m_nd'], rtol=1e-7)
    assert np.allclose(_fields(best), GOLD['gs_best'], rtol=1e-7)
    assert 'm_nd=' in capsys.readouterr().out                       # improvements are printed


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,missing,unfused', [(np.float32, True, False), (np.float64, True, False), (np.float64, False, False),
                                                   (np.float64, True, True), (np.float32, True, True)])
def test_session_append_rows_equals_fresh_session(dtype, missing, unfused, monkeypatch):
    """trmf_session_append_rows: a session grown by a block of new timestamps (CSR appended, CSC rebuilt on the device,
    W rolled forward on the device) must continue exactly like a fresh session created on the whole matrix with the
    host-side warm start.  unfused: the two-kernel X-solve (TRMF_NO_HV_TILE), whose packed Gram cache is re-sized with T."""
    if unfused:
        monkeypatch.setenv('TRMF_NO_HV_TILE', '1')
    import scipy.sparse as smat
    from helpers import make_model
    from trmf import session, synth
    T0, Tn, n, k, lags = 300, 37, 80, 8, [1, 2, 6]
    d = trmf.Model.syn_gen(T0 + Tn, n, k, lags, seed=9, dtype=np.float64)
    Y = d['Y'] + 0.05 * np.random.RandomState(9).randn(T0 + Tn, n)
    if missing:
        Y = Y * (np.random.RandomState(10).rand(T0 + Tn, n) < 0.3)
    Y = np.ascontiguousarray(Y, dtype=dtype)
    wrap = (lambda a: smat.csr_matrix(a)) if missing else (lambda a: a)
    m0 = trmf.Model.initialize(Y[:T0], lags, k, seed=1, dtype=dtype)
    grown = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
    with session.Session(wrap(Y[:T0]), grown, missing=missing, **synth.HYPER) as s:
        s.run(3).download()
        first = make_model(grown.W, grown.H, grown.lag_val, grown.lag_set)       # state after the first window
        assert s.rows() == T0
        s.append_rows(wrap(Y[T0:]))
        assert s.rows() == T0 + Tn
        grown2 = trmf.Model.initialize(Y, lags, k, seed=1, dtype=dtype, warm_start_model=first)
        s.model = grown2
        s.download()
        warm = trmf.Model.initialize(Y, lags, k, seed=1, dtype=dtype, warm_start_model=first)
        assert np.array_equal(grown2.W, warm.W) and np.array_equal(grown2.H, warm.H)      # device roll-out == host roll-out
        s.run(3); st_g = s.stats(3); s.download()
    fresh = make_model(warm.W, warm.H, warm.lag_val, warm.lag_set)
    with session.Session(wrap(Y), fresh, missing=missing, **synth.HYPER) as s:
        s.run(3); st_f = s.stats(3); s.download()
    assert [x['cg_iter'] for x in st_g] == [x['cg_iter'] for x in st_f]
    assert np.array_equal(grown2.W, fresh.W) and np.array_equal(grown2.H, fresh.H) and np.array_equal(grown2.lag_val, fresh.lag_val)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_session_series_transform_equals_host_preprocess(dtype):
    """trmf_session_set_series_transform: a session holding the RAW dense matrix with the per-series affine map applied
    on the device trains exactly like a fresh session on NormalizedTransform.preprocess(Y) evaluated by NumPy -- also
    after the raw matrix has grown and the transform has been refitted (the rolling-window flow)."""
    from helpers import make_model
    from trmf import session, synth
    T0, Tn, n, k, lags = 260, 20, 31, 5, [1, 2, 7]
    d = trmf.Model.syn_gen(T0 + Tn, n, k, lags, seed=4, dtype=np.float64)
    Y = np.ascontiguousarray(3.0 * d['Y'] + 5.0 + 0.1 * np.random.RandomState(4).randn(T0 + Tn, n), dtype=dtype)
    hyper = dict(lambdaI=0.5, lambdaAR=125.0, lambdaLag=2.0)
    m0 = trmf.Model.initialize(Y[:T0], lags, k, seed=1, transform=True)
    dev = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
    with session.Session(Y[:T0], dev, missing=False, **hyper) as s:
        s.set_transform(m0.transform)
        s.run(3).download()
        host = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
        with session.Session(np.ascontiguousarray(m0.transform.preprocess(Y[:T0]).astype(dtype)), host, missing=False, **hyper) as s2:
            s2.run(3).download()
        assert np.array_equal(dev.W, host.W) and np.array_equal(dev.H, host.H) and np.array_equal(dev.lag_val, host.lag_val)
        # grow by a block of RAW rows, refit the transform on the grown prefix
        first = make_model(dev.W, dev.H, dev.lag_val, dev.lag_set)
        first.transform = m0.transform                       # a warm start inherits (and refits) the previous model's transform
        m1 = trmf.Model.initialize(Y, lags, k, seed=1, warm_start_model=first)
        s.append_rows(Y[T0:])
        s.set_transform(m1.transform)
        dev1 = make_model(m1.W, m1.H, m1.lag_val, m1.lag_set)
        s.model = dev1
        s.run(3).download()
    host1 = make_model(m1.W, m1.H, m1.lag_val, m1.lag_set)
    with session.Session(np.ascontiguousarray(m1.transform.preprocess(Y).astype(dtype)), host1, missing=False, **hyper) as s3:
        s3.run(3).download()
    assert np.array_equal(dev1.W, host1.W) and np.array_equal(dev1.H, host1.H) and np.array_equal(dev1.lag_val, host1.lag_val)
