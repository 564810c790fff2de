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
