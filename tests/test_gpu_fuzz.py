"""GPU (-m gpu): seeded random small problems across the shapes the kernels specialise on -- every rank class
(k = 2..64 covers all NT / KMAX instantiations), short and long lag sets (lag 0 included, reach up to T/2), very few items
or timestamps, empty rows and columns, both precisions, observed-entries and full-observation training (dense C / F
order and sparse) -- two ALS iterations against the C restatement at the SURVEY.md 8(d) gates."""
import numpy as np
import pytest
import scipy.sparse as smat

import oracle_py as O
import trmf
from helpers import TOL, make_model, relfro

pytestmark = pytest.mark.gpu

# Round 2 relaxed the fp32 factor gate 5x (5e-3) for every case.  Measured in round 3 (profiles/r03_fuzz_margins.txt: the
# margin of every case to the DIRECT gate of SURVEY.md 8(d), printed with -s): the worst fp32 case sits at 0.165 of the
# direct gate, all others below 0.02 -- no case needs a relaxation, so none is granted.
RELAXED_F32_SEEDS = set()
RELAXED_F32_PERIOD_SEEDS = set()


def _case(seed):
    rng = np.random.RandomState(1000 + seed)
    dtype = [np.float32, np.float64][seed % 2]
    k = int(rng.choice([2, 3, 5, 8, 9, 16, 17, 24, 31, 32, 33, 40, 41, 47, 48, 49, 56, 57, 63, 64]))
    nlag = int(rng.choice([1, 2, 3, 5, 8, 13, 24]))
    T = int(rng.randint(40, 500))
    reach = int(rng.randint(nlag, max(nlag + 1, T // 2)))
    lags = np.sort(rng.choice(np.arange(0 if seed % 5 == 0 else 1, reach + 1), size=min(nlag, reach), replace=False))
    n = int(rng.choice([2, 3, 4, 5, 17, 64, 150, 333]))       # n = 1 is the reference's quirk Q2 (row vector tagged column-major): a no-op
    missing = seed % 3 != 0
    hyper = dict(lambdaI=float(rng.choice([0.1, 0.5, 1.0, 2.0])), lambdaAR=float(rng.choice([0.5, 50.0, 625.0])),
                 lambdaLag=float(rng.choice([0.5, 2.0])))
    d = trmf.Model.syn_gen(T, n, min(k, 8), [1, 2], seed=seed, dtype=np.float64)
    Yd = d['Y'] + 0.05 * rng.randn(T, n)
    if missing:
        mask = rng.rand(T, n) < rng.choice([0.05, 0.3, 0.9])
        mask[rng.randint(T)] = False                     # an empty timestamp
        if n > 2:
            mask[:, rng.randint(n)] = False              # an empty item
        Y = smat.csr_matrix(np.where(mask, Yd, 0.0).astype(dtype))
        Y.eliminate_zeros()
    else:
        kind = seed % 4
        Y = np.asfortranarray(Yd.astype(dtype)) if kind == 0 else np.ascontiguousarray(Yd.astype(dtype))
        if kind == 1:
            Y = smat.csr_matrix(np.where(rng.rand(T, n) < 0.4, Y, 0).astype(dtype))
    return dtype, k, lags.astype(np.uint32), Y, missing, hyper


@pytest.mark.parametrize('seed', range(36))
def test_random_shapes_vs_restatement(seed):
    dtype, k, lags, Y, missing, hyper = _case(seed)
    m0 = trmf.Model.initialize(Y, lags, k, seed=seed, dtype=dtype)
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    log = O.train_port(Y, m0.lag_set, W, H, Th, hyper, max_iter=2, missing=missing, threads=4)
    model = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
    trmf.train(Y, model, max_iter=2, missing=missing, **hyper)
    tol = TOL[np.dtype(dtype).name]
    fac = 5 * tol['factor'] if (dtype == np.float32 and seed in RELAXED_F32_SEEDS) else tol['factor']
    what = 'seed %d: %s k=%d lags=%s T=%d n=%d missing=%s %s' % (seed, np.dtype(dtype).name, k, lags.tolist(), Y.shape[0], Y.shape[1],
                                                              missing, 'sparse' if smat.issparse(Y) else 'dense')
    worst = max(relfro(model.W, W), relfro(model.H, H), relfro(model.lag_val, Th) / 10)
    print(what, '| relfro W %.1e H %.1e Th %.1e | FUZZ-MARGIN seed %d %s: %.3f of the direct gate' % (
        relfro(model.W, W), relfro(model.H, H), relfro(model.lag_val, Th), seed, np.dtype(dtype).name, worst / tol['factor']))
    assert np.all(np.isfinite(model.W)) and np.all(np.isfinite(model.H)) and np.all(np.isfinite(model.lag_val)), what
    assert relfro(model.W, W) < fac and relfro(model.H, H) < fac and relfro(model.lag_val, Th) < 10 * fac, what


@pytest.mark.parametrize('seed', range(8))
def test_random_period_gating_vs_restatement(seed):
    """period_W / period_H / period_Lag other than (1, 1, 2): a phase runs when iter % period == 0 with iter starting at 1
    (trmf.cpp:647-693) -- a period above max_iter switches the phase off.  Five iterations against the restatement."""
    rng = np.random.RandomState(500 + seed)
    dtype = [np.float64, np.float32][seed % 2]
    periods = tuple(int(x) for x in rng.choice([1, 2, 3, 7], size=3))
    Y = smat.random(180, 70, density=0.25, random_state=rng, format='csr', dtype=np.float64).astype(dtype)
    lags = [1, 2, 5]
    m0 = trmf.Model.initialize(Y, lags, 12, seed=seed, dtype=dtype)
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    hyper = dict(lambdaI=0.5, lambdaAR=50.0, lambdaLag=0.5)
    O.train_port(Y, m0.lag_set, W, H, Th, hyper, max_iter=5, periods=periods, threads=2)
    model = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
    trmf.train(Y, model, max_iter=5, period_W=periods[0], period_H=periods[1], period_Lag=periods[2], missing=True, **hyper)
    fac = TOL[np.dtype(dtype).name]['factor'] * (5 if (dtype == np.float32 and seed in RELAXED_F32_PERIOD_SEEDS) else 1)
    print('FUZZ-MARGIN period seed %d %s periods %s: %.3f of the direct gate' % (seed, np.dtype(dtype).name, periods, max(
        relfro(model.W, W), relfro(model.H, H), relfro(model.lag_val, Th) / 10) / TOL[np.dtype(dtype).name]['factor']))
    assert relfro(model.W, W) < fac and relfro(model.H, H) < fac and relfro(model.lag_val, Th) < 10 * fac, periods
    if periods[0] > 5: assert np.array_equal(model.W, m0.W)
    if periods[1] > 5: assert np.array_equal(model.H, m0.H)
    if periods[2] > 5: assert np.array_equal(model.lag_val, m0.lag_val)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('T,lags,k,ar_ti,missing', [
    (1100, [1, 2, 3, 4, 7, 200], 8, None, True),                            # one tile, TI + max lag > 1024: two-buffer layout, two passes
    (3000, [1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 30], 20, None, True),      # runs of four between single lags, one-pass tiles
    (3000, [1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 30], 20, 64, False),       # the same with many small tiles, shared-Gram product
    (2500, [2, 3, 4, 5, 400], 33, None, True),                              # long reach, k = 33 (KP = 48: six column groups)
    (1500, list(range(1, 17)) + [97, 98, 99, 100], 60, 944, False),         # forced TI beyond one pass (944 + 100 > 1024), 155 KB of LDS in fp64
])
def test_unfused_ar_tile_layouts(dtype, T, lags, k, ar_ti, missing, monkeypatch):
    """The unfused CG's AR tile kernel in each of its forms: lag sets split into runs of four consecutive lags and single
    lags (ar_lag_steps), residual rows reusing the operand rows' LDS (one pass) or not, one tile or many -- three ALS
    iterations against the restatement at the 8(d) gates."""
    monkeypatch.setenv('TRMF_NO_HV_TILE', '1')
    if ar_ti is not None:
        monkeypatch.setenv('TRMF_AR_TI', str(ar_ti))
    n = 40
    rng = np.random.RandomState(7)
    d = trmf.Model.syn_gen(T, n, 6, [1, 2], seed=7, dtype=np.float64)
    Yd = (d['Y'] + 0.05 * rng.randn(T, n)).astype(dtype)
    Y = smat.csr_matrix(np.where(rng.rand(T, n) < 0.5, Yd, 0)) if missing else np.ascontiguousarray(Yd)
    hyper = dict(lambdaI=0.5, lambdaAR=50.0, lambdaLag=0.5)
    m0 = trmf.Model.initialize(Y, lags, k, seed=3, dtype=dtype)
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    O.train_port(Y, m0.lag_set, W, H, Th, hyper, max_iter=3, missing=missing, threads=4)
    model = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
    trmf.train(Y, model, max_iter=3, missing=missing, **hyper)
    fac = TOL[np.dtype(dtype).name]['factor']
    print('T=%d lags=%s k=%d TI=%s %s: relfro W %.1e H %.1e Th %.1e' % (T, lags, k, ar_ti, np.dtype(dtype).name, relfro(model.W, W), relfro(model.H, H), relfro(model.lag_val, Th)))
    assert relfro(model.W, W) < fac and relfro(model.H, H) < fac and relfro(model.lag_val, Th) < 10 * fac


@pytest.mark.parametrize('dtype,k,nlag,T,n', [
    (np.float64, 65, 5, 300, 120),        # one past the register-tiled kernels
    (np.float32, 80, 8, 260, 150),
    (np.float64, 128, 3, 200, 90),
    (np.float32, 200, 4, 240, 70),
    (np.float64, 256, 2, 150, 40),        # the last rank with one thread per column in apply_kernel
    (np.float32, 300, 2, 120, 30),        # apply_wide_kernel
    (np.float64, 520, 1, 90, 24),
    (np.float64, 8, 160, 700, 60),        # |L| = 160: the |L| x |L| Theta systems no longer fit LDS in fp64
    (np.float32, 80, 160, 600, 50),       # both at once
    (np.float32, 12, 230, 900, 40),       # past the fp32 LDS limit as well
])
def test_ranks_and_lag_sets_beyond_the_tiled_kernels(dtype, k, nlag, T, n):
    """the reference computes for any rank (k x k scratch per thread, trmf.cpp:362-365) and any lag set
    (trmf.cpp:79-147, 425-484); up to round 3 the drop-in answered `[ERR MSG]` above k = 64 or 128 lags.  Ranks 65..1024 run the
    generic kernels (csrc/generic_kernels.hpp) + the unfused CG, long lag sets keep the Theta systems in global scratch.
    Two ALS iterations against the restatement at the SURVEY.md 8(d) gates."""
    rng = np.random.RandomState(77 + k + nlag)
    d = trmf.Model.syn_gen(T, n, 6, [1, 2], seed=k, dtype=np.float64)
    Yd = d['Y'] + 0.05 * rng.randn(T, n)
    mask = rng.rand(T, n) < 0.5
    mask[rng.randint(T)] = False
    mask[:, rng.randint(n)] = False
    Y = smat.csr_matrix(np.where(mask, Yd, 0.0).astype(dtype))
    Y.eliminate_zeros()
    lags = np.arange(1, nlag + 1, dtype=np.uint32)
    hyper = dict(lambdaI=0.5, lambdaAR=50.0, lambdaLag=0.5)
    m0 = trmf.Model.initialize(Y, lags, k, seed=3, dtype=dtype)
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    log = O.train_port(Y, m0.lag_set, W, H, Th, hyper, max_iter=2, missing=True, threads=4)
    model = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
    with trmf.session.Session(Y, model, missing=True, **hyper) as s:
        s.run(2); st = s.stats(2); s.download(); desc = s.describe()
    tol = TOL[np.dtype(dtype).name]
    print('k=%d |L|=%d %s (%s): relfro W %.1e H %.1e Th %.1e; CG %s vs %s' % (k, nlag, np.dtype(dtype).name, desc, relfro(model.W, W), relfro(model.H, H),
                                                                      relfro(model.lag_val, Th), [x['cg_iter'] for x in st], [l['cg_iter'] for l in log]))
    assert np.all(np.isfinite(model.W)) and np.all(np.isfinite(model.H)) and np.all(np.isfinite(model.lag_val))
    assert relfro(model.W, W) < tol['factor'] and relfro(model.H, H) < tol['factor'] and relfro(model.lag_val, Th) < 10 * tol['factor']
    assert all(abs(a['cg_iter'] - b['cg_iter']) <= 1 for a, b in zip(st, log))
    assert ('generic' in desc) == (k > 64)


@pytest.mark.parametrize('dtype,k,kind', [(np.float64, 70, 'dense_c'), (np.float32, 96, 'dense_f'), (np.float64, 130, 'sparse'), (np.float32, 300, 'dense_c')])
def test_full_observation_path_beyond_rank_64(dtype, k, kind):
    """missing = 0 above rank 64: Y^T W / Y H, W^T W + lambda I, one Cholesky with n right-hand sides, the CG with one shared
    Gram -- generic kernels (csrc/generic_kernels.hpp) in place of the MFMA ones.  Two iterations against the restatement."""
    rng = np.random.RandomState(5 + k)
    T, n, lags = 180, 50, np.array([1, 2, 5], dtype=np.uint32)
    d = trmf.Model.syn_gen(T, n, 6, [1, 2], seed=k, dtype=np.float64)
    Yd = (d['Y'] + 0.05 * rng.randn(T, n)).astype(dtype)
    Y = {'dense_c': np.ascontiguousarray(Yd), 'dense_f': np.asfortranarray(Yd), 'sparse': smat.csr_matrix(np.where(rng.rand(T, n) < 0.5, Yd, 0))}[kind]
    hyper = dict(lambdaI=0.5, lambdaAR=50.0, lambdaLag=0.5)
    m0 = trmf.Model.initialize(Y, lags, k, seed=1, dtype=dtype)
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    O.train_port(Y, m0.lag_set, W, H, Th, hyper, max_iter=2, missing=False, threads=4)
    model = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
    trmf.train(Y, model, max_iter=2, missing=False, **hyper)
    tol = TOL[np.dtype(dtype).name]
    print('full path k=%d %s %s: relfro W %.1e H %.1e Th %.1e' % (k, kind, np.dtype(dtype).name, relfro(model.W, W), relfro(model.H, H), relfro(model.lag_val, Th)))
    assert relfro(model.W, W) < tol['factor'] and relfro(model.H, H) < tol['factor'] and relfro(model.lag_val, Th) < 10 * tol['factor']


def test_limits_that_remain_are_reported(capfd):
    """What the drop-in still refuses, loudly and without touching the outputs: a rank above 1024, more than 1024 lags."""
    rng = np.random.RandomState(0)
    T, n = 60, 30
    Yd = rng.randn(T, n)
    for k, nlag, missing, what in ((1025, 2, True, 'rank k=1025'), (1025, 2, False, 'rank k=1025'), (4, 1025, True, '|lag_set|=1025')):
        Y = smat.csr_matrix(Yd) if missing else Yd
        lags = np.arange(1, nlag + 1, dtype=np.uint32) if nlag < T else np.arange(nlag, dtype=np.uint32)
        m = trmf.Model.initialize(Y, lags, k, seed=0, dtype=np.float64)
        W0 = m.W.copy()
        trmf.train(Y, m, max_iter=1, missing=missing, lambdaI=0.5, lambdaAR=1.0, lambdaLag=0.5)
        err = capfd.readouterr().err
        assert '[ERR MSG]' in err and what in err, err
        assert np.array_equal(m.W, W0)
