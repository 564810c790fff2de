"""Seeded synthetic TRMF workloads: low-rank + AR temporal structure,
observed on a uniformly random sparse pattern.  Mirrors ``Model.syn_gen`` (reference
python/trmf/trmf.py:195-220) without materialising the dense T x n matrix.

Used by bench.py, the parity tests and the golden-vector script; pure NumPy/SciPy.
"""
import numpy as np
import scipy.sparse as smat


def sparse_problem(n, T, k, nlag, density, dtype=np.float32, seed=0, noise=0.01, chunk=1 << 20):
    """Return dict(Y=csr T x n, lag_set=uint32[nlag], X, F, Theta) for a seeded synthetic problem."""
    rng = np.random.RandomState(seed)
    lag_set = np.arange(1, nlag + 1, dtype=np.uint32)
    midx = int(lag_set[-1]) if nlag else 0
    theta = rng.randn(nlag, k)
    theta = theta / (np.abs(theta).sum(axis=0, keepdims=True) + 0.1)
    X = np.zeros((T, k))
    X[:max(midx, 1)] = rng.randn(max(midx, 1), k)
    eps = noise * rng.randn(T, k)
    lags = lag_set.astype(np.int64)
    for i in range(max(midx, 1), T):
        X[i] = (theta * X[i - lags]).sum(axis=0) + eps[i] if nlag else rng.randn(k)
    F = rng.randn(n, k)
    nnz0 = int(n * T * density)
    rows = rng.randint(0, T, size=nnz0)
    cols = rng.randint(0, n, size=nnz0)
    vals = np.empty(nnz0)
    for s in range(0, nnz0, chunk):
        e = min(nnz0, s + chunk)
        vals[s:e] = np.einsum('ij,ij->i', X[rows[s:e]], F[cols[s:e]])
    vals += noise * rng.randn(nnz0)
    Y = smat.coo_matrix((vals, (rows, cols)), shape=(T, n)).tocsr()   # duplicates are summed
    Y.sort_indices()
    return {'Y': Y.astype(dtype), 'lag_set': lag_set, 'X': X, 'F': F, 'Theta': theta}


def _latent_factors(rng, n, T, k, lag_set, noise):
    """Temporal factor X (AR over lag_set), item factor F and Theta of the generators below (same recipe as sparse_problem)."""
    lag_set = np.asarray(lag_set, dtype=np.uint32)
    nlag = len(lag_set)
    midx = int(lag_set[-1]) if nlag else 0
    theta = rng.randn(nlag, k)
    theta = theta / (np.abs(theta).sum(axis=0, keepdims=True) + 0.1)
    X = np.zeros((T, k))
    X[:max(midx, 1)] = rng.randn(max(midx, 1), k)
    eps = noise * rng.randn(T, k)
    lags = lag_set.astype(np.int64)
    for i in range(max(midx, 1), T):
        X[i] = (theta * X[i - lags]).sum(axis=0) + eps[i] if nlag else rng.randn(k)
    return X, rng.randn(n, k), theta


def pattern_problem(n, T, k, lag_set, keys, dtype=np.float32, seed=0, noise=0.01, chunk=1 << 20):
    """Synthetic problem observed on a GIVEN pattern: `keys` = sorted unique cell numbers row * n + col of the T x n matrix."""
    rng = np.random.RandomState(seed + 1)
    X, F, theta = _latent_factors(rng, n, T, k, lag_set, noise)
    rows, cols = keys // n, keys % n
    vals = np.empty(len(keys))
    for s in range(0, len(keys), chunk):
        e = min(len(keys), s + chunk)
        vals[s:e] = np.einsum('ij,ij->i', X[rows[s:e]], F[cols[s:e]])
    vals += noise * rng.randn(len(keys))
    indptr = np.zeros(T + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=T), out=indptr[1:])
    Y = smat.csr_matrix((vals.astype(dtype), cols.astype(np.int32), indptr), shape=(T, n))   # keys sorted: rows ascending, columns sorted
    return {'Y': Y, 'lag_set': np.asarray(lag_set, dtype=np.uint32), 'X': X, 'F': F, 'Theta': theta}


def imputation_problem(n, T, k, lag_set, observed=0.8, dtype=np.float32, seed=0, noise=0.01):
    """A dense panel with entries missing at random (the imputation use of the paper's data, trmf.py `missing=True`): every cell
    is observed with probability `observed`, so every item row holds ~observed * T entries -- FEW, LONG rows."""
    rng = np.random.RandomState(seed)
    keys = np.flatnonzero(rng.rand(T * n) < observed)
    return pattern_problem(n, T, k, lag_set, keys, dtype=dtype, seed=seed, noise=noise)


def powerlaw_problem(n, T, k, lag_set, nnz0, alpha_items=0.8, alpha_time=0.6, full_items=20, full_times=2, dtype=np.float32, seed=0, noise=0.01):
    """Power-law observation pattern: items and timestamps drawn from Zipf-like weights (rank^-alpha, randomly permuted), nnz0 draws
    with replacement, duplicates dropped; on top the `full_items` heaviest items are observed at EVERY timestamp and the `full_times`
    heaviest timestamps at EVERY item (complete series / census days): row lengths span 1 .. T on the item side, 1 .. n on the time side."""
    rng = np.random.RandomState(seed)
    wi = (1.0 + np.arange(n)) ** -alpha_items
    wt = (1.0 + np.arange(T)) ** -alpha_time
    pi_, pt_ = rng.permutation(n), rng.permutation(T)
    ci = np.cumsum(wi / wi.sum()); ct = np.cumsum(wt / wt.sum())
    keys = []
    step = 1 << 22
    for s in range(0, nnz0, step):
        m = min(step, nnz0 - s)
        r = pt_[np.minimum(np.searchsorted(ct, rng.rand(m)), T - 1)]
        c = pi_[np.minimum(np.searchsorted(ci, rng.rand(m)), n - 1)]
        keys.append(r.astype(np.int64) * n + c)
    for j in range(min(full_items, n)):
        keys.append(np.arange(T, dtype=np.int64) * n + pi_[j])
    for j in range(min(full_times, T)):
        keys.append(np.int64(pt_[j]) * n + np.arange(n, dtype=np.int64))
    keys = np.unique(np.concatenate(keys))
    return pattern_problem(n, T, k, lag_set, keys, dtype=dtype, seed=seed, noise=noise)


def make(cfg, seed=0):
    """Problem dict of a CONFIGS entry (sparse kinds: 'uniform' (default), 'imp', 'zipf')."""
    dtype = np.dtype(cfg['dtype'])
    lags = cfg['lags'] if 'lags' in cfg else list(range(1, cfg['nlag'] + 1))
    kind = cfg.get('kind', 'uniform')
    if cfg.get('dense'):
        return dense_problem(cfg['n'], cfg['T'], cfg['k'], lags, dtype=dtype, seed=seed)
    if kind == 'imp':
        return imputation_problem(cfg['n'], cfg['T'], cfg['k'], lags, observed=cfg['observed'], dtype=dtype, seed=seed)
    if kind == 'zipf':
        return powerlaw_problem(cfg['n'], cfg['T'], cfg['k'], lags, cfg['nnz0'], dtype=dtype, seed=seed,
                                **{key: cfg[key] for key in ('alpha_items', 'alpha_time', 'full_items', 'full_times') if key in cfg})
    return sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=dtype, seed=seed)


def initial_model(Y, lag_set, k, seed=0, dtype=None):
    """``Model.initialize`` semantics (reference trmf.py:222-251): rand W, H; randn Theta (F order)."""
    from .model import Model
    return Model.initialize(Y, lag_set, k, seed=seed, dtype=dtype)


def dense_problem(n, T, k, lag_set, dtype=np.float64, seed=0, noise=0.05):
    """Dense low-rank + AR matrix (Model.syn_gen + observation noise): stand-in for the electricity /
    traffic matrices, which are not available offline."""
    from .model import Model
    d = Model.syn_gen(T, n, k, lag_set, seed=seed, dtype=np.float64)
    Y = d['Y'] + noise * np.random.RandomState(seed).randn(T, n)
    return {'Y': np.ascontiguousarray(Y, dtype=dtype), 'lag_set': d['lag_set']}


# BASELINE.json configs (index = position in BASELINE.json:configs)
CONFIGS = {
    # config 1: electricity shape, dense, full-observation path (missing=0), hyper-parameters of
    # python/exp-scripts/run_electricity.py:9-25 (k and lags as BASELINE.json states them)
    'c1': dict(n=370, T=26304, k=4, lags=[1, 2, 3], dense=True, dtype='float64',
               hyper=dict(lambdaI=0.5, lambdaAR=125.0, lambdaLag=2.0)),
    # the shape python/exp-scripts/run_electricity.py:9-25 really trains at: k=60, 48 lags {1..24} u {168..191}, missing=0
    'c1p': dict(n=370, T=26304, k=60, lags=list(range(1, 25)) + list(range(168, 192)), dense=True, dtype='float64',
                hyper=dict(lambdaI=0.5, lambdaAR=125.0, lambdaLag=2.0)),
    'c2': dict(n=10000, T=5000, k=16, nlag=8, density=0.01, dtype='float32'),
    'c3': dict(n=100000, T=10000, k=40, nlag=16, density=0.01, dtype='float32'),
    'c5': dict(n=1000000, T=50000, k=64, nlag=32, density=0.001, dtype='float64'),
    # measurement aids: config 3's sizes at the fp32 ranks whose F-solve needs four column tiles (scripts/bench_fsolve.py)
    'c3k56': dict(n=100000, T=10000, k=56, nlag=16, density=0.01, dtype='float32'),
    'c3k64': dict(n=100000, T=10000, k=64, nlag=16, density=0.01, dtype='float32'),
    # skewed / long-row workloads (round 6: the split path of long rows)
    # imp: the paper's electricity panel used for IMPUTATION -- 26 304 x 370, 80 % of the cells observed, missing = 1: 370 item rows of ~21 000 entries
    'imp': dict(kind='imp', n=370, T=26304, k=40, nlag=16, observed=0.8, dtype='float32'),
    'imp60': dict(kind='imp', n=370, T=26304, k=60, lags=list(range(1, 25)) + list(range(168, 192)), observed=0.8, dtype='float32',
                  hyper=dict(lambdaI=0.5, lambdaAR=125.0, lambdaLag=2.0)),
    # zipf: config 3's size and nnz with power-law row lengths on both sides; 20 complete series (10 000 entries) and 2 census timestamps (100 000 entries)
    'zipf': dict(kind='zipf', n=100000, T=10000, k=40, nlag=16, nnz0=12300000, dtype='float32'),
    # small shapes for tests
    'tiny': dict(n=300, T=200, k=8, nlag=3, density=0.05, dtype='float32'),
    'small40': dict(n=2000, T=600, k=40, nlag=16, density=0.03, dtype='float32'),
}
HYPER = dict(lambdaI=0.5, lambdaAR=50.0, lambdaLag=0.5)
