"""GPU (-m gpu): the split path of long rows (csrc/gram_kernels.hpp "split rows").

The reference schedules rows dynamically (trmf.cpp:371 `schedule(dynamic,64)`, :234,252,273 `schedule(dynamic,32)`) and has no
cliff at a long row; here rows above a threshold are cut into items whose partial Grams are summed in item order.  Checked against
the oracle (C restatement of the reference, CSR-order sums) at the usual gates (helpers.TOL):
  * forced geometry on small problems (TRMF_TEST knobs TRMF_LONG_ROW / TRMF_LONG_CHUNK): every row split into several items,
    every rank class / element type / right-hand-side form, both X-solve paths (full and packed Gram cache);
  * the default rule on a small skewed problem (a few complete series and census timestamps among short rows);
  * the new bench workloads at full size: `imp` (26 304 x 370, 80 % observed: 370 item rows of ~21 000 entries) and `zipf`
    (config 3's size, power-law row lengths, 10 000-entry item rows and 100 000-entry timestamp rows);
  * the uniform workloads never take the path: results bit-identical with the path switched off.
"""
import os

import numpy as np
import pytest

import oracle_py as O
import trmf
from helpers import TOL, assert_within_fp32_noise, evidence, fp32_noise_yardstick, make_model, relfro, relmax
from trmf import session, synth

pytestmark = pytest.mark.gpu
NCPU = os.cpu_count() or 8
BIG = 10 ** 6


def run_product(Y, lag_set, W0, H0, Th0, hyper, max_iter, periods=(1, 1, 2)):
    model = make_model(W0, H0, Th0, lag_set)
    trmf.train(Y, model, max_iter=max_iter, period_W=periods[0], period_H=periods[1], period_Lag=periods[2], missing=True, **hyper)
    return model


def run_oracle(Y, lag_set, W0, H0, Th0, hyper, max_iter, periods=(1, 1, 2)):
    W, H, Th = W0.copy(), H0.copy(), np.asfortranarray(Th0.copy())
    # (a small problem on all 256 hardware threads of the GPU box spends its time entering OpenMP regions: 65 s per forced-split case)
    threads = NCPU if Y.nnz > 2000000 else min(8, NCPU)
    log = O.train_port(Y, lag_set, W, H, Th, hyper, max_iter=max_iter, periods=periods, threads=threads)
    return W, H, Th, log


def describe_of(Y, model, hyper):
    with session.Session(Y, model, missing=True, **hyper) as s:
        return s.describe()


CASES = [(np.float32, 40, 16), (np.float32, 16, 4), (np.float32, 8, 3), (np.float32, 24, 5), (np.float32, 32, 2), (np.float32, 48, 6),
         (np.float32, 56, 4), (np.float32, 64, 8), (np.float64, 64, 8), (np.float64, 24, 4), (np.float64, 40, 16), (np.float64, 13, 3)]


@pytest.mark.parametrize('path', ['fused', 'unfused'])
@pytest.mark.parametrize('dtype,k,nlag', CASES)
def test_forced_split_every_row_vs_oracle(dtype, k, nlag, path, monkeypatch):
    """Every row of both orientations is split (threshold 24 entries, items of 32): F-solve alone and X-solve alone at the direct-solve
    gates, then 3 full iterations at the parity gates."""
    monkeypatch.setenv('TRMF_LONG_ROW', '24')
    monkeypatch.setenv('TRMF_LONG_CHUNK', '32')
    if path == 'unfused':
        monkeypatch.setenv('TRMF_NO_HV_TILE', '1')          # packed Gram cache: gram_x_long_kernel<.., PACKED = true>
    p = synth.sparse_problem(n=700, T=520, k=k, nlag=nlag, density=0.2, dtype=dtype, seed=11)
    Y, lags = p['Y'], p['lag_set']
    m0 = synth.initial_model(Y, lags, k, seed=11)
    tol = TOL[np.dtype(dtype).name]
    tight = 1e-6 if dtype == np.float64 else 2e-4
    # F-solve only
    mf = run_product(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 1, periods=(BIG, 1, BIG))
    _, Hf, _, _ = run_oracle(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 1, periods=(BIG, 1, BIG))
    assert relmax(mf.H, Hf) < tight
    # X-solve only (Gram cache of every timestamp from summed partials, then the CG)
    mx = run_product(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 1, periods=(1, BIG, BIG))
    Wx, _, _, logx = run_oracle(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 1, periods=(1, BIG, BIG))
    assert relfro(mx.W, Wx) < (1e-6 if dtype == np.float64 else tol['factor'])
    # three full iterations
    m = run_product(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 3)
    W, H, Th, _ = run_oracle(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 3)
    Jo = O.objective(Y, lags, W, H, Th, synth.HYPER)
    Jp = O.objective(Y, lags, m.W, m.H, m.lag_val, synth.HYPER)
    direct = abs(Jp - Jo) / Jo < tol['objective'] and relfro(m.H, H) < tol['factor'] and relfro(m.W, W) < tol['factor']
    if not direct:
        # fp32 only: the truncated CG's noise floor, measured on the reference side on the same inputs (helpers.fp32_noise_yardstick)
        assert dtype == np.float32, (abs(Jp - Jo) / Jo, relfro(m.H, H), relfro(m.W, W))
        ys = fp32_noise_yardstick(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 3, threads=min(8, NCPU))
        assert_within_fp32_noise(m, ys, lags, synth.HYPER, what='forced split k=%d |L|=%d %s' % (k, nlag, path))
    d = describe_of(Y, make_model(m0.W, m0.H, m0.lag_val, lags), synth.HYPER)
    assert 'split rows' in d and 'F 700 rows' in d and 'X 520 rows' in d, d


def test_forced_split_is_deterministic_and_differs_only_by_rounding(monkeypatch):
    """Two runs of the split path are bit-identical (fixed summation order); against the row kernels the factors differ by rounding only."""
    p = synth.sparse_problem(n=900, T=640, k=40, nlag=16, density=0.15, dtype=np.float32, seed=3)
    Y, lags = p['Y'], p['lag_set']
    m0 = synth.initial_model(Y, lags, 40, seed=3)
    base = run_product(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 2)
    monkeypatch.setenv('TRMF_LONG_ROW', '48')
    monkeypatch.setenv('TRMF_LONG_CHUNK', '48')
    a = run_product(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 2)
    b = run_product(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 2)
    assert np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H) and np.array_equal(a.lag_val, b.lag_val)
    assert not np.array_equal(a.H, base.H)                  # a different summation order really ran
    assert relfro(a.H, base.H) < 1e-4 and relfro(a.W, base.W) < 1e-3


@pytest.mark.parametrize('cfgname', ['small40', 'tiny'])
def test_uniform_workloads_do_not_take_the_split_path(cfgname, monkeypatch):
    """BASELINE's uniform patterns stay on the row kernels: same bits with the split path switched off."""
    cfg = synth.CONFIGS[cfgname]
    p = synth.make(cfg, seed=0)
    m0 = synth.initial_model(p['Y'], p['lag_set'], cfg['k'], seed=0)
    a = run_product(p['Y'], p['lag_set'], m0.W, m0.H, m0.lag_val, synth.HYPER, 3)
    d = describe_of(p['Y'], make_model(m0.W, m0.H, m0.lag_val, p['lag_set']), synth.HYPER)
    assert 'split rows' not in d, d
    monkeypatch.setenv('TRMF_LONG_ROW', '0')
    b = run_product(p['Y'], p['lag_set'], m0.W, m0.H, m0.lag_val, synth.HYPER, 3)
    assert np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H) and np.array_equal(a.lag_val, b.lag_val)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_default_rule_on_a_small_skewed_problem(dtype):
    """No knobs: three complete series (2600 entries each) and two census timestamps (3000 entries each) among short rows -- the default
    thresholds (512 / 2048 entries, fp32; 2048 / 2048, fp64) split exactly those rows; everything else runs on the row kernels."""
    n, T, k = 3000, 2600, 40
    p = synth.powerlaw_problem(n, T, k, list(range(1, 9)), nnz0=120000, alpha_items=0.3, alpha_time=0.2, full_items=3, full_times=2,
                               dtype=dtype, seed=5)
    Y, lags = p['Y'], p['lag_set']
    m0 = synth.initial_model(Y, lags, k, seed=5)
    d = describe_of(Y, make_model(m0.W, m0.H, m0.lag_val, lags), synth.HYPER)
    assert 'split rows: F 3 rows' in d and 'X 2 rows' in d, d
    m = run_product(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 3)
    W, H, Th, _ = run_oracle(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 3)
    tol = TOL[np.dtype(dtype).name]
    Jo = O.objective(Y, lags, W, H, Th, synth.HYPER)
    Jp = O.objective(Y, lags, m.W, m.H, m.lag_val, synth.HYPER)
    assert abs(Jp - Jo) / Jo < tol['objective']
    assert relfro(m.H, H) < tol['factor'] and relfro(m.W, W) < tol['factor']
    # the split rows themselves, one F-solve: the complete series' factor rows
    mf = run_product(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 1, periods=(BIG, 1, BIG))
    _, Hf, _, _ = run_oracle(Y, lags, m0.W, m0.H, m0.lag_val, synth.HYPER, 1, periods=(BIG, 1, BIG))
    long_items = np.flatnonzero(np.diff(Y.tocsc().indptr) >= 2048)
    assert len(long_items) == 3
    assert relmax(mf.H[long_items], Hf[long_items]) < (1e-6 if dtype == np.float64 else 2e-4)


@pytest.mark.parametrize('cfgname', ['imp', 'zipf', 'imp60'])
def test_long_row_workloads_full_size_vs_oracle(cfgname):
    """The bench workloads of this path at full size, 2 ALS iterations from the random start vs the restatement on all host cores."""
    cfg = synth.CONFIGS[cfgname]
    p = synth.make(cfg, seed=0)
    Y, lags = p['Y'], p['lag_set']
    hyper = dict(cfg.get('hyper', synth.HYPER))
    m0 = synth.initial_model(Y, lags, cfg['k'], seed=0)
    iters = 2
    W, H, Th, log = run_oracle(Y, lags, m0.W, m0.H, m0.lag_val, hyper, iters)
    model = make_model(m0.W, m0.H, m0.lag_val, lags)
    with session.Session(Y, model, missing=True, **hyper) as s:
        d = s.describe()
        s.run(iters); st = s.stats(iters); s.download()
    Jo = O.objective(Y, lags, W, H, Th, hyper)
    Jp = O.objective(Y, lags, model.W, model.H, model.lag_val, hyper)
    cg_o, cg_p = [l['cg_iter'] for l in log], [x['cg_iter'] for x in st]
    evidence('%s full size (%s): J oracle %.10g gpu %.10g rel %.2e; relfro W %.2e H %.2e Th %.2e; CG oracle %s gpu %s' % (
        cfgname, d, Jo, Jp, abs(Jp - Jo) / Jo, relfro(model.W, W), relfro(model.H, H), relfro(model.lag_val, Th), cg_o, cg_p))
    assert 'split rows' in d
    if not (abs(Jp - Jo) / Jo < 1e-5 and relfro(model.H, H) < 1e-3 and relfro(model.W, W) < 1e-3):
        # (fp32: the measured noise floor of the truncated CG on these inputs instead -- helpers.fp32_noise_yardstick)
        ys = fp32_noise_yardstick(Y, lags, m0.W, m0.H, m0.lag_val, hyper, iters, with_ref=False)
        assert_within_fp32_noise(model, ys, lags, hyper, what=cfgname + ' full size')
    assert all(abs(a - b) <= 1 for a, b in zip(cg_o, cg_p))
    # one F-solve from the random start, the long rows alone (direct solve: tight gate)
    mf = run_product(Y, lags, m0.W, m0.H, m0.lag_val, hyper, 1, periods=(BIG, 1, BIG))
    _, Hf, _, _ = run_oracle(Y, lags, m0.W, m0.H, m0.lag_val, hyper, 1, periods=(BIG, 1, BIG))
    long_items = np.flatnonzero(np.diff(Y.tocsc().indptr) >= 512)
    evidence('%s full size: one F-solve, %d split item rows: relmax(H rows) %.2e; all rows %.2e' % (
        cfgname, len(long_items), relmax(mf.H[long_items], Hf[long_items]), relmax(mf.H, Hf)))
    assert relmax(mf.H[long_items], Hf[long_items]) < 2e-4 and relmax(mf.H, Hf) < 2e-4


@pytest.mark.parametrize('dtype,k', [(np.float32, 40), (np.float64, 24), (np.float32, 16)])
def test_full_observation_path_with_a_sparse_Y_splits_long_rows_too(dtype, k, monkeypatch):
    """missing = 0 with a SPARSE Y (zeros are observations; trmf.cpp:299-351, 155-215 through gmat_x_dmat): the products Y^T W and Y H are
    one wavefront per row in spmm_rows_kernel; rows above the threshold go through spmm_part_kernel + spmm_reduce_kernel (item order).
    Forced geometry (every row split) and, second, the default rule on a pattern with complete series / census timestamps."""
    p = synth.sparse_problem(n=700, T=520, k=k, nlag=4, density=0.2, dtype=dtype, seed=17)
    q = synth.powerlaw_problem(3000, 2600, k, [1, 2, 3, 4], nnz0=120000, alpha_items=0.3, alpha_time=0.2, full_items=3, full_times=2, dtype=dtype, seed=5)
    tol = TOL[np.dtype(dtype).name]
    for prob, forced in ((p, True), (q, False)):
        Y, lags = prob['Y'], prob['lag_set']
        m0 = synth.initial_model(Y, lags, k, seed=17)
        if forced:
            monkeypatch.setenv('TRMF_LONG_ROW', '24'); monkeypatch.setenv('TRMF_LONG_CHUNK', '32')
        else:
            monkeypatch.delenv('TRMF_LONG_ROW', raising=False); monkeypatch.delenv('TRMF_LONG_CHUNK', raising=False)
        model = make_model(m0.W, m0.H, m0.lag_val, lags)
        with session.Session(Y, model, missing=False, **synth.HYPER) as s:
            d = s.describe()
            s.run(3); s.download()
        assert 'split rows' in d, d
        W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
        O.train_port(Y, lags, W, H, Th, synth.HYPER, max_iter=3, missing=False, threads=min(8, NCPU))
        assert relfro(model.H, H) < tol['factor'] and relfro(model.W, W) < tol['factor'] and relfro(model.lag_val, Th) < tol['factor'] * 10, \
            (forced, relfro(model.H, H), relfro(model.W, W))
        if not forced:      # and the same bits as the row kernels' single chains up to rounding: switch the path off
            monkeypatch.setenv('TRMF_LONG_ROW', '0')
            base = make_model(m0.W, m0.H, m0.lag_val, lags)
            trmf.train(Y, base, max_iter=3, missing=False, **synth.HYPER)
            assert relfro(model.H, base.H) < tol['factor'] and not np.array_equal(model.H, base.H)
