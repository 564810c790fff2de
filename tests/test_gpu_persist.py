"""The persistent CG kernel (csrc/cg_persist.hpp: one cooperative launch per X-solve, workgroups exchanging tagged records
through memory) against the launch-per-step path it replaces on one GPU: same tiles, same element mapping, same summation
order, so the factors, the CG counts and every number of the TRON line must be BIT-IDENTICAL.  The other parity tests
(goldens, fuzz, full size) run through the persistent kernel by default and compare it with the oracle; the launch-per-step
kernels stay covered here, by the multi-rank tests and by the multi-wave grids of test_gpu_fullsize.py."""
import numpy as np
import pytest

from helpers import make_model

pytestmark = pytest.mark.gpu


def _run(p, m0, dtype, iters, missing, monkeypatch, persist, tile=None):
    from trmf import session, synth
    if tile is None:
        monkeypatch.delenv('TRMF_TILE', raising=False)
    else:
        monkeypatch.setenv('TRMF_TILE', tile)
    if persist:
        monkeypatch.delenv('TRMF_PERSIST', raising=False)
    else:
        monkeypatch.setenv('TRMF_PERSIST', '0')
    model = make_model(m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
    Y = p['Y'].astype(dtype)
    with session.Session(Y, model, missing=missing, **synth.HYPER) as s:
        s.run(iters); st = s.stats(iters); s.download(); desc = s.describe()
    return model, st, desc


@pytest.mark.parametrize('shape', [
    dict(n=900, T=400, k=12, nlag=4, density=0.06),          # two column tiles
    dict(n=701, T=353, k=5, nlag=3, density=0.08),           # odd rank, short last tile
    dict(n=3000, T=1200, k=40, nlag=16, density=0.04),       # config 3 / 4's rank and lag set
    dict(n=2000, T=2500, k=16, nlag=8, density=0.02),        # config 2's
    dict(n=1500, T=700, k=64, nlag=6, density=0.05),         # the widest Gram slice (8-byte Gram loads in fp32)
    dict(n=400, T=300, k=8, nlag=0, density=0.1),            # no lags at all
    dict(n=600, T=260, k=24, nlag=5, density=0.08, lags=[0, 1, 2, 7, 24]),     # lag 0 is legal (trmf.py:354)
])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_persistent_kernel_bit_identical_to_launch_per_step(shape, dtype, monkeypatch):
    from trmf import synth
    c = dict(shape)
    lags = c.pop('lags', None)
    p = synth.sparse_problem(n=c['n'], T=c['T'], k=c['k'], nlag=c['nlag'], density=c['density'], dtype=np.float64, seed=5)
    if lags is not None:
        p['lag_set'] = np.array(lags, dtype=np.uint32)
    m0 = synth.initial_model(p['Y'], p['lag_set'], c['k'], seed=5)
    iters = 4
    a, sa, da = _run(p, m0, dtype, iters, True, monkeypatch, persist=True)
    b, sb, db = _run(p, m0, dtype, iters, True, monkeypatch, persist=False)
    if 'unfused' in da:
        pytest.skip('lag reach does not fit the fused tile: ' + da)
    assert 'persistent' in da and 'one launch per CG step' in db, (da, db)
    assert np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H) and np.array_equal(a.lag_val, b.lag_val)
    for x, y in zip(sa, sb):
        for key in ('f', 'fnew', 'actred', 'prered', 'gnorm', 'cg_rnorm', 'cg_iter', 'accepted', 'delta', 'normF', 'normX', 'normLV'):
            assert x[key] == y[key], (key, x[key], y[key])


@pytest.mark.parametrize('dense', [False, True])
def test_persistent_kernel_full_observation_path(dense, monkeypatch):
    """missing = 0: one shared Gram for every timestamp (gstride 0) -- the fused CG covers it when the lag reach is short."""
    from trmf import synth
    p = synth.dense_problem(300, 500, 6, [1, 2, 3], dtype=np.float64, seed=3)
    if not dense:
        import scipy.sparse as smat
        p = dict(p, Y=smat.csr_matrix(p['Y']))
    m0 = synth.initial_model(p['Y'], p['lag_set'], 6, seed=3)
    for dtype in (np.float32, np.float64):
        a, sa, da = _run(p, m0, dtype, 4, False, monkeypatch, persist=True)
        b, sb, db = _run(p, m0, dtype, 4, False, monkeypatch, persist=False)
        assert 'persistent' in da and 'one launch per CG step' in db, (da, db)
        assert np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H) and np.array_equal(a.lag_val, b.lag_val)
        assert [x['cg_iter'] for x in sa] == [x['cg_iter'] for x in sb]


def test_persistent_kernel_early_stop_and_iteration_cap(monkeypatch):
    """Both ends of the CG: a start so close to the optimum that the gradient (or the first step) meets the tolerance, and an
    ill-conditioned start that runs into the 20-step cap."""
    from trmf import session, synth
    p = synth.sparse_problem(n=800, T=300, k=16, nlag=4, density=0.08, dtype=np.float64, seed=9)
    m0 = synth.initial_model(p['Y'], p['lag_set'], 16, seed=9)
    for dtype in (np.float32, np.float64):
        a, sa, _ = _run(p, m0, dtype, 12, True, monkeypatch, persist=True)
        b, sb, _ = _run(p, m0, dtype, 12, True, monkeypatch, persist=False)
        assert np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H)
        cg = [x['cg_iter'] for x in sa]
        assert cg == [x['cg_iter'] for x in sb]
        assert max(cg) == 20 and min(cg) < 20, cg


def _run_quiet(p, m0, dtype, iters, missing, monkeypatch, env):
    """log_norms off (the mode c_trmf_train and bench.py run in): the Theta-solve may move to its own stream."""
    from trmf import session, synth
    for key in ('TRMF_NO_OVERLAP', 'TRMF_OVERLAP_ALWAYS', 'TRMF_NO_CG_FOLLOW'):
        monkeypatch.delenv(key, raising=False)
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    model = make_model(m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
    with session.Session(p['Y'].astype(dtype), model, missing=missing, log_norms=False, **synth.HYPER) as s:
        # two calls: the last iteration of a call never leaves a Theta-solve outstanding, the first iterations do
        s.run(iters - 2); s.run(2); st = s.stats(iters); s.download(); desc = s.describe()
    return model, st, desc


@pytest.mark.parametrize('case', ['sparse_fused', 'sparse_long_reach', 'dense_long_reach'])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_theta_on_its_own_stream_and_followed_stop_change_nothing(case, dtype, monkeypatch):
    """Round 5: the Theta-solve of iteration t under the F-solve of iteration t + 1 (its own stream, session_state.hpp) and the
    unfused CG that stops enqueuing once the device reports the stop (cg_note) are scheduling changes: same kernels, same operands,
    so factors, CG counts and the TRON numbers must be bit-identical with both switched off."""
    from trmf import synth
    if case == 'sparse_fused':
        p = synth.sparse_problem(n=900, T=400, k=12, nlag=4, density=0.06, dtype=np.float64, seed=11)
        missing = True
    elif case == 'sparse_long_reach':                      # lag reach 191: the unfused two-kernel CG step
        p = synth.sparse_problem(n=500, T=1500, k=8, nlag=4, density=0.05, dtype=np.float64, seed=12)
        p['lag_set'] = np.array([1, 2, 24, 191], dtype=np.uint32)
        missing = True
    else:                                                  # the paper scripts' form: dense Y, missing = 0, long reach
        p = synth.dense_problem(60, 1200, 6, [1, 2, 24, 168, 191], dtype=np.float64, seed=13)
        missing = False
    k = 12 if case == 'sparse_fused' else 8 if case == 'sparse_long_reach' else 6
    m0 = synth.initial_model(p['Y'], p['lag_set'], k, seed=7)
    iters = 6
    a, sa, da = _run_quiet(p, m0, dtype, iters, missing, monkeypatch, {'TRMF_OVERLAP_ALWAYS': '1'})
    b, sb, db = _run_quiet(p, m0, dtype, iters, missing, monkeypatch, {'TRMF_NO_OVERLAP': '1', 'TRMF_NO_CG_FOLLOW': '1'})
    if case != 'sparse_fused':
        assert 'unfused' in da, da
    assert np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H) and np.array_equal(a.lag_val, b.lag_val)
    for x, y in zip(sa, sb):
        for key in ('f', 'fnew', 'actred', 'prered', 'gnorm', 'cg_rnorm', 'cg_iter', 'accepted', 'delta'):
            assert x[key] == y[key], (key, x[key], y[key])
    assert min(x['cg_iter'] for x in sa) < 20            # the stop was there to be followed


@pytest.mark.parametrize('shape', [
    dict(n=900, T=400, k=12, nlag=4, density=0.06),
    dict(n=701, T=353, k=5, nlag=3, density=0.08),           # odd rank, short last tile
    dict(n=3000, T=1200, k=40, nlag=16, density=0.04),       # config 3's rank and lag set: 51-row tiles
    dict(n=1500, T=700, k=64, nlag=6, density=0.05),         # the widest Gram slice
    dict(n=400, T=300, k=8, nlag=0, density=0.1),            # no lags
    dict(n=1200, T=2600, k=40, nlag=16, density=0.02),       # more wide tiles than one chunk of records per wavefront
])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_wide_tiles_both_cg_forms_bit_identical_and_equal_to_narrow_up_to_rounding(shape, dtype, monkeypatch):
    """Round 5: wide tiles (512 threads, one workgroup per CU; chosen by one rank when the narrow tiles outnumber the CUs -- config 3 at
    full size, tests/test_gpu_fullsize.py runs through them).  Forced here at small shapes: the persistent kernel and the
    launch-per-step path must be bit-identical to each other in the wide geometry too (cg_persist_kernel<KQ, false, 512> against
    hv_tile_kernel<MODE, KQ, false, 512> / cg_close_kernel<false, 512> / accept_tile_kernel<512>), and the wide geometry must give
    the narrow one's iterates up to the rounding of differently grouped sums."""
    from trmf import synth
    c = dict(shape)
    p = synth.sparse_problem(n=c['n'], T=c['T'], k=c['k'], nlag=c['nlag'], density=c['density'], dtype=np.float64, seed=21)
    m0 = synth.initial_model(p['Y'], p['lag_set'], c['k'], seed=21)
    iters = 4
    a, sa, da = _run(p, m0, dtype, iters, True, monkeypatch, persist=True, tile='wide')
    b, sb, db = _run(p, m0, dtype, iters, True, monkeypatch, persist=False, tile='wide')
    n_, sn, dn = _run(p, m0, dtype, iters, True, monkeypatch, persist=True, tile='narrow')
    assert '512 threads' in da and 'persistent' in da and '512 threads' in db and 'one launch per CG step' in db, (da, db)
    assert '256 threads' in dn, dn
    assert np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H) and np.array_equal(a.lag_val, b.lag_val)
    for x, y in zip(sa, sb):
        for key in ('f', 'fnew', 'actred', 'prered', 'gnorm', 'cg_rnorm', 'cg_iter', 'accepted', 'delta', 'normF', 'normX', 'normLV'):
            assert x[key] == y[key], (key, x[key], y[key])
    # (measured: fp32 factors come out IDENTICAL -- the per-tile sums are fp64 and differ in their last bits only, which the cast of
    # alpha / rho to val_type removes -- fp64 1e-13 ... 1.4e-9 after four iterations of a truncated CG; scripts/wide_vs_narrow.py)
    tol = 1e-7 if dtype == np.float64 else 2e-3
    for u, v in ((a.W, n_.W), (a.H, n_.H), (a.lag_val, n_.lag_val)):
        assert np.linalg.norm(u.astype(np.float64) - v.astype(np.float64)) <= tol * np.linalg.norm(v.astype(np.float64))
    assert all(abs(x['cg_iter'] - y['cg_iter']) <= 1 for x, y in zip(sa, sn))
    assert all(abs(x['f'] - y['f']) <= (1e-8 if dtype == np.float64 else 1e-5) * abs(y['f']) for x, y in zip(sa, sn))


def test_phase_events_on_every_nth_iteration_only(monkeypatch):
    """trmf_session_set_timing: the seven HIP event records behind TrmfIterStats.ms_* are instrumentation (barrier packets between
    kernels, ~25 us per iteration): with period N only the iterations whose 1-based index is a multiple of N carry them, the others
    report ms_* = -1; with 0 none does.  Everything else of the records and the factors is unaffected."""
    from trmf import session, synth
    p = synth.sparse_problem(n=900, T=400, k=12, nlag=4, density=0.06, dtype=np.float32, seed=31)
    m0 = synth.initial_model(p['Y'], p['lag_set'], 12, seed=31)
    out = {}
    for timing in (1, 3, 0):
        model = make_model(m0.W.astype(np.float32), m0.H.astype(np.float32), np.asfortranarray(m0.lag_val.astype(np.float32)), p['lag_set'])
        with session.Session(p['Y'].astype(np.float32), model, missing=True, log_norms=False, timing=timing, **synth.HYPER) as s:
            s.run(4); s.run(5); st = s.stats(9); s.download()
        out[timing] = (model, st)
    for timing in (3, 0):
        assert np.array_equal(out[timing][0].W, out[1][0].W) and np.array_equal(out[timing][0].H, out[1][0].H)
        assert [x['cg_iter'] for x in out[timing][1]] == [x['cg_iter'] for x in out[1][1]]
        assert [x['f'] for x in out[timing][1]] == [x['f'] for x in out[1][1]]
    assert all(x['ms_F'] > 0 and x['ms_X'] > 0 and x['ms_F_kernel'] > 0 for x in out[1][1])
    for i, x in enumerate(out[3][1]):                     # iterations 1..9
        carried = (i + 1) % 3 == 0
        assert (x['ms_F'] > 0 and x['ms_X'] > 0 and x['ms_X_gram'] > 0) if carried else (x['ms_F'] == -1 and x['ms_X'] == -1 and x['ms_LV'] == -1), (i, x)
    assert all(x['ms_F'] == -1 and x['ms_F_kernel'] == -1 for x in out[0][1])


@pytest.mark.parametrize('T,expect', [(6400, '256 threads'), (6425, '512 threads'), (13056, '512 threads'), (13100, '256 threads')])
def test_tile_geometry_rule_at_its_boundaries(T, expect, monkeypatch):
    """The rule itself (no TRMF_TILE): at rank 40 a narrow tile holds 25 timestamps and a wide one at most 51.  6400 timestamps are
    exactly 256 narrow tiles -> narrow; one tile more -> wide (ceil(T / CUs) timestamps per tile); 13056 = 256 x 51 is the last size
    the wide tiles cover with one workgroup per CU; beyond it narrow again (more tiles than the persistent kernel's table: the
    launch-per-step path).  Whatever it picks, both CG forms agree bit for bit and the factors stay within the parity gate of the
    other geometry."""
    from trmf import session, synth
    lib = session.lib_for(np.float32)
    p = synth.sparse_problem(n=1500, T=T, k=40, nlag=16, density=0.01, dtype=np.float64, seed=51)
    m0 = synth.initial_model(p['Y'], p['lag_set'], 40, seed=51)
    a, sa, da = _run(p, m0, np.float32, 2, True, monkeypatch, persist=True)
    b, sb, db = _run(p, m0, np.float32, 2, True, monkeypatch, persist=False)
    c, sc, dc = _run(p, m0, np.float32, 2, True, monkeypatch, persist=True, tile='narrow')
    assert expect in da, da
    if T == 13100:
        assert 'one launch per CG step' in da, da            # 524 narrow tiles: beyond the persistent kernel's 512
    else:
        assert 'persistent' in da, da
    assert np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H) and np.array_equal(a.lag_val, b.lag_val)
    assert [x['cg_iter'] for x in sa] == [x['cg_iter'] for x in sb]
    for u, v in ((a.W, c.W), (a.H, c.H), (a.lag_val, c.lag_val)):
        assert np.linalg.norm(u.astype(np.float64) - v.astype(np.float64)) <= 2e-3 * np.linalg.norm(v.astype(np.float64))
