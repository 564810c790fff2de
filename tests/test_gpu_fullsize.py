"""GPU (-m gpu): BASELINE.json's configurations at their FULL sizes against the oracle.

* config 3 (100k x 10k, 1 %, k=40, |L|=16, fp32): 2 ALS iterations vs the C restatement on all host
  cores -- the headline size itself, not only a scaled-down shape.
* config 5 (1M x 50k, 0.1 %, k=64, |L|=32, fp64) on ONE GPU: 1 ALS iteration vs the restatement at the
  fp64 gates, plus size-independent properties.
* the fused CG (one launch per iteration, multi-wave grids) against the unfused kernels.
* the paper scripts' own shape (dense, missing=0, k=60 / 40, 48 lags up to 191).
Tolerances: helpers.TOL."""
import os

import numpy as np
import pytest

import oracle_py as O
import trmf
from helpers import evidence, TOL, make_model, relfro, relmax
from trmf import session, synth

pytestmark = pytest.mark.gpu
NCPU = os.cpu_count() or 8


def test_config3_full_size_vs_oracle(monkeypatch):
    monkeypatch.setenv('TRMF_CG_DIRECT', '1')      # diagnostics: the closing H s pass (|-g - H s| evaluated directly) -- off by default since round 6
    cfg = synth.CONFIGS['c3']
    p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=np.float32, seed=0)
    m0 = synth.initial_model(p['Y'], p['lag_set'], cfg['k'], seed=0)
    iters = 2
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    log = O.train_port(p['Y'], p['lag_set'], W, H, Th, synth.HYPER, max_iter=iters, threads=NCPU)
    model = make_model(m0.W, m0.H, m0.lag_val, p['lag_set'])
    with session.Session(p['Y'], model, missing=True, **synth.HYPER) as s:
        s.run(iters); st = s.stats(iters); s.download()
    Jo = O.objective(p['Y'], p['lag_set'], W, H, Th, synth.HYPER)
    Jp = O.objective(p['Y'], p['lag_set'], model.W, model.H, model.lag_val, synth.HYPER)
    cg_o, cg_p = [l['cg_iter'] for l in log], [x['cg_iter'] for x in st]
    evidence('c3 full size: J oracle %.10g gpu %.10g rel %.2e; relfro W %.2e H %.2e Th %.2e; CG oracle %s gpu %s' % (
        Jo, Jp, abs(Jp - Jo) / Jo, relfro(model.W, W), relfro(model.H, H), relfro(model.lag_val, Th), cg_o, cg_p))
    assert abs(Jp - Jo) / Jo < 1e-5
    assert relfro(model.H, H) < 1e-3 and relfro(model.W, W) < 1e-3
    assert all(abs(a - b) <= 1 for a, b in zip(cg_o, cg_p))
    # the CG's r^T r is the recurrence rho - 2 alpha <r,Hd> + alpha^2 <Hd,Hd> (the reference recomputes r^T r, rf_tron.h:492); the
    # stop test reads it.  Its drift against the DIRECT norm |-g - H s|^2 of the same step, at the stopping iteration:
    drift = [abs(x['cg_rnorm'] ** 2 - x['cg_rnorm_direct'] ** 2) / x['cg_rnorm_direct'] ** 2 for x in st if x['cg_rnorm_direct'] > 0]
    evidence('c3 full size: CG recurrence vs direct residual norm at the stopping step, |rho_rec - rho_direct| / rho_direct per iteration: %s; '
             'cg_rnorm / cgtol-side |g|: %s' % (['%.1e' % d for d in drift], ['%.4f' % (x['cg_rnorm'] / x['gnorm']) for x in st]))
    assert drift and max(drift) < 1e-2      # far inside the margin of the eps_cg = 0.1 exit


def test_config3_full_size_vs_reference_build():
    """The headline size against the REAL reference (oracle/_ref: trmf.cpp compiled unmodified, OpenBLAS), not only its
    restatement: from the warm state bench.py's timed window starts from (five GPU iterations from the random start),
    two full ALS iterations on both sides -- the horizon over which an fp32 run measures the implementation rather than
    the noise floor of the truncated CG (profiles/r02_fp32_trajectory_c3.txt) -- at the north_star's gates: objective
    1e-5 relative, factors 1e-3 relative Frobenius.  One F-only and one X-only step from the same state attribute the
    difference to a phase (printed)."""
    if O.ref(np.float32) is None:
        pytest.skip('oracle/_ref not built (make -C oracle ref, build container only; the .so files travel with gpurun)')
    cfg = synth.CONFIGS['c3']
    p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=np.float32, seed=0)
    Y, lags = p['Y'], p['lag_set']
    m0 = synth.initial_model(Y, lags, cfg['k'], seed=0)
    warm = make_model(m0.W, m0.H, m0.lag_val, lags)
    with session.Session(Y, warm, missing=True, **synth.HYPER) as s:
        s.run(5); s.download()
    W0, H0, T0 = warm.W.copy(), warm.H.copy(), np.asfortranarray(warm.lag_val.copy())
    big = 10 ** 6
    threads = min(64, NCPU)          # the bundled OpenBLAS supports at most 64 calling threads (posv inside the OpenMP loop)

    def both(iters, periods):
        W, H, Th = W0.copy(), H0.copy(), np.asfortranarray(T0.copy())
        O.train_ref(Y, lags, W, H, Th, synth.HYPER, max_iter=iters, periods=periods, threads=threads)
        model = make_model(W0, H0, T0, lags)
        with session.Session(Y, model, missing=True, period_W=periods[0], period_H=periods[1], period_Lag=periods[2], **synth.HYPER) as s:
            s.run(iters); st = s.stats(iters); s.download()
        return (W, H, Th), model, st

    (Wf, Hf, _), mf, _ = both(1, (big, 1, big))                # F-solve only
    (Wx, Hx, _), mx, _ = both(1, (1, big, big))                # X-solve only
    evidence('config 3 vs the reference build, one phase from the warm state: F-solve relfro(H) %.2e; X-solve relfro(W) %.2e' % (
        relfro(mf.H, Hf), relfro(mx.W, Wx)))
    (W, H, Th), model, st = both(2, (1, 1, 2))
    Wp, Hp, Tp = W0.copy(), H0.copy(), np.asfortranarray(T0.copy())
    log = O.train_port(Y, lags, Wp, Hp, Tp, synth.HYPER, max_iter=2, threads=NCPU)
    Jr = O.objective(Y, lags, W, H, Th, synth.HYPER)
    Jg = O.objective(Y, lags, model.W, model.H, model.lag_val, synth.HYPER)
    Jp = O.objective(Y, lags, Wp, Hp, Tp, synth.HYPER)
    cg_p, cg_g = [l['cg_iter'] for l in log], [x['cg_iter'] for x in st]
    evidence('config 3 vs the reference build, 2 iterations from the warm state: J ref %.10g gpu %.10g rel %.2e (restatement vs ref %.2e); '
          'relfro W %.2e H %.2e Theta %.2e; CG restatement %s gpu %s' % (Jr, Jg, abs(Jg - Jr) / Jr, abs(Jp - Jr) / Jr, relfro(model.W, W),
                                                                        relfro(model.H, H), relfro(model.lag_val, Th), cg_p, cg_g))
    assert abs(Jg - Jr) / Jr < 1e-5
    assert relfro(model.W, W) < 1e-3 and relfro(model.H, H) < 1e-3
    assert relfro(mf.H, Hf) < 1e-3 and relfro(mx.W, Wx) < 1e-3
    assert all(abs(a - b) <= 1 for a, b in zip(cg_p, cg_g))       # the restatement's CG counts are pinned to the reference's (goldens)


def test_config3_fp64_10_iterations_vs_reference_build():
    """north_star's horizon at the headline size, where it can be shown: config 3's workload in fp64, TEN ALS iterations from the
    random start on the GPU's fp64 library against the real reference's fp64 build.  In fp32 the truncated CG amplifies rounding
    differences from iteration to iteration (the reference's own fp32 build drifts 1.5e-4 from its line-by-line restatement by
    iteration 10, profiles/r02_fp32_trajectory_c3.txt), so the fp32 tests compare over two iterations; fp64 has no such noise floor
    and both builds exist: factors 1e-6 (max-rel), CG counts EQUAL to the ones the reference prints on its TRON line
    (rf_tron.h:219), objective 1e-7 -- SURVEY 8(d)'s 1e-8 is the gate after ONE iteration (tested on every golden and at config 5's
    full size); over ten iterations of a CG truncated at 10 % the last bits of every Gram (summation order: MFMA chain here,
    scalar loop + LAPACK posv there) and of r^T r (a recurrence here, recomputed there: rf_tron.h:492) are amplified.  Measured:
    1.7e-8.  The yardstick is the reference's own line-by-line restatement run beside it over the same horizon: it lands 2.6e-9 from
    the reference on 256 OpenMP threads and 1.1e-7 on 64 -- the order of its dot-product reductions alone moves the ten-iteration
    objective by that much (printed, not gated); the two fp32 builds are 1e-4 apart there.  Gate: 1e-7.  Gated as well (round 6): the fp32 GPU run is no farther from that common fp64
    trajectory than twice the farthest of the reference side's own fp32 runs (its fp32 build on 8 and 64 threads, its fp32 restatement)."""
    import re
    from helpers import capture_fds
    if O.ref(np.float64) is None:
        pytest.skip('oracle/_ref not built (make -C oracle ref, build container only; the .so files travel with gpurun)')
    cfg = synth.CONFIGS['c3']
    p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=np.float64, seed=0)
    Y, lags = p['Y'], p['lag_set']
    m0 = synth.initial_model(Y, lags, cfg['k'], seed=0, dtype=np.float64)
    iters, threads = 10, 8                     # 8 threads: the reference's fastest setting on this host (posv inside OpenMP, bench.py)
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    with capture_fds() as cap:
        O.train_ref(Y, lags, W, H, Th, synth.HYPER, max_iter=iters, threads=threads, verbose=2)
    cg_ref = [int(m.group(1)) for l in cap.out for m in [re.search(r'CG\s+(\d+)', l)] if m]
    model = make_model(m0.W, m0.H, m0.lag_val, lags)
    with session.Session(Y, model, missing=True, **synth.HYPER) as s:
        s.run(iters); st = s.stats(iters); s.download()
    Jr = O.objective(Y, lags, W, H, Th, synth.HYPER)
    Jg = O.objective(Y, lags, model.W, model.H, model.lag_val, synth.HYPER)
    cg_gpu = [x['cg_iter'] for x in st]
    evidence('config 3 in fp64, 10 iterations vs the reference build: J ref %.14g gpu %.14g rel %.2e; relmax W %.2e H %.2e Theta %.2e; '
             'CG reference %s gpu %s' % (Jr, Jg, abs(Jg - Jr) / Jr, relmax(model.W, W), relmax(model.H, H), relmax(model.lag_val, Th), cg_ref, cg_gpu))
    Wp, Hp, Tp = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    O.train_port(Y, lags, Wp, Hp, Tp, synth.HYPER, max_iter=iters, threads=min(64, NCPU))
    Jp = O.objective(Y, lags, Wp, Hp, Tp, synth.HYPER)
    evidence('config 3 in fp64, 10 iterations: the restatement against the reference build over the same horizon: J rel %.2e; relmax W %.2e H %.2e' % (
        abs(Jp - Jr) / Jr, relmax(Wp, W), relmax(Hp, H)))
    assert abs(Jg - Jr) / Jr < 1e-7
    assert relmax(model.W, W) < 1e-6 and relmax(model.H, H) < 1e-6 and relmax(model.lag_val, Th) < 1e-6
    assert len(cg_ref) == iters and cg_gpu == cg_ref
    # north_star's fp32 gate -- objective 1e-5 after TEN iterations -- at this size, as a tested statement about the reference itself
    # (VERDICT r5 item 3): the fp32 GPU run against this fp64 trajectory next to the reference's own fp32 build (8 and 64 OpenMP
    # threads: its BLAS dot products are order dependent, so the two differ) and its fp32 restatement.  The GPU must be no farther
    # from the fp64 trajectory than TWICE the farthest reference-side fp32 run, in objective, W and H; and where the reference's two
    # thread counts agree with each other to 1e-5 the GPU must meet 1e-5 against them as well.
    Y32 = Y.astype(np.float32)
    g32 = make_model(m0.W.astype(np.float32), m0.H.astype(np.float32), np.asfortranarray(m0.lag_val.astype(np.float32)), lags)
    with session.Session(Y32, g32, missing=True, **synth.HYPER) as s:
        s.run(iters); s.download()
    Jg32 = O.objective(Y, lags, g32.W, g32.H, g32.lag_val, synth.HYPER)
    got = dict(J=abs(Jg32 - Jr) / Jr, W=relfro(g32.W, W), H=relfro(g32.H, H))
    side = []
    runs = [('restatement fp32, %d threads' % min(64, NCPU), lambda a, b, c: O.train_port(Y32, lags, a, b, c, synth.HYPER, max_iter=iters, threads=min(64, NCPU)))]
    if O.ref(np.float32) is not None:
        for t in sorted({8, min(64, NCPU)}):
            runs.append(('reference fp32 build, %d threads' % t, lambda a, b, c, t=t: O.train_ref(Y32, lags, a, b, c, synth.HYPER, max_iter=iters, threads=t)))
    for name, fn in runs:
        W32, H32, T32 = m0.W.astype(np.float32), m0.H.astype(np.float32), np.asfortranarray(m0.lag_val.astype(np.float32))
        fn(W32, H32, T32)
        J32 = O.objective(Y, lags, W32, H32, T32, synth.HYPER)
        side.append(dict(name=name, J=abs(J32 - Jr) / Jr, W=relfro(W32, W), H=relfro(H32, H), Jabs=J32))
    evidence('config 3 after 10 iterations, distance to the fp64 trajectory (objective rel / relfro W / relfro H): fp32 GPU %.2e / %.2e / %.2e; %s' % (
        got['J'], got['W'], got['H'], '; '.join('%s %.2e / %.2e / %.2e (GPU vs it, objective: %.2e)' % (r['name'], r['J'], r['W'], r['H'], abs(Jg32 - r['Jabs']) / Jr) for r in side)))
    for key in ('J', 'W', 'H'):
        assert got[key] <= 2.0 * max(r[key] for r in side), (key, got[key], [r[key] for r in side])
    # ... and against the reference's fp32 BUILD itself: that build is reproducible (8 and 64 OpenMP threads agree to 1e-14), but its own
    # line-by-line restatement -- the same algorithm, the dot products summed in plain loop order instead of OpenBLAS's kernel order --
    # lands 1.5e-4 from it over these ten iterations: the sensitivity of the truncated fp32 CG to summation order at this size.  The GPU
    # (a third summation order) must be no farther from the reference build than twice that.  north_star's literal 1e-5 is met over ten
    # iterations at config 2's size (tests/test_gpu_parity.py::test_objective_parity_fp32_10_iterations_config2_shape) and over two
    # iterations here; at ten iterations of config 3 no two fp32 implementations of the reference's own algorithm agree to 1e-5.
    refs = [r for r in side if r['name'].startswith('reference')]
    rest = [r for r in side if r['name'].startswith('restatement')][0]
    if refs:
        yard = abs(rest['Jabs'] - refs[0]['Jabs']) / Jr
        gpu_vs_ref = abs(Jg32 - refs[0]['Jabs']) / Jr
        evidence('config 3 after 10 iterations, fp32 objective: GPU vs the reference build %.2e; the reference\'s restatement vs the reference build %.2e; '
                 'the reference build against itself on %s threads %.2e' % (gpu_vs_ref, yard, ' / '.join(r['name'].split(', ')[1].split()[0] for r in refs),
                                                                           abs(refs[0]['Jabs'] - refs[-1]['Jabs']) / Jr))
        assert gpu_vs_ref <= 2.0 * yard + 1e-6


def test_config5_full_size_single_gpu_vs_oracle():
    """1M x 50k, 0.1 %, k=64, |L|=32, fp64 on one GPU (HBM capacity + the fp64 rank-64 kernels + T = 50k in
    the CG): one ALS iteration vs the restatement on all host cores, fp64 gates."""
    try:
        avail_gb = os.sysconf('SC_AVPHYS_PAGES') * os.sysconf('SC_PAGE_SIZE') / 2 ** 30
    except (ValueError, OSError):
        avail_gb = 1e9
    if avail_gb < 48:
        pytest.skip('needs ~40 GB of host memory to generate config 5')
    cfg = synth.CONFIGS['c5']
    p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=np.float64, seed=0)
    Y = p['Y']
    m0 = synth.initial_model(Y, p['lag_set'], cfg['k'], seed=0)
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    log = O.train_port(Y, p['lag_set'], W, H, Th, synth.HYPER, max_iter=1, threads=NCPU)
    model = make_model(m0.W, m0.H, m0.lag_val, p['lag_set'])
    with session.Session(Y, model, missing=True, **synth.HYPER) as s:
        s64 = 8
        assert abs(s.fsolve_bytes() - (Y.nnz * (4 + s64 + 64 * s64) + (cfg['n'] + 1) * 8 + cfg['n'] * 64 * s64)) < 1
        s.run(1); st = s.stats(1); Jdev = s.objective(); s.download()
        s.run(2); st2 = s.stats(2); J3 = s.objective()
    Jo = O.objective(Y, p['lag_set'], W, H, Th, synth.HYPER)
    Jp = O.objective(Y, p['lag_set'], model.W, model.H, model.lag_val, synth.HYPER)
    evidence('c5 full size: J oracle %.12g gpu %.12g (device %.12g) rel %.2e; relmax H %.2e W %.2e; CG oracle %s gpu %s; ms F %.2f X %.2f' % (
        Jo, Jp, Jdev, abs(Jp - Jo) / Jo, relmax(model.H, H), relmax(model.W, W), [l['cg_iter'] for l in log],
        [x['cg_iter'] for x in st], st[0]['ms_F'], st[0]['ms_X']))
    assert relmax(model.H, H) < 1e-6 and relmax(model.W, W) < 1e-6
    assert abs(Jp - Jo) / Jo < 1e-8 and abs(Jdev - Jp) / Jp < 1e-10
    assert [x['cg_iter'] for x in st] == [l['cg_iter'] for l in log]
    assert all(x['accepted'] == 1 for x in st + st2) and J3 < Jdev       # later iterations keep descending


@pytest.mark.parametrize('dtype,k,T,nlag', [(np.float64, 40, 27000, 4), (np.float32, 40, 60000, 16), (np.float64, 24, 20000, 6)])
def test_fused_cg_equals_unfused_on_multiwave_grids(dtype, k, T, nlag, monkeypatch):
    """More CG tiles than the chip holds at once (one workgroup per tile: 2250 / 2400 / 1539 tiles vs <= 512
    resident), solves stopping early on eps_cg, so such a solve has ONE stopping launch followed by launches
    that must do nothing: the fused path (hv_tile_kernel) must reproduce the unfused kernels (TRMF_NO_HV_TILE)
    -- same arithmetic, same stop iteration."""
    p = synth.sparse_problem(n=50, T=T, k=k, nlag=nlag, density=0.04, dtype=dtype, seed=13)
    m0 = synth.initial_model(p['Y'], p['lag_set'], k, seed=13)
    iters = 4

    def run():
        model = make_model(m0.W, m0.H, m0.lag_val, p['lag_set'])
        with session.Session(p['Y'], model, missing=True, **synth.HYPER) as s:
            s.run(iters); st = s.stats(iters); s.download()
        return model, [x['cg_iter'] for x in st]

    fused, cg_f = run()
    monkeypatch.setenv('TRMF_NO_HV_TILE', '1')
    plain, cg_u = run()
    assert min(cg_f) >= 1 and min(cg_f) < 20, cg_f              # at least one solve stopped by eps_cg, not by the iteration cap
    tol = TOL[np.dtype(dtype).name]
    print('fused vs unfused: CG %s / %s, relmax W %.2e H %.2e' % (cg_f, cg_u, relmax(fused.W, plain.W), relmax(fused.H, plain.H)))
    if dtype == np.float64:
        assert cg_f == cg_u
        assert relmax(fused.W, plain.W) < tol['factor'] and relmax(fused.H, plain.H) < tol['factor']
    else:
        assert all(abs(a - b) <= 1 for a, b in zip(cg_f, cg_u))
        assert relfro(fused.W, plain.W) < tol['factor'] and relfro(fused.H, plain.H) < tol['factor']


@pytest.mark.parametrize('dtype,k,nlag', [(np.float64, 64, 8), (np.float32, 40, 16), (np.float64, 21, 5), (np.float32, 7, 3), (np.float64, 2, 2)])
def test_packed_grams_equal_full_grams_on_the_unfused_path(dtype, k, nlag, monkeypatch):
    """Unfused X-solve: the cached Grams are kept as packed upper triangles and apply_kernel<true, STAGES> stages them
    through LDS (all three STAGES instantiations are hit: k = 64 -> 17, 40 -> 10, 21 / 7 / 2 -> 5, with row groups that
    straddle wavefronts and a short last group).  Same values multiplied in the same order as with full k x k Grams
    (TRMF_GRAM_FULL): the factors must be bit-identical."""
    p = synth.sparse_problem(n=60, T=1237, k=k, nlag=nlag, density=0.05, dtype=dtype, seed=29)
    m0 = synth.initial_model(p['Y'], p['lag_set'], k, seed=29)
    monkeypatch.setenv('TRMF_NO_HV_TILE', '1')

    def run():
        model = make_model(m0.W, m0.H, m0.lag_val, p['lag_set'])
        with session.Session(p['Y'], model, missing=True, **synth.HYPER) as s:
            s.run(3); st = s.stats(3); s.download()
        return model, [x['cg_iter'] for x in st]

    packed, cg_p = run()
    monkeypatch.setenv('TRMF_GRAM_FULL', '1')
    full, cg_f = run()
    assert cg_p == cg_f
    assert np.array_equal(packed.W, full.W) and np.array_equal(packed.H, full.H) and np.array_equal(packed.lag_val, full.lag_val)


PAPER_LAGS = list(range(1, 25)) + list(range(7 * 24, 8 * 24))          # run_electricity.py:13, run_traffic.py:13


@pytest.mark.parametrize('T,n,k,hyper', [
    (6000, 370, 60, dict(lambdaI=0.5, lambdaAR=125.0, lambdaLag=2.0)),        # run_electricity.py:9-25
    (4000, 963, 40, dict(lambdaI=2.0, lambdaAR=625.0, lambdaLag=0.5)),        # run_traffic.py:9-25
])
def test_paper_script_shape_full_observation(T, n, k, hyper):
    """The shape both paper scripts train at: dense Y, missing=0, fp64, k=60 (electricity) / 40 (traffic), 48 lags
    {1..24} u {168..191} (max lag 191 -> unfused CG, 148 KB-class LDS requests of the full-path kernels), vs the
    restatement at the fp64 gates.  T is shortened (the real series have 26 304 / 10 560 timestamps)."""
    d = trmf.Model.syn_gen(T, n, k, PAPER_LAGS, seed=1, dtype=np.float64)
    Y = np.ascontiguousarray(d['Y'] + 0.05 * np.random.RandomState(1).randn(T, n))
    m0 = trmf.Model.initialize(Y, PAPER_LAGS, k, seed=0)
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    O.train_port(Y, m0.lag_set, W, H, Th, hyper, max_iter=3, missing=False, threads=NCPU)
    model = make_model(m0.W, m0.H, m0.lag_val, m0.lag_set)
    trmf.train(Y, model, max_iter=3, missing=False, **hyper)
    print('paper shape k=%d: relmax W %.2e H %.2e Th %.2e' % (k, relmax(model.W, W), relmax(model.H, H), relmax(model.lag_val, Th)))
    assert relmax(model.W, W) < 1e-6 and relmax(model.H, H) < 1e-6 and relmax(model.lag_val, Th) < 1e-5
    J = lambda A, B: 0.5 * np.sum((Y - A @ B.T) ** 2)
    assert abs(J(model.W, model.H) - J(W, H)) / J(W, H) < 1e-8


def _reference_engine(monkeypatch):
    """Route trmf.train to the CPU engine (the real reference build where oracle/_ref travelled, else the
    restatement) so that the SAME Python harness drives both engines."""
    import trmf.trmf as front
    cpu_train = O.train_ref if O.ref(np.float64) is not None else O.train_port

    def train(Y, model, lambdaI=0.1, lambdaAR=0.1, lambdaLag=0.1, max_iter=10, period_W=1, period_H=1, period_Lag=2,
              threads=1, missing=False, verbose=0):
        if model.transform is not None:
            Y = model.transform.preprocess(Y)
        cpu_train(np.ascontiguousarray(Y), model.lag_set, model.W, model.H, model.lag_val,
                  dict(lambdaI=lambdaI, lambdaAR=lambdaAR, lambdaLag=lambdaLag), max_iter=max_iter,
                  periods=(period_W, period_H, period_Lag), threads=threads, missing=missing, verbose=verbose)
        return model
    monkeypatch.setattr(front, 'train', train)
    return 'reference' if cpu_train is O.train_ref else 'restatement'


def test_paper_protocol_rolling_validation(monkeypatch):
    """run_electricity.py end to end at its real shape (26 304 x 370, k=60, 48 lags, missing=False, transform=True,
    rolling 24-step windows with warm starts): the resident-session harness on the GPU against the same harness
    driving the CPU engine.  Compared: every forecast metric.  TRMF_PAPER_FULL=1 runs all 7 windows x 40 iterations
    (the script's setting; minutes of CPU time) -- the default keeps 3 windows x 8 iterations."""
    import time
    full = os.environ.get('TRMF_PAPER_FULL') == '1'
    nr_windows, max_iter = (7, 40) if full else (3, 8)
    T, n, k = 26304, 370, 60
    d = trmf.Model.syn_gen(T, n, k, PAPER_LAGS, seed=2, dtype=np.float64)
    scale = np.random.RandomState(2).lognormal(3.0, 1.5, n)              # per-series level and spread, as in load data
    Y = np.ascontiguousarray((d['Y'] + 0.05 * np.random.RandomState(3).randn(T, n)) * scale + 2.0 * scale)
    kw = dict(k=k, window_size=24, nr_windows=nr_windows, lambdaI=0.5, lambdaAR=125, lambdaLag=2, max_iter=max_iter,
              threshold=None, transform=True, seed=0, missing=False)
    t0 = time.time(); gpu = trmf.rolling_validate(Y, PAPER_LAGS, **kw); t_gpu = time.time() - t0
    engine = _reference_engine(monkeypatch)
    threads = min(40, NCPU)                                               # run_electricity.py:18
    t0 = time.time(); cpu = trmf.rolling_validate(Y, PAPER_LAGS, threads=threads, resident=False, **kw); t_cpu = time.time() - t0
    print('paper protocol (%d windows x %d iterations): GPU %.1f s, CPU %s on %d threads %.1f s' %
          (nr_windows, max_iter, t_gpu, engine, threads, t_cpu))
    print('  GPU', gpu); print('  CPU', cpu)
    for name in ('nd', 'mase', 'nrmse', 'm_nd', 'm_mase', 'm_nrmse', 'mape'):
        a, b = getattr(gpu, name), getattr(cpu, name)
        assert abs(a - b) <= 1e-6 * abs(b), (name, a, b)
