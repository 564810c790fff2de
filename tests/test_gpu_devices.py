"""GPU (-m gpu): several GPUs behind the UNCHANGED reference entry -- TRMF_DEVICES / TRMF_GPUS (csrc/session_group.hpp).

The reference's caller is ONE process calling trmf.train -> c_trmf_train (trmf.py:253-264, trmf.cpp:696-725); no launcher, no
torch.  With TRMF_DEVICES=0,0 / 0,0,0,0 the ranks are threads of this process (virtual ranks on the box's one device -- the same
code path as one device per rank, minus the peer copies), joined by the in-process communicator.  The sessions are the ones the SPMD
launch builds, so the yardstick is the same as tests/test_dist.py's: bit-identical to ONE rank in the tile geometry several ranks use
(TRMF_TEST=1 TRMF_TILE=narrow)."""
import os

import numpy as np
import pytest

import trmf
from dist_worker import _problem
from helpers import capture_fds, make_model
from trmf import session, synth

pytestmark = pytest.mark.gpu


def _train(p, m0, dtype, iters, missing=True, hyper=synth.HYPER, periods=(1, 1, 2)):
    model = make_model(m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
    Y = p['Y'].astype(dtype)
    trmf.train(Y, model, max_iter=iters, missing=missing, period_W=periods[0], period_H=periods[1], period_Lag=periods[2], **hyper)   # c_trmf_train
    return model


def _devs(spec):
    """'0,0,0' as written on the one-GPU test box; a device per rank where the box has them and scripts/scale_day.sh asks for it."""
    n = len(spec.split(','))
    if os.environ.get('TRMF_TEST_DEVICE_PER_RANK') and session.lib_for(np.float32).trmf_device_count() >= n:
        return ','.join(str(i) for i in range(n))
    return spec


def _same(a, b):
    return np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H) and np.array_equal(a.lag_val, b.lag_val)


@pytest.mark.parametrize('devices', ['0,0', '0,0,0,0', '0,0,0'])
@pytest.mark.parametrize('shape', ['c4', 'odd'])
def test_c_trmf_train_under_TRMF_DEVICES_is_bit_identical_to_one_rank(devices, shape, monkeypatch):
    """trmf.train (no torch, no launcher) with the ranks as threads: F rows / X-side Gram rows sharded or replicated and the form of
    the CG chosen by the measure-once rules exactly as under the SPMD launch; same bits as one rank, fp32 and fp64."""
    p, m0 = _problem(shape)
    iters = 4
    for dtype in (np.float32, np.float64):
        monkeypatch.delenv('TRMF_DEVICES', raising=False)
        monkeypatch.setenv('TRMF_TILE', 'narrow')
        one = _train(p, m0, dtype, iters)
        monkeypatch.delenv('TRMF_TILE', raising=False)
        monkeypatch.setenv('TRMF_DEVICES', _devs(devices))
        many = _train(p, m0, dtype, iters)
        assert _same(one, many), (np.dtype(dtype).name, devices)


@pytest.mark.parametrize('env', [{'TRMF_CG': 'timeshard'}, {'TRMF_CG': 'p2p'}, {'TRMF_CG': 'replicate'},
                                 {'TRMF_CG': 'persist', 'TRMF_PERSIST_TIMEOUT_MS': '120000'},
                                 {'TRMF_NO_HV_TILE': '1'}, {'TRMF_NO_HV_TILE': '1', 'TRMF_CG': 'p2p'},
                                 {'TRMF_FOVERLAP': '2', 'TRMF_FSHARD': 'shard', 'TRMF_CG': 'timeshard'},
                                 {'TRMF_LONG_ROW': '40', 'TRMF_LONG_CHUNK': '32'}])
def test_every_form_of_the_sharded_solver_runs_between_threads(env, monkeypatch):
    """The forms of DESIGN.md section 6, forced: time-sharded CG through the communicator, peer to peer (arenas exchanged as raw
    pointers between threads instead of IPC handles), replicated, one persistent kernel per rank, the unfused two-kernel step and its
    peer-to-peer transport, the overlapped chunked gather of H, and the split path of long rows (positions of the long-row list per
    rank) -- two ranks as threads, bit-identical to one rank under the same switches."""
    p, m0 = _problem('c4')
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dtype = np.float32
    monkeypatch.setenv('TRMF_TILE', 'narrow')
    one = _train(p, m0, dtype, 3)
    monkeypatch.delenv('TRMF_TILE', raising=False)
    monkeypatch.setenv('TRMF_DEVICES', _devs('0,0'))
    two = _train(p, m0, dtype, 3)
    assert _same(one, two), env


def test_full_observation_path_under_TRMF_DEVICES(monkeypatch):
    """missing = 0 (dense Y; BASELINE config 1's path): Y^T W rows sharded + gathered, the shared-Gram CG replicated."""
    pd = synth.dense_problem(90, 700, 6, [1, 2, 24], dtype=np.float64, seed=13)
    m0 = synth.initial_model(pd['Y'], pd['lag_set'], 6, seed=7)
    one = _train(pd, m0, np.float64, 3, missing=False)
    monkeypatch.setenv('TRMF_DEVICES', _devs('0,0'))
    two = _train(pd, m0, np.float64, 3, missing=False)
    assert _same(one, two)


def test_resident_session_api_under_TRMF_GPUS_style_lists(monkeypatch):
    """The session API follows: create / run / stats / objective / describe / mark / rewind / append_rows / download run as one task on
    every rank's thread; getters answer from rank 0."""
    p, m0 = _problem('c4')
    dtype = np.float32
    Y = p['Y'].astype(dtype).tocsr()
    head, tail = Y[:1100], Y[1100:]

    def run():
        model = make_model(m0.W[:1100].astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
        with session.Session(head, model, missing=True, **synth.HYPER) as s:
            d = s.describe()
            s.run(2); s.mark(); s.run(2); st_a = s.stats(2); Ja = s.objective()
            s.rewind(); s.run(2); st_b = s.stats(2); Jb = s.objective()
            assert [x['cg_iter'] for x in st_a] == [x['cg_iter'] for x in st_b] and Ja == Jb      # the same two iterations again
            s.append_rows(tail)
            assert s.rows() == 1200
            grown = make_model(np.zeros((1200, m0.W.shape[1]), dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
            s.model = grown
            s.run(2); J = s.objective(); s.download()
        return grown, J, d

    monkeypatch.setenv('TRMF_TILE', 'narrow')
    one, J1, d1 = run()
    monkeypatch.delenv('TRMF_TILE', raising=False)
    monkeypatch.setenv('TRMF_DEVICES', _devs('0, 0'))
    two, J2, d2 = run()
    assert '2 ranks' in d2 and 'threads of this process' in d2 and '1 rank' in d1, (d1, d2)
    assert _same(one, two) and J1 == J2


def test_second_call_reuses_the_threads_communicators_and_decisions(monkeypatch):
    """The worker threads and their communicators are kept by the process between sessions (session_group.hpp: GroupRuntime cache), so the
    second call of a grid_search finds the first call's measure-once decisions in the process-level cache instead of measuring again."""
    p, m0 = _problem('c4')
    monkeypatch.setenv('TRMF_DEVICES', _devs('0,0'))
    monkeypatch.setenv('TRMF_PERSIST_TIMEOUT_MS', '2001')      # part of the decision cache's key (every TRMF_* variable is): a key no earlier test of this process has used
    descs = []
    for _ in range(2):
        model = make_model(m0.W.astype(np.float32), m0.H.astype(np.float32), np.asfortranarray(m0.lag_val.astype(np.float32)), p['lag_set'])
        with session.Session(p['Y'].astype(np.float32), model, missing=True, **synth.HYPER) as s:
            s.run(2); descs.append(s.describe())
    assert 'decided in' in descs[0] and '(cached)' in descs[1], descs
    # trmf_release_cached() also retires the idle worker threads and their communicators: the next session measures again
    assert session.lib_for(np.float32).trmf_release_cached() == 0
    model = make_model(m0.W.astype(np.float32), m0.H.astype(np.float32), np.asfortranarray(m0.lag_val.astype(np.float32)), p['lag_set'])
    with session.Session(p['Y'].astype(np.float32), model, missing=True, **synth.HYPER) as s:
        s.run(1); d3 = s.describe()
    assert '(cached)' not in d3, d3


def test_bad_device_lists_fail_loudly_and_leave_the_outputs_alone(monkeypatch):
    p, m0 = _problem('small')
    for bad in ('0,7', '0,x'):
        monkeypatch.setenv('TRMF_DEVICES', bad)
        model = make_model(m0.W.astype(np.float32), m0.H.astype(np.float32), np.asfortranarray(m0.lag_val.astype(np.float32)), p['lag_set'])
        W0, H0 = model.W.copy(), model.H.copy()
        with capture_fds() as cap:
            trmf.train(p['Y'].astype(np.float32), model, max_iter=2, missing=True, **synth.HYPER)
        assert any('TRMF_DEVICES' in l for l in cap.err), cap.err
        assert np.array_equal(model.W, W0) and np.array_equal(model.H, H0)
    monkeypatch.setenv('TRMF_DEVICES', '0')            # one device listed: the plain one-rank call
    model = make_model(m0.W.astype(np.float32), m0.H.astype(np.float32), np.asfortranarray(m0.lag_val.astype(np.float32)), p['lag_set'])
    W0 = model.W.copy()
    trmf.train(p['Y'].astype(np.float32), model, max_iter=1, missing=True, **synth.HYPER)
    assert not np.array_equal(model.W, W0)


def test_a_failing_rank_fails_the_call_as_a_whole(monkeypatch):
    """All-or-nothing across ranks (trmf.cpp:632-634): a failed download on rank 0 / a broken set-up stage on one rank leaves W, H,
    lag_val as passed, and no rank is left waiting for another."""
    p, m0 = _problem('small')
    monkeypatch.setenv('TRMF_DEVICES', _devs('0,0'))
    monkeypatch.setenv('TRMF_FAIL_DOWNLOAD', '1')
    model = make_model(m0.W.astype(np.float32), m0.H.astype(np.float32), np.asfortranarray(m0.lag_val.astype(np.float32)), p['lag_set'])
    W0, H0, T0 = model.W.copy(), model.H.copy(), model.lag_val.copy()
    with capture_fds() as cap:
        trmf.train(p['Y'].astype(np.float32), model, max_iter=2, missing=True, **synth.HYPER)
    assert any('outputs untouched' in l for l in cap.err), cap.err
    assert np.array_equal(model.W, W0) and np.array_equal(model.H, H0) and np.array_equal(model.lag_val, T0)
    monkeypatch.delenv('TRMF_FAIL_DOWNLOAD')
    trmf.train(p['Y'].astype(np.float32), model, max_iter=2, missing=True, **synth.HYPER)      # and the library is fine afterwards
    assert not np.array_equal(model.W, W0)


def test_verbose_lines_appear_once(monkeypatch):
    """The reference's `>> iter` lines (trmf.cpp:661,672,687) come from rank 0 only."""
    p, m0 = _problem('small')
    monkeypatch.setenv('TRMF_DEVICES', _devs('0,0'))
    model = make_model(m0.W.astype(np.float32), m0.H.astype(np.float32), np.asfortranarray(m0.lag_val.astype(np.float32)), p['lag_set'])
    with capture_fds() as cap:
        trmf.train(p['Y'].astype(np.float32), model, max_iter=2, missing=True, verbose=1, **synth.HYPER)
    assert sum(1 for l in cap.err if l.startswith('>> iter 1 F')) == 1 and sum(1 for l in cap.err if l.startswith('>> iter 2 X')) == 1, cap.err
