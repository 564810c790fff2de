"""World_size-2 coverage of the sharded ALS loop.

* CPU / gloo (`-m "not gpu"`): the library's partitioner + the all-gather layout, with the oracle
  doing the arithmetic of each shard (the product has no CPU compute path).
* GPU (`-m gpu`): two processes share the one GPU of the box and exchange blocks through the
  host-staged communicator (gloo); the device kernels run on each rank's shard.  The RCCL
  communicator itself needs >= 2 GPUs and is exercised by bench.py --gpus N on the driver's node;
  with one rank it is covered below (all-gather degenerates to a no-op)."""
import socket

import os

import numpy as np
import pytest

from helpers import evidence, make_model, relfro


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    return port


def _spawn(fn, world, *args):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    out, want, waited = [], (1 if fn.__name__ == 'cpu_sharded_fsolve' else world), 0
    while len(out) < want:          # a rank that died will never report: fail at once instead of waiting out the timeout
        try:
            out.append(q.get(timeout=5))
        except Exception:
            waited += 5
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or waited > 1500:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError('rank process exit codes {} after {} s'.format([p.exitcode for p in procs], waited))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return out


def test_gloo_world2_sharded_fsolve_matches_unsharded():
    import dist_worker
    (name, ok, bounds), = _spawn(dist_worker.cpu_sharded_fsolve, 2)
    assert ok, bounds
    assert 0 < bounds[1] < bounds[2]


@pytest.mark.gpu
@pytest.mark.parametrize('shape', ['small', 'odd'])
def test_two_ranks_one_gpu_match_single_process(shape):
    """'odd': rank 5 -- Gram blocks of 100 (fp32) / 200 (fp64) bytes, so the gathered blocks start at offsets that are
    only 4- / 8-byte aligned (the narrow paths of the unpack kernel) and every block has its own size."""
    import dist_worker
    from trmf import session, synth
    iters = 3
    out = dict(_spawn(dist_worker.gpu_host_staged, 2, iters, shape))
    p, m0 = dist_worker._problem(shape)
    for dtype in (np.float32, np.float64):
        name = np.dtype(dtype).name
        W0, H0, T0 = m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype))
        model = make_model(W0, H0, T0, p['lag_set'])
        with session.Session(p['Y'].astype(dtype), model, missing=True, **synth.HYPER) as s:
            s.run(iters); st = s.stats(iters); s.download()
        for r in (0, 1):
            W, H, Th, cg, second_session_same = out[r][name][:5]
            # every kernel is deterministic and every rank derives the same scalars: bit-identical everywhere
            assert np.array_equal(W, model.W) and np.array_equal(H, model.H) and np.array_equal(Th, model.lag_val)
            assert cg == [x['cg_iter'] for x in st]
            assert second_session_same      # a second session under the same communicator (its staging pool outlives streams)


def _single_process(p, m0, dtype, iters):
    """The one-process run the N-rank runs are compared with bit for bit: in the tile geometry several ranks always use (one rank
    alone takes WIDE tiles where the narrow ones outnumber the CUs -- config 3 at full size; same iterates up to rounding, not
    bit for bit: tests/test_gpu_persist.py)."""
    from trmf import session, synth
    model = make_model(m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
    old = os.environ.get('TRMF_TILE')
    os.environ['TRMF_TILE'] = 'narrow'
    try:
        with session.Session(p['Y'].astype(dtype), model, missing=True, **synth.HYPER) as s:
            s.run(iters); st = s.stats(iters); s.download()
    finally:
        if old is None: os.environ.pop('TRMF_TILE', None)
        else: os.environ['TRMF_TILE'] = old
    return model, [x['cg_iter'] for x in st]


@pytest.mark.gpu
@pytest.mark.parametrize('world,shape,mode', [(2, 'small', 'timeshard'), (2, 'odd', 'timeshard'), (2, 'c4', 'timeshard'),
                                              (4, 'c4', 'timeshard'), (8, 'c4', 'timeshard'), (3, 'c4', 'measure'),
                                              (4, 'c4', 'replicate'), (3, 'c4', 'overlap'), (2, 'odd', 'overlap'), (2, 'small', 'p2p'), (2, 'c4', 'p2p'), (4, 'c4', 'p2p'),
                                              (8, 'c4', 'p2p'), (3, 'c4', 'split'), (2, 'small', 'persist'), (3, 'odd', 'persist'), (2, 'c4', 'persist'), (4, 'c4', 'persist')])
def test_time_sharded_cg_matches_single_process(world, shape, mode):
    """The CG sharded over TIME: every rank runs the tiles of its own block of timestamps, the tile
    records (three scalars per CG step) and midx halo rows per neighbour are exchanged after every launch.  Same
    records summed in the same order => bit-identical to the single-process run, in both precisions, for 2 / 3 / 4 / 8
    ranks (uneven last block at 'odd'), forced (TRMF_CG=timeshard), forced off, and under the measure-once rule (which
    runs iterations 1-2 replicated and 3-4 time-sharded, then decides: 5 iterations cover the switch both ways).
    'p2p': the peer-to-peer form of the exchange -- every rank's kernels write their tile records and edge rows straight
    into the other ranks' IPC-mapped message buffers and synchronise through flag words (bounded waits); here the "peers"
    are processes sharing the one GPU, the code path (IPC handles, remote stores, system-scope flags) is the multi-GPU one.
    'persist': ONE persistent kernel per rank and solve (csrc/cg_persist.hpp, SHARD): tagged records into every rank's arena, tagged
    edge rows into the neighbours', no launch and no sync kernel inside the solve.  (Up to 4 ranks here: persistent kernels of EIGHT
    processes on one device are time-sliced against each other -- bit-identical too when run, but ~30 s per solve; with more than 4
    ranks on one device the measure-once rule leaves the form out, and on a real node every rank has its own GPU.)
    'overlap': the all-gather of H in 2 / 4 chunks (a byte range per rank each) behind the next chunk of the F-solve, forced on at these sizes."""
    import dist_worker
    iters = 5 if mode == 'measure' else 3
    env = {} if mode == 'measure' else {'TRMF_CG': mode}
    if mode == 'persist':       # processes sharing ONE device may be time-sliced against each other: slow progress must not read as a failure here
        env['TRMF_PERSIST_TIMEOUT_MS'] = '120000'
    if mode == 'split':         # the split path of long rows (forced geometry): every rank covers the positions of the long-row list inside its block
        env = {'TRMF_LONG_ROW': '40', 'TRMF_LONG_CHUNK': '32', 'TRMF_CG': 'timeshard'}
    if mode == 'overlap':       # the F-solve in two launches, the first halves of H gathered on a side stream under the second
        env = {'TRMF_FOVERLAP': '4' if world == 3 else '2', 'TRMF_FSHARD': 'shard', 'TRMF_CG': 'timeshard'}     # chunks
    out = dict(_spawn(dist_worker.gpu_host_staged, world, iters, shape, env))
    p, m0 = dist_worker._problem(shape)
    for dtype in (np.float32, np.float64):
        name = np.dtype(dtype).name
        keep = {k: os.environ.get(k) for k in ('TRMF_LONG_ROW', 'TRMF_LONG_CHUNK')}
        if mode == 'split':     # the one-rank reference splits the same rows into the same items
            os.environ.update({k: env[k] for k in keep})
        try:
            model, cg1 = _single_process(p, m0, dtype, iters)
        finally:
            for k, v in keep.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        for r in range(world):
            W, H, Th, cg, second_session_same = out[r][name][:5]
            assert np.array_equal(W, model.W) and np.array_equal(H, model.H) and np.array_equal(Th, model.lag_val), (r, name)
            assert cg == cg1 and second_session_same
            if mode == 'split':
                assert 'split rows' in out[r][name][6], out[r][name][6]
            if mode == 'persist':
                assert 'one persistent kernel per rank' in out[r][name][6], out[r][name][6]


@pytest.mark.gpu
@pytest.mark.parametrize('world,shape,env,expect', [
    (2, 'c4', {}, 'available'), (4, 'c4', {}, 'available'), (8, 'c4', {}, 'available'), (3, 'odd', {}, 'available'),
    (2, 'c4', {'TRMF_P2P_FAIL': 'alloc'}, 'unavailable'),           # no rank gets its uncached arena
    (4, 'c4', {'TRMF_P2P_FAIL': 'export:2'}, 'unavailable'),        # ONE rank cannot export: every rank must fall back together
    (4, 'c4', {'TRMF_P2P_FAIL': 'open:1'}, 'unavailable'),          # one rank cannot map a peer
    (2, 'c4', {'TRMF_P2P_FAIL': 'fence:0'}, 'unavailable'),         # the trial exchange fails on one rank
    (2, 'c4', {'TRMF_NO_P2P': '1'}, 'unavailable'),
    (2, 'c4', {'TRMF_AUTOTUNE': '0'}, 'available'),                 # round-3 behaviour: the decisions inside the first iterations of run()
])
def test_default_path_measures_p2p_and_falls_back_safely(world, shape, env, expect):
    """The NO-FLAG multi-rank path (what `bench.py --gpus N` runs): the peer-to-peer transport is set up as a trial -- IPC arenas,
    mapping, a flags-only exchange with a 200 ms bound -- and becomes a third candidate of the measure-once rule next to the
    replicated CG and the time-sharded CG through the communicator; the decisions are taken in set-up iterations whose effect
    on W / H / Theta is undone, so the run is bit-identical to one process from its first iteration.  When any stage of the
    set-up fails on ANY rank (env hook), every rank drops the arena, says so in describe(), and the session continues with the
    communicator forms -- never an error, still bit-identical."""
    import dist_worker
    iters = 12 if env.get('TRMF_AUTOTUNE') == '0' else 3
    out = dict(_spawn(dist_worker.gpu_host_staged, world, iters, shape, env))
    p, m0 = dist_worker._problem(shape)
    for dtype in (np.float32, np.float64):
        name = np.dtype(dtype).name
        model, cg1 = _single_process(p, m0, dtype, iters)
        descs = set()
        for r in range(world):
            W, H, Th, cg, second_session_same, _, desc = out[r][name]
            assert np.array_equal(W, model.W) and np.array_equal(H, model.H) and np.array_equal(Th, model.lag_val), (r, name)
            assert cg == cg1 and second_session_same
            descs.add(desc)
        assert len(descs) == 1, descs                      # every rank took the same decisions from the same numbers
        desc = descs.pop()
        print('world %d %s %s %s: %s' % (world, shape, name, env, desc))
        assert 'peer-to-peer transport ' + expect in desc, desc
        assert 'measuring' not in desc and 'undecided' not in desc, desc
        if expect == 'available':
            assert 'time-sharded (peer to peer) ' in desc.split('[')[1], desc      # it was a measured candidate
        else:
            assert 'peer to peer' not in desc.split('[')[1], desc


@pytest.mark.gpu
def test_overlapped_h_gather_agrees_on_chunk_count_with_uneven_blocks():
    """the overlapped all-gather of H took its chunk count from the RANK'S OWN row count; an nnz-balanced
    partition gives the ranks different row counts, so near the size threshold they issued different numbers of
    collectives.  Here the threshold (TRMF_FOVERLAP_BYTES) is put between the smallest and the largest block of a skewed
    3-rank partition -- without TRMF_FOVERLAP, so the real rule runs; the old rule would have chosen 0 chunks on one rank and
    2 on another (a hang).  Now every rank derives the count from the largest block: same count, bit-identical result."""
    import scipy.sparse as smat
    import dist_worker
    from trmf import session
    world, shape, iters = 3, 'odd', 3
    p, m0 = dist_worker._problem(shape)
    bounds = [int(b) for b in session.partition_by_nnz(smat.csc_matrix(p['Y']).indptr, world, dtype=np.float64)]
    rows = [bounds[r + 1] - bounds[r] for r in range(world)]
    assert min(rows) < max(rows), rows
    KP = 16                                                  # k = 5 padded
    for dtype in (np.float32, np.float64):
        sz = np.dtype(dtype).itemsize
        thresh = (min(rows) + max(rows)) // 2 * KP * sz      # smallest block below, largest above
        assert min(rows) * KP * sz < thresh <= max(rows) * KP * sz
        env = {'TRMF_FOVERLAP_BYTES': str(thresh), 'TRMF_FSHARD': 'shard', 'TRMF_CG': 'timeshard'}
        out = dict(_spawn(dist_worker.gpu_host_staged, world, iters, shape, env, (np.dtype(dtype).name,)))
        model, cg1 = _single_process(p, m0, dtype, iters)
        for r in range(world):
            W, H, Th, cg, same, _, desc = out[r][np.dtype(dtype).name]
            assert np.array_equal(W, model.W) and np.array_equal(H, model.H) and np.array_equal(Th, model.lag_val), r
            assert cg == cg1 and same
            assert 'overlapped chunks' in desc, desc


_C3_REF = {}


def _c3_single_process_digests(iters):
    """W / H / Theta digests and CG counts of the single-process run of config 3 at full size (computed once)."""
    if iters not in _C3_REF:
        import hashlib
        from trmf import synth
        cfg = synth.CONFIGS['c3']
        p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=np.float32, seed=0)
        m0 = synth.initial_model(p['Y'], p['lag_set'], cfg['k'], seed=0)
        model, cg = _single_process(p, m0, np.float32, iters)
        _C3_REF[iters] = ([hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in (model.W, model.H, model.lag_val)], cg)
    return _C3_REF[iters]


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2, 4, 8])
def test_config4_full_size_sharded_paths_on_one_gpu(world):
    """BASELINE config 4 = config 3 (100k x 10k, 1 %, k=40, |L|=16, fp32) at its FULL size through the sharded ALS loop
    with 2, 4 and 8 ranks sharing the one GPU (host-staged exchange): F rows sharded, X-side Gram rows sharded, and the
    CG both replicated (Grams gathered) and sharded over time (tile records + halo rows exchanged per launch).  Every
    rank's factors must be bit-identical to the single-process run (compared through SHA-256 digests: the factors of the
    full problem stay in the workers).  The per-rank phase times are printed; they share one GPU, so only their sum over
    ranks is meaningful here (scripts/shard_compute_times.py measures a rank's share alone)."""
    import json
    import os
    import dist_worker
    iters = 2
    ref_dig, ref_cg = _c3_single_process_digests(iters)
    report = {}
    for mode in ('replicate', 'timeshard', 'p2p', 'auto') + (('persist',) if world <= 4 else ()):
        # 'auto': NO switch set -- the path bench.py --gpus N takes: every measure-once decision (F rows, X-side Gram rows, the form
        # of the CG with the peer-to-peer transport as a candidate) taken in set-up iterations on the full problem
        env = {} if mode == 'auto' else {'TRMF_CG': mode, 'TRMF_FSHARD': 'shard', 'TRMF_GRAMX': 'shard'}
        if mode == 'persist':   # (see test_time_sharded_cg_matches_single_process: time-slicing on a shared device is slow, not wrong)
            env['TRMF_PERSIST_TIMEOUT_MS'] = '120000'
        out = dict(_spawn(dist_worker.gpu_host_staged, world, iters, 'c3full', env, ('float32',)))
        for r in range(world):
            dig, _, _, cg, second_session_same, phases, desc = out[r]['float32']
            assert dig == ref_dig, (mode, r)
            assert cg == ref_cg and second_session_same
        if mode == 'auto':
            evidence('config 4 full size, %d ranks on one GPU, no switches: %s' % (world, out[0]['float32'][6]))
            assert 'peer-to-peer transport available' in out[0]['float32'][6] and 'measuring' not in out[0]['float32'][6]
        report[mode] = {r: out[r]['float32'][5] for r in range(world)}
        evidence('config 4 full size, %d ranks on one GPU, CG %s: CG %s; last iteration per rank (ms F / F kernel / X / Theta): %s' % (
            world, mode, ref_cg, ['%.2f/%.2f/%.2f/%.2f' % tuple(report[mode][r][-1]) for r in range(world)]))
    root = os.environ.get('GRAFT_REPO_ROOT')
    if root and os.path.isdir(os.path.join(root, 'gpurun_out')):
        with open(os.path.join(root, 'gpurun_out', 'c4_one_gpu_world%d.json' % world), 'w') as fh:
            json.dump(report, fh)


@pytest.mark.gpu
def test_two_ranks_one_gpu_config4_shape_replicated_cg():
    """BASELINE config 4's rank and lag set (k=40, |L|=16; sizes scaled to a one-GPU test): F rows and X-Gram rows
    sharded over two ranks, fused CG replicated -- bit-identical to the single-process run, in both precisions."""
    import dist_worker
    from trmf import session, synth
    iters = 3
    out = dict(_spawn(dist_worker.gpu_host_staged, 2, iters, 'c4', {'TRMF_CG': 'replicate'}))
    p, m0 = dist_worker._problem('c4')
    for dtype in (np.float32, np.float64):
        name = np.dtype(dtype).name
        model = make_model(m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
        with session.Session(p['Y'].astype(dtype), model, missing=True, **synth.HYPER) as s:
            s.run(iters); st = s.stats(iters); s.download()
        for r in (0, 1):
            W, H, Th, cg = out[r][name][:4]
            assert np.array_equal(W, model.W) and np.array_equal(H, model.H) and np.array_equal(Th, model.lag_val)
            assert cg == [x['cg_iter'] for x in st]


@pytest.mark.gpu
@pytest.mark.parametrize('world,transport', [(2, 'comm'), (4, 'comm'), (2, 'p2p'), (4, 'p2p'), (2, 'replicate')])
def test_time_sharded_unfused_cg(world, transport, monkeypatch):
    """The UNFUSED CG (long lag sets; forced here with TRMF_NO_HV_TILE) sharded over time: every kernel of the solve runs on the
    rank's own block of AR tiles, per step the ranks exchange midx edge rows of d, r, H d and their slots of the partial-sum
    arrays -- nothing T-sized.  All ranks bit-identical to each other; equal to the single-process unfused run up to the
    grouping of the partial sums (fp64 1e-9, fp32 1e-3), same CG counts.  'p2p': the exchange peer to peer (the partial-sum
    arrays and two alternating edge messages in the IPC-exported arena, pushed by a small kernel after each application)."""
    import dist_worker
    from trmf import session, synth
    iters = 3
    env = {'TRMF_NO_HV_TILE': '1'}
    if transport == 'p2p':
        env['TRMF_CG'] = 'p2p'
    if transport == 'replicate':    # the CG on every rank, the X-side Gram build sharded: the packed Grams are all-gathered
        env['TRMF_CG'] = 'replicate'; env['TRMF_GRAMX'] = 'shard'
    out = dict(_spawn(dist_worker.gpu_host_staged, world, iters, 'c4', env))
    p, m0 = dist_worker._problem('c4')
    monkeypatch.setenv('TRMF_NO_HV_TILE', '1')
    for dtype in (np.float32, np.float64):
        name = np.dtype(dtype).name
        model, cg1 = _single_process(p, m0, dtype, iters)
        W0, H0, T0, cg0 = out[0][name][:4]
        for r in range(1, world):
            W, H, Th, cg = out[r][name][:4]
            assert np.array_equal(W0, W) and np.array_equal(H0, H) and np.array_equal(T0, Th) and cg0 == cg
        tol = 1e-9 if dtype == np.float64 else 1e-3
        assert relfro(W0, model.W) < tol and relfro(H0, model.H) < tol and relfro(T0, model.lag_val) < 10 * tol
        assert all(abs(a - b) <= (0 if dtype == np.float64 else 1) for a, b in zip(cg0, cg1))
        assert out[0][name][4]      # second session under the same communicator


_C5S_REF = {}


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2, 4, 8])
@pytest.mark.parametrize('mode', ['comm', 'p2p', 'shard', 'auto'])
def test_config5_shape_multi_rank(world, mode):
    """Config 5's own kernel set with several ranks: k = 64, |L| = 32, fp64 -- the UNFUSED CG with KP = 64,
    `apply_kernel<true, 17>` on packed Grams, midx = 32, and the all-gather of H in 4 overlapped chunks under the F-solve (the size
    threshold lowered with TRMF_FOVERLAP_BYTES so that the REAL rule picks the chunks at this test's size; config 5's 64 MB blocks
    give 4 chunks by themselves).  'comm': time-sharded unfused CG, grouped exchange through the communicator; 'p2p': the same peer
    to peer; 'shard': the round-2 form (Gram product sharded, H d rows gathered per step); 'auto': no CG switch -- the
    measure-once rule chooses between the two transports.  Ranks bit-identical to each other, within 1e-9 of the single-process
    run (the partial sums of the product are grouped per rank), CG counts equal."""
    import dist_worker
    iters = 2
    env = {'TRMF_FOVERLAP_BYTES': str(1 << 20), 'TRMF_FSHARD': 'shard'}
    if mode == 'comm':
        env['TRMF_CG'] = 'timeshard'
    elif mode in ('p2p', 'shard'):
        env['TRMF_CG'] = mode
    out = dict(_spawn(dist_worker.gpu_host_staged, world, iters, 'c5s', env, ('float64',)))
    if 'ref' not in _C5S_REF:
        p, m0 = dist_worker._problem('c5s')
        _C5S_REF['ref'] = _single_process(p, m0, np.float64, iters)
    model, cg1 = _C5S_REF['ref']
    W0, H0, T0, cg0, same0, _, desc = out[0]['float64']
    evidence('config 5 shape, %d ranks, %s: %s' % (world, mode, desc))
    for r in range(1, world):
        W, H, Th, cg = out[r]['float64'][:4]
        assert np.array_equal(W0, W) and np.array_equal(H0, H) and np.array_equal(T0, Th) and cg0 == cg, r
    assert relfro(W0, model.W) < 1e-9 and relfro(H0, model.H) < 1e-9 and relfro(T0, model.lag_val) < 1e-8
    assert cg0 == cg1 and same0
    assert 'unfused CG' in desc and 'overlapped chunks' in desc, desc
    if mode == 'auto':
        assert 'measuring' not in desc and 'peer-to-peer transport available' in desc, desc


@pytest.mark.gpu
def test_two_ranks_one_gpu_sharded_cg_gram_product(monkeypatch):
    """The sharded form of the CG (unfused path: every rank multiplies its own timestamps' cached Grams, the rows of
    H d and the partial sums are all-gathered each step): both ranks bit-identical to each other, and
    equal to the single-process unfused run up to the summation order of the partials (fp64 gate 1e-9, fp32 1e-3)."""
    import dist_worker
    from trmf import session, synth
    iters = 3
    env = {'TRMF_NO_HV_TILE': '1', 'TRMF_CG': 'shard'}
    out = dict(_spawn(dist_worker.gpu_host_staged, 2, iters, 'c4', env))
    p, m0 = dist_worker._problem('c4')
    monkeypatch.setenv('TRMF_NO_HV_TILE', '1')
    for dtype in (np.float32, np.float64):
        name = np.dtype(dtype).name
        model = make_model(m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype)), p['lag_set'])
        with session.Session(p['Y'].astype(dtype), model, missing=True, **synth.HYPER) as s:
            s.run(iters); st = s.stats(iters); s.download()
        W0, H0, T0, cg0 = out[0][name][:4]
        W1, H1, T1, cg1 = out[1][name][:4]
        assert np.array_equal(W0, W1) and np.array_equal(H0, H1) and np.array_equal(T0, T1) and cg0 == cg1
        tol = 1e-9 if dtype == np.float64 else 1e-3
        assert relfro(W0, model.W) < tol and relfro(H0, model.H) < tol and relfro(T0, model.lag_val) < 10 * tol
        assert all(abs(a - b) <= (0 if dtype == np.float64 else 1) for a, b in zip(cg0, [x['cg_iter'] for x in st]))


RCCL_SCRIPT = r"""
import sys, os, ctypes
ROOT = sys.argv[1]
for p in (os.path.join(ROOT, 'exp-trmf-nips16_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
if sys.argv[2] == 'torch_first':
    import torch                      # one HIP/RCCL runtime per process: torch's, loaded first
    torch.cuda.init()
import numpy as np
from trmf import session, synth
from helpers import make_model
lib = session.lib_for(np.float32)
buf = ctypes.create_string_buffer(128)
assert lib.trmf_dist_get_unique_id(buf) == 0, lib.trmf_last_error()
assert lib.trmf_dist_init(0, 1, buf.raw) == 0, lib.trmf_last_error()
assert lib.trmf_dist_world() == 1 and lib.trmf_dist_rank() == 0
p = synth.sparse_problem(n=500, T=300, k=8, nlag=3, density=0.05, dtype=np.float32, seed=2)
m = synth.initial_model(p['Y'], p['lag_set'], 8, seed=2)
a = make_model(m.W, m.H, m.lag_val, p['lag_set'])
with session.Session(p['Y'], a, missing=True, **synth.HYPER) as s:
    s.run(1)
    lib.trmf_dist_finalize()          # the session keeps the communicator it was created under alive
    s.run(1); s.download()
b = make_model(m.W, m.H, m.lag_val, p['lag_set'])
with session.Session(p['Y'], b, missing=True, **synth.HYPER) as s:
    s.run(2); s.download()
assert np.array_equal(a.W, b.W) and np.array_equal(a.H, b.H)
print('RCCL_OK')
"""


@pytest.mark.gpu
@pytest.mark.parametrize('order', ['torch_first', 'standalone'])
def test_rccl_single_rank_communicator(order):
    """RCCL bootstrap + communicator lifecycle with world == 1 (all a one-GPU box can run), in a
    fresh process for each supported load order: torch imported first (bench.py --gpus N: the process
    binds to torch's bundled HIP/RCCL) or no torch at all (/opt/rocm's HIP/RCCL).  Loading this
    library first and torch afterwards puts two HIP runtimes in one process and is not supported."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run([sys.executable, '-c', RCCL_SCRIPT, root, order], capture_output=True, text=True,
                         timeout=400, env=env)
    assert 'RCCL_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
