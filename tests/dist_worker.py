"""Worker for the world_size>1 tests (launched by test_dist_*.py through torch.multiprocessing)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'exp-trmf-nips16_amd'), os.path.join(ROOT, 'oracle'), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


SHAPES = {
    'small': dict(n=900, T=400, k=12, nlag=4, density=0.06),
    'odd': dict(n=701, T=353, k=5, nlag=3, density=0.08),
    'c4': dict(n=3000, T=1200, k=40, nlag=16, density=0.04),       # BASELINE config 4's rank and lag set, scaled down
    # BASELINE config 5's rank, lag set and element type (k = 64, |L| = 32, fp64; sizes scaled to a one-GPU test): the unfused CG
    # with KP = 64, apply_kernel<true, 17> on packed Grams, midx = 32
    'c5s': dict(n=40000, T=6000, k=64, nlag=32, density=0.005),
}


def _problem(shape='small'):
    from trmf import synth
    c = SHAPES[shape]
    p = synth.sparse_problem(n=c['n'], T=c['T'], k=c['k'], nlag=c['nlag'], density=c['density'], dtype=np.float64, seed=21)
    m = synth.initial_model(p['Y'], p['lag_set'], c['k'], seed=21)
    return p, m


def cpu_sharded_fsolve(rank, world, port, out):
    """gloo, CPU only: every rank solves ITS row block of the F-solve (partition from the library's
    host logic) with the oracle, blocks are all-gathered in the library's layout; the result must be
    bit-identical to the unsharded solve."""
    import torch
    import torch.distributed as dist
    import oracle_py as O
    import scipy.sparse as smat
    from trmf import session, synth
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{}'.format(port), rank=rank, world_size=world)
    p, m = _problem()
    Yc = smat.csc_matrix(p['Y'])
    Yt = smat.csr_matrix((Yc.data, Yc.indices, Yc.indptr), shape=(Yc.shape[1], Yc.shape[0]))   # items x timestamps
    bounds = session.partition_by_nnz(Yc.indptr, world, dtype=np.float64)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    H = m.H.copy()
    blk = H[lo:hi].copy()
    O.fsolve_port(Yt[lo:hi], m.W, blk, synth.HYPER['lambdaI'])
    H[lo:hi] = blk
    # all-gather of uneven row blocks, root = owner (same protocol as csrc/comm.hpp)
    k = H.shape[1]
    for r in range(world):
        a, b = int(bounds[r]), int(bounds[r + 1])
        if b > a:
            piece = torch.from_numpy(H[a:b].copy() if r == rank else np.empty((b - a, k)))
            dist.broadcast(piece, src=r)
            H[a:b] = piece.numpy()
    ref = m.H.copy()
    O.fsolve_port(Yt, m.W, ref, synth.HYPER['lambdaI'])
    ok = bool(np.array_equal(H, ref)) and int(bounds[0]) == 0 and int(bounds[-1]) == Yt.shape[0]
    if rank == 0:
        out.put(('cpu_sharded_fsolve', ok, [int(b) for b in bounds]))
    dist.barrier()
    dist.destroy_process_group()


def gpu_host_staged(rank, world, port, out, iters, shape='small', env=None, dtypes=('float32', 'float64')):
    """`world` processes on ONE GPU, host-staged all-gather over gloo: the sharded device path must give
    results bit-identical across ranks and equal to the single-process run.  `env`: switches of the library
    (e.g. TRMF_CG=timeshard|replicate, TRMF_NO_HV_TILE to run the sharded Gram product of the unfused CG).
    shape 'c3full': BASELINE config 3 / 4 at its full size (100k x 10k, k=40, |L|=16)."""
    import torch.distributed as dist
    os.environ.update(env or {})
    from trmf import dist as tdist, session, synth
    from helpers import make_model
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{}'.format(port), rank=rank, world_size=world)
    res = {}
    for name in dtypes:
        dtype = np.dtype(name).type
        if os.environ.get('TRMF_TEST_DEVICE_PER_RANK'):      # a real multi-GPU node (scripts/scale_day.sh): a device per rank
            lib = session.lib_for(dtype)
            if lib.trmf_set_device(rank % max(1, lib.trmf_device_count())) != 0:
                raise RuntimeError(lib.trmf_last_error().decode())
        if shape == 'c3full':
            cfg = synth.CONFIGS['c3']
            p = synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=dtype, seed=0)
            m0 = synth.initial_model(p['Y'], p['lag_set'], cfg['k'], seed=0)
        else:
            p, m0 = _problem(shape)
        Y = p['Y'].astype(dtype)
        W0, H0, T0 = m0.W.astype(dtype), m0.H.astype(dtype), np.asfortranarray(m0.lag_val.astype(dtype))
        tdist.init_host_staged(dtype)
        model = make_model(W0, H0, T0, p['lag_set'])
        with session.Session(Y, model, missing=True, **synth.HYPER) as s:
            s.run(iters); st = s.stats(iters); s.download(); desc = s.describe()
        # a SECOND session under the same live communicator (its staging buffers remember the first session's stream,
        # which no longer exists): same result
        again = make_model(W0, H0, T0, p['lag_set'])
        with session.Session(Y, again, missing=True, **synth.HYPER) as s:
            s.run(iters); s.download(); desc2 = s.describe()
        same = bool(np.array_equal(again.W, model.W) and np.array_equal(again.H, model.H) and np.array_equal(again.lag_val, model.lag_val))
        # ... and it takes the first session's measure-once decisions from the process-level cache instead of measuring again
        # (round 5: what the second c_trmf_train call of a grid_search sees)
        if 'decided in' in desc and 'decided in 0 set-up' not in desc and os.environ.get('TRMF_AUTOTUNE') != '0':
            same = same and '(cached)' in desc2 and desc2.split('; decided')[0] == desc.split('; decided')[0]
        tdist.finalize(dtype)
        phases = [(x['ms_F'], x['ms_F_kernel'], x['ms_X'], x['ms_LV']) for x in st]
        if shape == 'c3full':       # the factors of the full problem stay in the worker: digests + a sample travel
            import hashlib
            dig = [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in (model.W, model.H, model.lag_val)]
            res[name] = (dig, None, None, [x['cg_iter'] for x in st], same, phases, desc)
        else:
            res[name] = (model.W.copy(), model.H.copy(), model.lag_val.copy(), [x['cg_iter'] for x in st], same, phases, desc)
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()
