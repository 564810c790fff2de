#!/usr/bin/env python3
"""bench.py -- ALS iterations/sec of the MI355X TRMF solver on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3] [--no-cpu-baseline]

A "step" is one ALS iteration (F-solve, X-solve, Theta-solve every 2nd iteration; trmf.cpp:647-693)
over the synthetic config-3 problem (100k x 10k, 1% dense, k=40, |L|=16, fp32),
with Y and the factors already resident in HBM when the timed region starts.  For N > 1 the driver
launches this file under torch.distributed.run, one rank per GPU; the item rows of the F-solve and
the timestamp rows of the X-side Gram build are partitioned across ranks inside the library (RCCL
all-gathers over xGMI), so total work is fixed: "scaling": "strong".

Rank 0 prints ONE JSON line.  `roofline` describes the F-solve kernel (HBM-bound under the
gather-inclusive algorithmic byte model B_F of SURVEY.md 8(d) / BASELINE.md section 3), timed with HIP
events recorded on the solver's own stream around that kernel.  `cpu_baseline` times the reference's
CPU path (oracle/_ref when present, else the C restatement) on a bounded sample on this box's host
cores; it is a reported baseline, not the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def physical_cores():
    """Physical cores of this host (hardware threads / threads-per-core), from /proc/cpuinfo."""
    try:
        pairs = set()
        phys = core = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        return len(pairs) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def make_problem(cfg):
    """(problem dict, hyper-parameters, missing flag) of a synth.CONFIGS entry."""
    import numpy as np
    from trmf import synth
    dtype = np.dtype(cfg['dtype'])
    if cfg.get('dense'):
        return synth.dense_problem(cfg['n'], cfg['T'], cfg['k'], cfg['lags'], dtype=dtype, seed=0), dict(cfg['hyper']), False
    return (synth.sparse_problem(cfg['n'], cfg['T'], cfg['k'], cfg['nlag'], cfg['density'], dtype=dtype, seed=0),
            dict(synth.HYPER), True)


def cpu_baseline_worker(config, iters, kind, threads):
    """Child process: time `iters` ALS iterations of the CPU path; prints one JSON line."""
    os.environ['OPENBLAS_NUM_THREADS'] = '1'   # before NumPy loads OpenBLAS: its pool fights OpenMP
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import oracle_py as O                      # test infrastructure: the checker / CPU baseline only
    from trmf import synth
    cfg = synth.CONFIGS[config]
    prob, hyper, missing = make_problem(cfg)
    m0 = synth.initial_model(prob['Y'], prob['lag_set'], cfg['k'], seed=0)
    W, H, Th = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    run = O.train_ref if kind == 'reference' else O.train_port
    t0 = time.perf_counter()
    run(prob['Y'], prob['lag_set'], W, H, Th, hyper, max_iter=iters, threads=threads, missing=missing)
    print(json.dumps({'seconds': time.perf_counter() - t0}))


def cpu_baseline(config, iters):
    """Reference CPU path on the same workload, in a child process (a crash there cannot take the
    GPU measurement down).  oracle/_ref (the real reference, OpenBLAS from the NumPy wheel) when it
    is present, else the C restatement.  The reference calls LAPACK posv from inside its OpenMP
    loop (trmf.cpp:371-396); the bundled OpenBLAS supports at most 64 calling threads, so the
    reference leg is capped at 64 threads."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    have_ref = os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'trmf_float32.so'))
    cores = physical_cores()
    for kind in (['reference'] if have_ref else []) + ['port']:
        threads = min(cores, 64) if kind == 'reference' else cores
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', kind,
                                  '--config', config, '--cpu-iters', str(iters), '--cpu-threads', str(threads)],
                                 capture_output=True, text=True, timeout=900)
            line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
            dt = json.loads(line)['seconds']
            return {'value': iters / dt, 'unit': 'iter/s', 'cores': threads, 'kind': kind,
                    'host_physical_cores': cores, 'seconds': dt,
                    'sample': '{} ALS iterations of the same {} workload from the same random start, {} OpenMP threads'.format(
                        iters, config, threads)}
        except Exception as exc:   # noqa: BLE001 - fall through to the next kind
            sys.stderr.write('cpu_baseline kind={} failed: {}\n'.format(kind, exc))
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='c3')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-iters', type=int, default=4)
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--cpu-baseline-worker', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.config, args.cpu_iters, args.cpu_baseline_worker, args.cpu_threads)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('WORLD_SIZE={} but --gpus {}'.format(world, args.gpus))

    dist = None
    use_dist = world > 1 or bool(os.environ.get('TRMF_BENCH_FORCE_DIST'))   # override: exercise this branch with 1 rank
    if use_dist:
        # torch first: its bundled HIP/RCCL runtimes must be the ones this process binds to
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))

    import numpy as np
    from trmf import session, synth

    cfg = dict(synth.CONFIGS[args.config]); cfg['name'] = args.config
    dtype = np.dtype(cfg['dtype'])
    lib = session.lib_for(dtype)
    if lib.trmf_device_count() < 1:
        raise SystemExit('no HIP device: the MI355X TRMF solver has no CPU fallback')
    if lib.trmf_set_device(local_rank) != 0:
        raise SystemExit(lib.trmf_last_error().decode())

    if use_dist:
        import ctypes
        ident = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            if lib.trmf_dist_get_unique_id(buf) != 0:
                raise SystemExit(lib.trmf_last_error().decode())
            ident = [buf.raw]
        dist.broadcast_object_list(ident, src=0)
        if lib.trmf_dist_init(rank, world, ident[0]) != 0:
            raise SystemExit(lib.trmf_last_error().decode())

    t_gen = time.perf_counter()
    prob, hyper, missing = make_problem(cfg)
    model = synth.initial_model(prob['Y'], prob['lag_set'], cfg['k'], seed=0)
    t_gen = time.perf_counter() - t_gen

    def barrier():
        if dist is not None:
            dist.barrier()

    def device_sync(s):
        s.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()

    t_up = time.perf_counter()
    # verbose=0 run of the reference: no ||.||^2 log lines (trmf.cpp:659-688 evaluates them only under verbose)
    s = session.Session(prob['Y'], model, missing=missing, log_norms=False, **hyper)
    t_up = time.perf_counter() - t_up
    s.run(args.warmup)
    device_sync(s); barrier()
    t0 = time.perf_counter()
    s.run(args.steps)
    device_sync(s); barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = s.stats(args.steps)
    bytes_f = s.fsolve_bytes()
    ms_fk = float(np.mean([x['ms_F_kernel'] for x in st]))
    s.download()
    s.close()

    if rank == 0:
        nnz = int(prob['Y'].nnz) if hasattr(prob['Y'], 'nnz') else int(prob['Y'].size)
        achieved = bytes_f / (ms_fk * 1e-3) / 1e9 if ms_fk > 0 else 0.0
        traffic = None      # HBM bytes per launch from the committed PMC profile of this config (1 GPU only)
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'fsolve_traffic.json')))
            if tj.get('config') == args.config and world == 1:
                traffic = tj['traffic_bytes']
        except (OSError, ValueError, KeyError):
            pass
        out = {
            'metric': 'als_iterations_per_sec', 'value': args.steps / elapsed, 'unit': 'iter/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32' if dtype == np.float32 else 'f64',
            'data': 'synthetic',
            'config': {'workload': '{}: n={} T={} density={} nnz={} k={} |L|={} {} missing={} lambdaI={} lambdaAR={} lambdaLag={}'.format(
                args.config, cfg['n'], cfg['T'], cfg.get('density', 1.0), nnz, cfg['k'], len(prob['lag_set']), dtype.name,
                int(missing), hyper['lambdaI'], hyper['lambdaAR'], hyper['lambdaLag']),
                'parallelism': 'F rows / X-Gram rows sharded x{}, X-Gram build replicated instead when its all-gather costs more than it saves (decided once, after the first measured iteration), CG replicated'.format(world)},
            'roofline': {'kernel': 'fsolve_quad_kernel<3,40>' if dtype == np.float32 else 'fsolve_kernel', 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': bytes_f, 'avg_kernel_ms': ms_fk,
                         'byte_model': 'nnz*(4+s+k*s) + (rows+1)*8 + rows*k*s over this rank\'s item rows'},
            'phases_ms': {'F': float(np.mean([x['ms_F'] for x in st])), 'X': float(np.mean([x['ms_X'] for x in st])),
                          'Theta': float(np.mean([x['ms_LV'] for x in st])),
                          'cg_iter': [int(x['cg_iter']) for x in st]},
            'setup_s': {'generate': t_gen, 'upload_and_alloc': t_up},
        }
        if not args.no_cpu_baseline and world == 1:
            base = cpu_baseline(args.config, args.cpu_iters)
            if base is not None:
                out['cpu_baseline'] = base
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        lib.trmf_dist_finalize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
