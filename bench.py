#!/usr/bin/env python3
"""bench.py -- ALS iterations/sec of the MI355X TRMF solver on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3] [--no-cpu-baseline] [--cg replicate|timeshard|p2p|persist]

A "step" is one ALS iteration (F-solve, X-solve, Theta-solve every 2nd iteration; trmf.cpp:647-693)
over the synthetic config-3 problem (100k x 10k, 1% dense, k=40, |L|=16, fp32),
with Y and the factors already resident in HBM when the timed region starts.  For N > 1 the driver
launches this file under torch.distributed.run, one rank per GPU; the item rows of the F-solve and
the timestamp rows of the X-side Gram build are partitioned across ranks inside the library (RCCL
all-gathers over xGMI), so total work is fixed: "scaling": "strong".

Rank 0 prints ONE JSON line.  `roofline` describes the F-solve kernel (HBM-bound under the
gather-inclusive algorithmic byte model B_F of SURVEY.md 8(d) / BASELINE.md section 3), timed with HIP
events recorded on the solver's own stream around that kernel -- on every third iteration of the timed window (--timing 3: an
iteration's seven phase events are barrier packets that cost ~25 us, so they are sampled; `value` is wall clock over ALL iterations;
profiles/r05b_events.txt) --; `roofline_compute` prices the same launch in useful
arithmetic against the dtype's matrix peak (config 5's fp64 F-solve is bound by that, not by HBM).  `cpu_baseline` times the reference's
CPU path (oracle/_ref when present, else the C restatement) on this box's host cores with the protocol of
BASELINE.md section 3 (min(physical, 64) and 8 threads, 2 warm-up + 10 timed iterations from the state the
GPU's timed window started from, F / X / Theta split); it is a reported baseline, not the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'exp-trmf-nips16_amd'))

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
kParityIters = 2            # iterations of the GPU-vs-CPU parity figure carried by the cpu_baseline leg


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
    return synth.make(cfg, seed=0), dict(cfg.get('hyper', synth.HYPER)), not cfg.get('dense')


def fsolve_source_digest():
    """sha256 over the sources the F-solve kernel is built from: profiles/fsolve_traffic.json records the digest of
    the tree its PMC pass was taken on, and `roofline.traffic` is reported only while the two agree."""
    import hashlib
    h = hashlib.sha256()
    for name in ('common.hpp', 'gram_kernels.hpp'):
        with open(os.path.join(ROOT, 'exp-trmf-nips16_amd', 'csrc', name), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def cpu_baseline_worker(config, iters, kind, threads, state_file, gpu_file=''):
    """Child process: the CPU path on `threads` OpenMP threads, warm-started from the state the GPU's timed window
    started from (state_file: W, H, Theta after the GPU's warm-up iterations).  Protocol of BASELINE.md section 3:
    2 warm-up iterations (also from that state, discarded), then `iters` timed full ALS iterations, wall clock around
    the c_trmf_train-equivalent call; then the F / X / Theta split, each phase alone via period_* > max_iter
    (2 calls each, same state).  Prints one JSON line."""
    os.environ['OPENBLAS_NUM_THREADS'] = '1'   # before NumPy loads OpenBLAS: its pool fights OpenMP
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import oracle_py as O                      # test infrastructure: the checker / CPU baseline only
    from trmf import synth
    cfg = synth.CONFIGS[config]
    prob, hyper, missing = make_problem(cfg)
    if state_file and os.path.exists(state_file):
        z = np.load(state_file)
        W0, H0, T0 = z['W'], z['H'], np.asfortranarray(z['lag_val'])
    else:
        m0 = synth.initial_model(prob['Y'], prob['lag_set'], cfg['k'], seed=0)
        W0, H0, T0 = m0.W.copy(), m0.H.copy(), np.asfortranarray(m0.lag_val.copy())
    run = O.train_ref if kind == 'reference' else O.train_port
    big = 10 ** 6

    def timed(n_iter, periods):
        W, H, Th = W0.copy(), H0.copy(), np.asfortranarray(T0.copy())
        t0 = time.perf_counter()
        run(prob['Y'], prob['lag_set'], W, H, Th, hyper, max_iter=n_iter, periods=periods, threads=threads, missing=missing)
        return time.perf_counter() - t0

    def relfro(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

    timed(2, (1, 1, 2))                                            # warm-up: page in, spin up the thread pool
    full = timed(iters, (1, 1, 2))
    parity = None
    if gpu_file and os.path.exists(gpu_file) and missing:
        # the GPU ran the same kParityIters iterations from the same state (bench.py main): objective of both results by
        # the oracle's fp64 evaluator, and the distance of the factors.  Two iterations, not ten: in fp32 the truncated
        # CG amplifies rounding differences from iteration to iteration -- at config 3 the reference's own fp32 build is
        # 1.5e-4 (objective) / 2.7e-3 (W) away from its line-by-line restatement after ten iterations, and both are
        # 5e-4 / 2e-2 away from the fp64 run (profiles/r02_fp32_trajectory_c3.txt) -- so only a short horizon measures
        # the implementation rather than the noise floor of the problem
        g = np.load(gpu_file)
        W, H, Th = W0.copy(), H0.copy(), np.asfortranarray(T0.copy())
        run(prob['Y'], prob['lag_set'], W, H, Th, hyper, max_iter=kParityIters, periods=(1, 1, 2), threads=threads, missing=missing)
        Jc = O.objective(prob['Y'], prob['lag_set'], W, H, Th, hyper)
        Jg = O.objective(prob['Y'], prob['lag_set'], g['W'], g['H'], g['lag_val'], hyper)
        parity = {'iters': kParityIters, 'J_cpu': Jc, 'J_gpu': Jg, 'rel_diff': abs(Jg - Jc) / abs(Jc),
                  'relfro_W': relfro(g['W'], W), 'relfro_H': relfro(g['H'], H), 'relfro_Theta': relfro(g['lag_val'], Th)}
    split_calls = 2
    t_f = timed(split_calls, (big, 1, big)) / split_calls          # (period_W, period_H, period_Lag)
    t_x = timed(split_calls, (1, big, big)) / split_calls
    t_l = timed(split_calls, (big, big, 1)) / split_calls
    print(json.dumps({'seconds': full, 'iters': iters, 'threads': threads, 's_per_F': t_f, 's_per_X': t_x, 's_per_Theta': t_l,
                      'parity': parity}))


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(config, iters, state_file, gpu_file=''):
    """Reference CPU path on the same workload, in child processes (a crash there cannot take the GPU measurement
    down): oracle/_ref (the real reference, OpenBLAS from the NumPy wheel) when present, else the C restatement;
    once on min(physical cores, 64) threads and once on 8.  The reference calls LAPACK posv
    from inside its OpenMP loop (trmf.cpp:371-396) and the bundled OpenBLAS supports at most 64 calling threads,
    hence the cap.  `value` is the faster of the two runs; both are listed."""
    import subprocess
    have_ref = os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'trmf_float32.so'))
    cores = physical_cores()
    results = {}
    for kind in (['reference'] if have_ref else []) + ['port']:
        runs = []
        for threads in sorted({min(cores, 64), min(cores, 8)}, reverse=True):
            try:
                res = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', kind,
                                      '--config', config, '--cpu-iters', str(iters), '--cpu-threads', str(threads),
                                      '--cpu-state', state_file or '', '--cpu-gpu-result', (gpu_file or '') if kind == ('reference' if have_ref else 'port') else ''],
                                     capture_output=True, text=True, timeout=1200)
                line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
                r = json.loads(line)
                runs.append({'threads': threads, 'iter_per_s': r['iters'] / r['seconds'], 'seconds': r['seconds'], 'iters': r['iters'],
                             's_per_F_solve': r['s_per_F'], 's_per_X_solve': r['s_per_X'], 's_per_Theta_solve': r['s_per_Theta'],
                             'parity_vs_gpu': r.get('parity')})
            except Exception as exc:   # noqa: BLE001 - keep whatever else succeeded
                sys.stderr.write('cpu_baseline kind={} threads={} failed: {}\n'.format(kind, threads, exc))
        if runs:
            best = max(runs, key=lambda r: r['iter_per_s'])
            results[kind] = {'value': best['iter_per_s'], 'unit': 'iter/s', 'cores': best['threads'], 'kind': kind, 'runs': runs}
    if not results:
        return None
    # the headline baseline is the real reference build when it travelled with the tree (oracle/_ref), else the restatement; the
    # other one is listed beside it (BASELINE.md section 3 asks for the build's own OpenMP restatement on all cores and on 8)
    head = dict(results.get('reference') or results['port'])
    head.update({'host_physical_cores': cores, 'cpu_model': cpu_model_name(),
                 'sample': '{} timed ALS iterations of the same {} workload after 2 warm-up iterations, warm-started from the '
                           'factors the GPU\'s timed window started from; per-phase seconds from single-phase calls '
                           '(period_* > max_iter)'.format(iters, config)})
    if 'reference' in results and 'port' in results:
        head['restatement'] = results['port']
    return head


def x_byte_models(cfg, nnz, itemsize):
    """Algorithmic bytes of the X phase (SURVEY.md 8(d) "X-side figures"): the Gram build B_G = nnz*(4+s+k*s) + (T+1)*8 +
    T*(k^2+k)*s, and one CG pass over the cached Grams B_cg = T*k^2*s + 6*T*k*s."""
    T, k, s = cfg['T'], cfg['k'], itemsize
    return float(nnz) * (4 + s + k * s) + (T + 1) * 8.0 + float(T) * (k * k + k) * s, float(T) * k * k * s + 6.0 * T * k * s


def roofline_x(cfg, nnz, dtype, missing, world, ms_xg, ms_x, cg_steps, described):
    """The X phase priced like the F-solve: the Gram build under B_G, the CG solve as (steps, us per pass) against B_cg --
    both from HIP events on the solver stream (TrmfIterStats.ms_X_gram / ms_X).  One rank, observed-entries path only."""
    if world != 1 or not missing or ms_xg <= 0 or ms_x <= ms_xg:
        return None
    b_g, b_cg = x_byte_models(cfg, nnz, dtype.itemsize)
    ms_cg = ms_x - ms_xg
    passes = cg_steps + 1.0            # gradient + one product per CG step (the closing H s pass is gone since round 6: DESIGN.md 4.4)
    us_pass = 1e3 * ms_cg / passes
    return {'gram': {'kernel': 'gram_x_kernel', 'bound': 'hbm', 'algorithmic_bytes_per_launch': b_g, 'avg_ms': ms_xg,
                     'achieved': b_g / (ms_xg * 1e-3) / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': b_g / (ms_xg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     'byte_model': 'nnz*(4+s+k*s) + (T+1)*8 + T*(k^2+k)*s'},
            'cg': {'form': described, 'avg_ms_per_solve': ms_cg, 'cg_steps_per_solve': cg_steps, 'operator_passes_per_solve': passes,
                   'us_per_pass': us_pass, 'algorithmic_bytes_per_pass': b_cg, 'achieved': b_cg / (us_pass * 1e-6) / 1e9, 'peak': HBM_PEAK_GBPS,
                   'unit': 'GB/s', 'frac': b_cg / (us_pass * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                   'byte_model': 'T*k^2*s + 6*T*k*s per pass over the cached Grams (the persistent kernel keeps its Gram slice in registers: '
                                 'its HBM traffic per pass is the exchanged rows only, a pass is a latency chain -- DESIGN.md 4.4)'}}


def measure_one_shot(prob, cfg, hyper, missing, dtype, iters, calls=4):
    """The reference's own entry point as a caller of the reference API meets it: c_trmf_train(max_iter=iters) on host arrays --
    upload, train, download -- wall clock around the C call, with the library's own split of it (trmf_last_train_profile).  The
    first call of the process pays one-time costs (pinned ring, first slab of the pool); `steady` is the fastest later call
    (what the second and later calls of a grid_search see)."""
    import numpy as np
    import trmf.trmf as front
    from trmf import session, synth
    from trmf.rf_util import PyMatrix
    pyY = PyMatrix(prob['Y'], dtype=dtype)
    out = []
    for c in range(calls):
        m = synth.initial_model(prob['Y'], prob['lag_set'], cfg['k'], seed=0)
        t0 = time.perf_counter()
        front.get_clib().train(pyY, m.lag_set, m.pyW, m.pyH, m.pylag_val, warm_start=True, max_iter=iters, missing=missing, **hyper)
        wall = time.perf_counter() - t0
        pr = session.train_profile(dtype) or {}
        out.append({'wall_ms': 1e3 * wall, 'setup_ms': 1e3 * pr.get('setup_s', 0), 'upload_ms': 1e3 * pr.get('upload_s', 0),
                    'compute_ms': 1e3 * pr.get('compute_s', 0), 'download_ms': 1e3 * pr.get('download_s', 0),
                    'teardown_ms': 1e3 * pr.get('teardown_s', 0), 'bytes_h2d': pr.get('bytes_h2d'), 'bytes_d2h': pr.get('bytes_d2h'),
                    'device_mallocs': pr.get('device_mallocs'), 'failed': pr.get('failed')})
    steady = min(out[1:], key=lambda r: r['wall_ms']) if len(out) > 1 else out[0]
    return {'entry': 'c_trmf_train(max_iter={})'.format(iters), 'first_call_after_session': out[0], 'steady': steady, 'calls': out,
            'non_compute_ms_steady': steady['wall_ms'] - steady['compute_ms'],
            'pcie_floor_ms': (steady['bytes_h2d'] or 0) / 56e9 * 1e3,
            'note': 'wall clock around the C entry on host (NumPy) arrays; PyMatrix construction (the Python wrapper\'s CSR/CSC conversion) is outside'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='c3')
    ap.add_argument('--timing', type=int, default=3,
                    help='phase events (F / X / Theta split, kernel times of the roofline objects) on every N-th iteration of the timed '
                         'window; 1 = every iteration.  The seven event records of an iteration cost 21-26 us (2.5 %% of a config-3 '
                         'iteration, 22 %% of a config-2 one): the default samples every third iteration, which alternates between '
                         'iterations with and without a Theta-solve')
    ap.add_argument('--repeat', type=int, default=0,
                    help='run the timed window of --steps iterations R times, every time from the SAME post-warm-up state (the session is rewound '
                         'on the device between windows, outside the timed regions); `value` is over all windows, `windows` lists the median / '
                         'min / max of the individual ones.  0 (default): as many windows as make the timed region >= --min-timed-s seconds '
                         '(a 17 ms window is invisible to an external utilisation sampler); 1: a single window')
    ap.add_argument('--min-timed-s', type=float, default=2.0)
    ap.add_argument('--devices', default=None,
                    help='without a launcher: the device list of the in-process multi-GPU mode (TRMF_DEVICES; default for --gpus N: 0..N-1).  A device '
                         'may be listed several times ("0,0": virtual ranks on one GPU -- a dry run of the multi-GPU path on a one-GPU box)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cg', default=None, choices=['replicate', 'timeshard', 'p2p', 'persist', 'shard'],
                    help='multi-GPU CG form (sets TRMF_CG; default: the library measures replicated vs time-sharded)')
    ap.add_argument('--no-one-shot', action='store_true', help='skip the c_trmf_train (upload + train + download) leg')
    ap.add_argument('--one-shot-iters', type=int, default=10)
    ap.add_argument('--cpu-iters', type=int, default=10)
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--cpu-state', default='', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-gpu-result', default='', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-baseline-worker', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.config, args.cpu_iters, args.cpu_baseline_worker, args.cpu_threads, args.cpu_state,
                                   args.cpu_gpu_result)

    if args.cg:
        os.environ['TRMF_CG'] = args.cg
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('WORLD_SIZE={} but --gpus {}'.format(world, args.gpus))
    # No launcher (WORLD_SIZE unset) and --gpus N > 1: ONE process, the ranks are threads of the library (TRMF_DEVICES; csrc/session_group.hpp)
    # -- what a caller of the reference API gets.  Under torchrun (the driver's multi-GPU runs) every process is one rank as before.
    inproc = 0
    if world == 1 and (args.gpus > 1 or args.devices):
        devs = args.devices or ','.join(str(i) for i in range(args.gpus))
        os.environ['TRMF_DEVICES'] = devs
        inproc = len([d for d in devs.split(',') if d.strip() != ''])
        if args.gpus > 1 and inproc != args.gpus:
            raise SystemExit('--devices lists {} ranks but --gpus {}'.format(inproc, args.gpus))

    dist = None
    replicas_note = None
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
        # The library's own communicator (dlopen()ed RCCL).  If it cannot be set up on ANY rank -- a first on real hardware is a first --
        # every rank drops it and the job runs as N replicas of the one-GPU solver (same K iterations, no sharding: a valid, if
        # unimpressive, strong-scaling figure instead of no figure); config.parallelism says so.
        import torch
        init_rc = lib.trmf_dist_init(rank, world, ident[0]) if not os.environ.get('TRMF_BENCH_FAIL_DIST_INIT') else 1
        init_err = lib.trmf_last_error().decode() if init_rc != 0 else ''
        flag = torch.tensor([1 if init_rc == 0 else 0], dtype=torch.int32, device='cuda')
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if init_rc == 0:
                lib.trmf_dist_finalize()
            replicas_note = '{} replicas of the one-GPU solver (the library communicator could not be set up on every rank{})'.format(
                world, ': ' + init_err if init_err else '')
            if rank == 0:
                print('bench.py: ' + replicas_note, file=sys.stderr)

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
    if args.steps < 2 * max(1, args.timing):       # a window too short to hold two sampled iterations: events on every iteration
        args.timing = 1
    s = session.Session(prob['Y'], model, missing=missing, log_norms=False, timing=max(1, args.timing), **hyper)
    t_up = time.perf_counter() - t_up
    s.run(args.warmup)
    device_sync(s)

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # The timed region: R windows of EXACTLY --steps iterations, each bracketed by a barrier + device synchronisation on both sides
    # and each starting from the same state (mark / rewind on the device, outside the brackets) -- so every window does the same
    # work (same CG step counts) and the windows' spread is the measurement's noise.  The maximum over ranks is taken per window.
    can_rewind = hasattr(s.lib, 'trmf_session_mark')
    if not can_rewind:
        args.repeat = 1                         # a library from before round 6 (A/B runs): one window
    if args.repeat != 1:
        s.mark()
    windows = []
    repeat = max(1, args.repeat)
    w = 0
    while w < repeat:
        if w > 0:
            s.rewind()
        device_sync(s); barrier()
        t0 = time.perf_counter()
        s.run(args.steps)
        device_sync(s); barrier()
        windows.append(max_over_ranks(time.perf_counter() - t0))
        w += 1
        if args.repeat == 0 and w == 1:        # auto: from the first window's time (identical on every rank after the reduction)
            repeat = int(min(1000, max(1, -(-args.min_timed_s // max(windows[0], 1e-6)))))
    elapsed = float(sum(windows))
    total_steps = args.steps * len(windows)

    state_file = None
    if rank == 0 and world == 1 and not inproc and not args.no_cpu_baseline:
        # The factors the timed window started from (the CPU baseline is warm-started from the same state), reproduced AFTER the
        # timed window by a second session that repeats the warm-up from the same initial model -- the solver is deterministic (every
        # sum in a fixed order; tests/test_gpu_parity.py), so this is the state bit for bit.  Rounds 1-4 downloaded and saved it
        # between warm-up and timed window: ~0.2 s of idle GPU in front of a 20 ms window, which then started at idle clocks
        # (1022-1038 iter/s against 1060-1068 for the same build with --no-cpu-baseline).
        import tempfile
        m_w = synth.initial_model(prob['Y'], prob['lag_set'], cfg['k'], seed=0)
        with session.Session(prob['Y'], m_w, missing=missing, log_norms=False, **hyper) as s_w:
            s_w.run(args.warmup).download()
        state_file = os.path.join(tempfile.gettempdir(), 'trmf_bench_state_{}.npz'.format(os.getpid()))
        np.savez(state_file, W=m_w.W, H=m_w.H, lag_val=m_w.lag_val)
    one_shot = None
    if rank == 0 and world == 1 and not args.no_one_shot:
        # (under --gpus N without a launcher this is c_trmf_train itself on N devices: every rank uploads the problem)
        one_shot = measure_one_shot(prob, cfg, hyper, missing, dtype, args.one_shot_iters)
    st = s.stats(args.steps)
    described = s.describe()          # which phases are sharded / which CG form the measure-once rule chose (set-up iterations, untimed)
    if replicas_note:
        described = replicas_note + '; each: ' + described
    bytes_f = s.fsolve_bytes()
    timed = [x for x in st if x['ms_F'] >= 0]              # the iterations of the window that carried phase events (--timing)
    have_events = bool(timed)
    if not have_events:                                    # none did: phases / rooflines are reported as absent (null), not as averages of the -1 sentinels
        timed = st
    ms_fk = float(np.mean([x['ms_F_kernel'] for x in timed])) if have_events else 0.0
    ms_xg = float(np.mean([x['ms_X_gram'] for x in timed])) if have_events else 0.0
    ms_x = float(np.mean([x['ms_X'] for x in timed])) if have_events else 0.0
    cg_steps = float(np.mean([x['cg_iter'] for x in timed]))   # of the same iterations (us per pass = their CG time / their passes)
    s.download()
    s.close()

    # SURVEY.md 8(d)'s protocol beside the driver's flags: 2 warm-up + 10 timed iterations FROM THE RANDOM START (the early iterations
    # run their CG to the 20-step cap, later ones stop at 12-14: a different mix of work from a window after --warmup iterations)
    survey = None
    if world == 1 and can_rewind:
        m_s = synth.initial_model(prob['Y'], prob['lag_set'], cfg['k'], seed=0)
        with session.Session(prob['Y'], m_s, missing=missing, log_norms=False, timing=0, **hyper) as s_s:
            s_s.run(2).sync()
            s_s.mark()
            ts = []
            for _ in range(5):
                s_s.rewind()
                t1 = time.perf_counter(); s_s.run(10).sync(); ts.append(time.perf_counter() - t1)
            cg_s = [int(x['cg_iter']) for x in s_s.stats(10)]
        survey = {'warmup': 2, 'steps': 10, 'from': 'the random start (Model.initialize, seed 0)', 'iter_per_s': 10.0 / float(np.median(ts)),
                  'windows': len(ts), 'iter_per_s_min': 10.0 / max(ts), 'iter_per_s_max': 10.0 / min(ts), 'cg_iter': cg_s}

    if rank == 0:
        nnz = int(prob['Y'].nnz) if hasattr(prob['Y'], 'nnz') else int(prob['Y'].size)
        achieved = bytes_f / (ms_fk * 1e-3) / 1e9 if ms_fk > 0 else 0.0
        # HBM bytes per launch from the committed PMC profile of this config (1 GPU only) -- reported only while the
        # kernel sources are the ones that profile was taken on (digest recorded by scripts/make_traffic_json.py)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'fsolve_traffic.json')))
            tj = tj.get(args.config, tj if tj.get('config') == args.config else {})       # one entry per configuration
            if world == 1 and not inproc and tj.get('kernel_source_sha256') == fsolve_source_digest():
                traffic = tj['traffic_bytes']
        except (OSError, ValueError, KeyError):
            pass
        out = {
            'metric': 'als_iterations_per_sec', 'value': total_steps / elapsed, 'unit': 'iter/s',
            'n_gpus': inproc if inproc else world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / total_steps, 'higher_is_better': True,
            # R windows of `steps` iterations each, all from the same post-warm-up state; value = R * steps / (sum of the windows' times)
            'value_survey_protocol': survey,
            # (the first window starts at whatever clocks the warm-up left: it is listed by itself; `spread` = (max - min) / median of the others)
            'windows': (lambda r, q: {'repeat': len(windows), 'steps_each': args.steps, 'timed_region_s': elapsed, 'iter_per_s_median': float(np.median(r)),
                                      'iter_per_s_min': float(np.min(r)), 'iter_per_s_max': float(np.max(r)), 'spread': float((np.max(q) - np.min(q)) / np.median(q)),
                                      'p05_p95': [float(np.percentile(q, 5)), float(np.percentile(q, 95))],
                                      'first_window_iter_per_s': float(r[0])})(np.array([args.steps / t for t in windows]),
                                                                               np.array([args.steps / t for t in (windows[1:] if len(windows) > 2 else windows)])),
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32' if dtype == np.float32 else 'f64',
            'data': 'synthetic',
            'config': {'workload': '{}: n={} T={} density={} nnz={} k={} |L|={} {} missing={} lambdaI={} lambdaAR={} lambdaLag={}'.format(
                args.config, cfg['n'], cfg['T'], cfg.get('density', 1.0), nnz, cfg['k'], len(prob['lag_set']), dtype.name,
                int(missing), hyper['lambdaI'], hyper['lambdaAR'], hyper['lambdaLag']),
                # chosen by the library's measure-once rules in set-up iterations BEFORE the warm-up (their effect on the factors is undone;
                # DESIGN.md section 6): F rows / X-side Gram rows sharded or replicated, the CG replicated or sharded over time with the
                # exchange through RCCL or peer to peer -- with the slowest rank's measured X phase of every candidate
                'parallelism': described,
                'phases_ms_rank0': None if not have_events else {'F': float(np.mean([x['ms_F'] for x in timed])), 'X': float(np.mean([x['ms_X'] for x in timed])),
                                    'Theta': float(np.mean([x['ms_LV'] for x in timed]))},
                'degraded': 'replicas' if replicas_note else None},
            'roofline': None if not have_events else {'kernel': ('fsolve_quad_kernel' if dtype == np.float32 else 'fsolve_mfma_kernel') + '<{},{}>'.format((cfg['k'] + 15) // 16, (cfg['k'] + 7) // 8 * 8)
                                   + (' + the split path of long rows (gram_part_kernel, split_reduce_kernel, fsolve_*_long_kernel)' if 'split rows' in described else ''),
                         'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': bytes_f, 'avg_kernel_ms': ms_fk,
                         'byte_model': 'nnz*(4+s+k*s) + (rows+1)*8 + rows*k*s over this rank\'s item rows',
                         'traffic_source': None if traffic is None else 'profiles/fsolve_traffic.json: a PMC pass of this kernel taken by the builder on the '
                                           'same sources (digest-guarded), NOT measured by this run; achieved / avg_kernel_ms are this run\'s HIP events'},
            # the same launch priced in arithmetic (one rank only): upper triangle of the k x k Gram + rhs per observed entry,
            # k^3/3 + 2 k^2 per system; fp64's MFMA pipe issues one v_mfma_f64_16x16x4 per ~100 cycles per SIMD on this chip
            # (profiles/r03_f64_pipe_ubench.txt) -- config 5's F-solve is bound by THAT, not by HBM
            'roofline_compute': None if world != 1 or inproc or not missing or ms_fk <= 0 else (lambda fl, pk, pm: {
                'kernel': 'the F-solve kernel of `roofline`', 'bound': 'mfma', 'achieved': fl / (ms_fk * 1e-3) / 1e12, 'peak': pk, 'unit': 'TFLOP/s',
                'frac': fl / (ms_fk * 1e-3) / 1e12 / pk, 'peak_issue_rate_measured': pm, 'frac_of_measured_issue_rate': fl / (ms_fk * 1e-3) / 1e12 / pm,
                'algorithmic_flops_per_launch': fl, 'flop_model': 'nnz*(k*(k+1) + 2*k) + rows*(k^3/3 + 2*k^2); padded MFMA tiles are not counted'})(
                    float(nnz) * (cfg['k'] * (cfg['k'] + 1) + 2 * cfg['k']) + float(cfg['n']) * (cfg['k'] ** 3 / 3.0 + 2.0 * cfg['k'] ** 2),
                    157.3 if dtype == np.float32 else 78.6, 155.0 if dtype == np.float32 else 50.3),
            'roofline_x': roofline_x(cfg, nnz, dtype, missing, max(world, inproc), ms_xg, ms_x, cg_steps, described),
            'one_shot': one_shot,
            'phases_ms': {'F': float(np.mean([x['ms_F'] for x in timed])) if have_events else None, 'X': float(np.mean([x['ms_X'] for x in timed])) if have_events else None,
                          'Theta': float(np.mean([x['ms_LV'] for x in timed])) if have_events else None,
                          'cg_iter': [int(x['cg_iter']) for x in st],
                          'timed_iterations': len(timed) if have_events else 0, 'of': len(st),
                          'note': 'HIP events of the iterations that carried them (every %d-th of the window; each such iteration pays seven event '
                                  'records, ~25 us, the others none -- value / ms_per_step are wall clock over ALL iterations); F / X / Theta '
                                  'overlap where the Theta-solve runs on its own stream under the next F-solve' % max(1, args.timing)},
            'setup_s': {'generate': t_gen, 'upload_and_alloc': t_up},
        }
        if not args.no_cpu_baseline and world == 1 and not inproc:
            gpu_file = None
            if state_file and missing:
                # parity evidence at this size (untimed): the GPU repeats, from the state the timed window started from,
                # kParityIters iterations that the CPU baseline's child also runs; the child compares objective and factors
                z = np.load(state_file)
                from trmf import Model
                m2 = Model.from_arrays(z['W'], z['H'], z['lag_val'], prob['lag_set'])
                with session.Session(prob['Y'], m2, missing=missing, log_norms=False, **hyper) as s2:
                    s2.run(kParityIters).download()
                gpu_file = state_file.replace('.npz', '_gpu.npz')
                np.savez(gpu_file, W=m2.W, H=m2.H, lag_val=m2.lag_val)
            base = cpu_baseline(args.config, args.cpu_iters, state_file, gpu_file)
            if base is not None:
                # the CPU runs its first `cpu_iters` iterations from the state the GPU's timed window started from; later iterations
                # run shorter CG solves, so the like-for-like GPU figure is the one over the SAME iterations, from the
                # per-iteration HIP-event phase times
                # -- wall clock around a session that repeats exactly those iterations from that state (the second of two: the first
                # warms the pool and the kernels' attributes), no phase events
                if state_file:
                    z = np.load(state_file)
                    from trmf import Model
                    n_it = args.cpu_iters
                    for attempt in range(2):
                        m3 = Model.from_arrays(z['W'], z['H'], z['lag_val'], prob['lag_set'])
                        with session.Session(prob['Y'], m3, missing=missing, log_norms=False, timing=0, **hyper) as s3:
                            s3.sync()
                            t3 = time.perf_counter(); s3.run(n_it); s3.sync(); t3 = time.perf_counter() - t3
                    base['gpu_same_window'] = {'iters': n_it, 'iter_per_s': n_it / t3, 'clock': 'wall clock around run() + sync()'}
                out['cpu_baseline'] = base
            for f in (state_file, gpu_file):
                if f and os.path.exists(f):
                    os.remove(f)
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        if replicas_note is None:
            lib.trmf_dist_finalize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
