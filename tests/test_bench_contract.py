"""CPU: pieces of bench.py's contract that need no GPU -- the committed PMC traffic figure belongs to the kernel
sources in the tree (otherwise bench.py silently drops `roofline.traffic`), the byte model, the CPU-baseline worker."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_profile_belongs_to_the_current_kernel_sources():
    import bench
    table = json.load(open(os.path.join(ROOT, 'profiles', 'fsolve_traffic.json')))
    assert 'c3' in table, 'the headline configuration must carry a PMC traffic figure'
    shapes = {'c3': (100000, 10000, 40, 4), 'c5': (1000000, 50000, 64, 8)}      # n, T, k, sizeof(element)
    for config, tj in table.items():
        assert tj['config'] == config
        assert tj['kernel_source_sha256'] == bench.fsolve_source_digest(), \
            'profiles/fsolve_traffic.json[%s] is stale: re-run scripts/pmc_fsolve.sh on a GPU box and scripts/make_traffic_json.py' % config
        # byte model of SURVEY.md 8(d)
        n, T, k, s = shapes[config]
        nnz = {'c3': 9950287, 'c5': 49974995}[config]         # the synthetic generator's entry counts (seed 0), exactly
        assert tj['algorithmic_bytes'] == nnz * (4 + s + k * s) + (n + 1) * 8 + n * k * s
        assert tj['compulsory_bytes'] == nnz * (4 + s) + (n + 1) * 8 + T * k * s + n * k * s
        assert tj['traffic_bytes'] == round((2 * tj['fetch_size_kb'] + tj['write_size_kb']) * 1024)
        assert os.path.exists(os.path.join(ROOT, tj['source']))


def test_x_phase_byte_models():
    """roofline_x of the bench line: B_G and B_cg of SURVEY.md 8(d) "X-side figures" at the headline configuration."""
    import numpy as np
    import bench
    from trmf import synth
    cfg = synth.CONFIGS['c3']
    nnz, T, k, s = 9950287, 10000, 40, 4
    b_g, b_cg = bench.x_byte_models(cfg, nnz, s)
    assert b_g == nnz * (4 + s + k * s) + (T + 1) * 8 + T * (k * k + k) * s == 1737328224
    assert b_cg == T * k * k * s + 6 * T * k * s == 73600000
    r = bench.roofline_x(cfg, nnz, np.dtype(np.float32), True, 1, 0.2645, 0.529, 16.0, '1 rank')
    assert abs(r['gram']['achieved'] - b_g / 0.2645e-3 / 1e9) < 1e-6 and abs(r['gram']['frac'] - r['gram']['achieved'] / 8000.0) < 1e-12
    assert r['cg']['operator_passes_per_solve'] == 17.0 and abs(r['cg']['us_per_pass'] - 1e3 * (0.529 - 0.2645) / 17.0) < 1e-9
    assert bench.roofline_x(cfg, nnz, np.dtype(np.float32), True, 2, 0.2, 0.5, 16.0, '') is None       # one rank only


def test_cpu_baseline_worker_protocol():
    """The child process of bench.py's cpu_baseline: full-iteration time and the F / X / Theta split, one JSON line."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-baseline-worker', 'port', '--config', 'tiny',
                          '--cpu-iters', '3', '--cpu-threads', '2'], capture_output=True, text=True, timeout=300)
    line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
    r = json.loads(line)
    assert r['iters'] == 3 and r['threads'] == 2
    assert all(r[key] > 0 for key in ('seconds', 's_per_F', 's_per_X', 's_per_Theta'))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_through_the_launcher_line_with_one_rank():
    """The driver's N > 1 command with N = 1: torch.distributed.run -> NCCL(=RCCL) process group -> the library's own RCCL
    communicator (unique id broadcast, grouped in-place broadcasts on a 1-rank communicator) -> barriers and the
    MAX-over-ranks all-reduce -> one JSON line from rank 0.  With one rank every branch of the multi-GPU path runs."""
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, TRMF_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
           '--config', 'c2', '--no-cpu-baseline']
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-3000:]
    r = json.loads(lines[0])
    assert r['n_gpus'] == 1 and r['steps'] == 3 and r['value'] > 0 and r['metric'] == 'als_iterations_per_sec'
    assert r['roofline']['achieved'] > 0


@pytest.mark.gpu
def test_bench_falls_back_to_replicas_when_the_library_communicator_cannot_be_set_up():
    """bench.py --gpus N must not end without a figure because the library's own RCCL communicator failed on some rank: every rank
    drops it (agreed through torch.distributed) and runs the one-GPU solver; the line says so.  Forced here with one rank."""
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, TRMF_BENCH_FORCE_DIST='1', TRMF_BENCH_FAIL_DIST_INIT='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
           '--config', 'c2', '--no-cpu-baseline']
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-3000:]
    r = json.loads(lines[0])
    assert r['value'] > 0 and 'replicas of the one-GPU solver' in r['config']['parallelism']
    assert r['config']['degraded'] == 'replicas'          # machine-readable: value / n_gpus of this line is NOT a sharded figure


@pytest.mark.gpu
def test_bench_without_a_launcher_runs_the_in_process_multi_gpu_mode_and_repeats_its_window():
    """python bench.py --gpus 2 (no torchrun): TRMF_DEVICES inside one process (here two virtual ranks on the one device), the timed
    window repeated from the same state -- every window does the same work."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--devices', '0,0', '--steps', '4', '--warmup', '2', '--repeat', '3',
           '--config', 'c2', '--no-cpu-baseline', '--no-one-shot']
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-3000:]
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['steps'] == 4 and r['windows']['repeat'] == 3 and r['windows']['steps_each'] == 4
    assert '2 ranks' in r['config']['parallelism'] and 'threads of this process' in r['config']['parallelism']
    assert abs(r['ms_per_step'] * 4 * 3 / 1e3 - r['windows']['timed_region_s']) < 1e-9
    assert r['value_survey_protocol']['steps'] == 10 and r['value_survey_protocol']['iter_per_s'] > 0
