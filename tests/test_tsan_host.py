"""CPU: ThreadSanitizer over the library's multi-threaded HOST code with the HIP runtime replaced by host stand-ins (tests/tsan_host.cpp):
the process-level device pool + stream cache + pinned upload ring with its copy threads (csrc/device_pool.hpp), and the in-process
communicator of the TRMF_DEVICES session groups (csrc/comm.hpp: ThreadGroup, ThreadComm).  Any reported race fails."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_threads_are_race_free_under_tsan(tmp_path):
    csrc = os.path.join(ROOT, 'exp-trmf-nips16_amd', 'csrc')
    pool = open(os.path.join(csrc, 'device_pool.hpp')).read()
    a = pool.index('namespace trmf {')
    (tmp_path / 'pool_part.hpp').write_text(pool[a:])
    comm = open(os.path.join(csrc, 'comm.hpp')).read()
    base = comm[comm.index('struct Comm {'):comm.index('// ---- equal-slot staging shared by the communicators')]
    grp = comm[comm.index('// ---- ranks as threads of one process'):comm.index('// Minimal view of the RCCL C API')]
    (tmp_path / 'comm_part.hpp').write_text('namespace trmf {\n' + base + grp + '}\n')
    exe = str(tmp_path / 'tsan_host')
    subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=thread', '-DTSAN_POOL_HEADER="pool_part.hpp"', '-DTSAN_COMM_HEADER="comm_part.hpp"',
                    '-I', str(tmp_path), os.path.join(ROOT, 'tests', 'tsan_host.cpp'), '-o', exe, '-pthread'], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS='halt_on_error=0 report_signal_unsafe=0'))
    report = res.stdout + res.stderr
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        open(os.path.join(out, 'tsan_host.txt'), 'w').write(report)
    assert 'WARNING: ThreadSanitizer' not in report, report[-6000:]
    assert res.returncode == 0, report[-3000:]
    assert report.count(': ok') == 6, report
