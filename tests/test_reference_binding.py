"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the REFERENCE's own Python
wrapper, unmodified, loading THIS build's trmf_float32/64.so from its corelib/ directory -- INTEGRATION.md route A.
Without a GPU the call must come back with the "no HIP device" diagnostic and untouched factors, which proves the
loader glob, the symbol, the 17-argument prototype and the PyMatrix layout line up with the real caller."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PKG = '/root/reference/python/trmf'

SCRIPT = textwrap.dedent('''
    import sys, numpy as np, scipy
    for name in dir(np):                                   # the reference writes `import scipy as sp; sp.zeros(...)`
        if not name.startswith('_') and not hasattr(scipy, name):
            setattr(scipy, name, getattr(np, name))
    scipy.random, scipy.rand, scipy.randn = np.random, np.random.rand, np.random.randn
    sys.path.insert(0, sys.argv[1])
    import trmf                                            # the reference package (scratch copy)
    import scipy.sparse as smat
    Y = smat.random(60, 40, density=0.2, random_state=np.random.RandomState(0), format='csr', dtype=np.float32)
    for dtype in (np.float32, np.float64):
        m = trmf.Model.initialize(Y.astype(dtype), [1, 2, 5], 6, seed=0)
        W0, H0, T0 = m.W.copy(), m.H.copy(), m.lag_val.copy()
        trmf.train(Y.astype(dtype), m, lambdaI=0.5, lambdaAR=50, lambdaLag=0.5, max_iter=3, missing=True, threads=2, verbose=0)
        assert np.array_equal(m.W, W0) and np.array_equal(m.H, H0) and np.array_equal(m.lag_val, T0)
    print('BINDING_OK')
''')


@pytest.mark.skipif(not os.path.isdir(REF_PKG), reason='reference package not present (build container only)')
def test_reference_wrapper_binds_to_this_library(tmp_path):
    from trmf import session
    import numpy as np
    if session.lib_for(np.float32).trmf_device_count() > 0:
        pytest.skip('a GPU is present: the call would train; this test checks the no-device contract')
    pkg = tmp_path / 'trmf'
    shutil.copytree(REF_PKG, pkg)
    for name in ('trmf_float32.so', 'trmf_float64.so'):
        shutil.copy(os.path.join(ROOT, 'exp-trmf-nips16_amd', 'trmf', 'corelib', name), pkg / 'corelib' / name)
    res = subprocess.run([sys.executable, '-c', SCRIPT, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert 'BINDING_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
    assert res.stderr.count('no HIP device') >= 2          # one diagnostic per element-type library
