import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'exp-trmf-nips16_amd'), os.path.join(ROOT, 'oracle'), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


# The library reads its test-only knobs (forced forms, forced failures, ablations: TRMF_GRAMX, TRMF_P2P_FAIL, TRMF_NO_HV_TILE, ...)
# only when TRMF_TEST is set; worker processes of the multi-rank tests inherit it.
os.environ.setdefault('TRMF_TEST', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _gpu_available():
    try:
        from trmf import session
        import numpy as np
        return session.lib_for(np.float32).trmf_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope='session')
def have_gpu():
    return _gpu_available()


def pytest_collection_modifyitems(config, items):
    # -m gpu tests must FAIL (not skip) on a GPU box if the HIP path is unusable; on a box without
    # a GPU they are deselected by `-m "not gpu"` in the driver, and skipped if someone runs them.
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason='no HIP device in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
