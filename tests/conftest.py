import os
import sys

import pytest

# "no library GEMM on the hot path" is an invariant of the suite: any product / softmax of the path that would leave
# libgvd_hip.so for a torch library op raises (ops.library_fallback) instead of running silently.  Set before gvd_amd is
# imported; tests of the non-strict behaviour flip ops.set_strict() themselves.
os.environ.setdefault('GVD_STRICT', '1')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
