"""pytest in THIS process (so that tools/with_cflags.py can point the tests at a variant build of the library):
    python tools/with_cflags.py <tag> "<cflags>" tools/run_pytest.py tests/test_gpu_kernels.py -q -k n192"""
import sys

import pytest

sys.exit(pytest.main(sys.argv[1:]))
