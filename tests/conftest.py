import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# Hermetic plans: the library's default ("specialise_at_create" = 1) makes `create` pick up run-time kernels that an earlier
# test -- or an earlier run on this machine -- left in the on-disk code-object cache.  The suite pins the policy to 0 (read once,
# before the first plan) and points the cache at a directory of its own; the tests of the cache and of the policy set both
# themselves (fourier_hip_set_default_option / subprocesses).
import tempfile  # noqa: E402

os.environ.setdefault("FOURIER_HIP_SPECIALISE", "0")
if "FOURIER_HIP_CACHE_DIR" not in os.environ:
    import atexit
    import shutil

    _cache_dir = tempfile.mkdtemp(prefix="fourier-hip-test-cache-")  # one per pytest process (and per xdist worker); removed at exit
    os.environ["FOURIER_HIP_CACHE_DIR"] = _cache_dir
    atexit.register(shutil.rmtree, _cache_dir, ignore_errors=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O
