import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full golden sweep, opt-in via MQS_FULL_GOLDEN=1")


def _gpu_available():
    try:
        import mqslam_amd
        return mqslam_amd.loaded and mqslam_amd._lib.lib().mqs_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def golden3():
    return np.load(os.path.join(ROOT, "tests", "golden", "test_3_golden.npz"))


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as co
    co.build()
    return co


@pytest.fixture(scope="session")
def host_math():
    """ctypes handle of the device arithmetic compiled for the host (test-only, see host_math.cpp)."""
    import ctypes
    subprocess.check_call(["make", "-s", "-C", ROOT, "tests/libhost_math.so"])
    return ctypes.CDLL(os.path.join(ROOT, "tests", "libhost_math.so"))


@pytest.fixture(scope="session")
def mqs():
    import mqslam_amd
    return mqslam_amd


@pytest.fixture(scope="session")
def gpu(mqs):
    if not _gpu_available():
        pytest.fail("GPU test selected but libmqslam_hip.so / a HIP device is not available "
                    "(%r)" % (mqs._lib.load_error,))
    return mqs
