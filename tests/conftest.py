import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import helpers  # noqa: E402

helpers.limit_openmp()  # before torch or the oracle start an OpenMP runtime (see helpers.cpu_budget)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import helpers
    return helpers.oracle()


@pytest.fixture(scope="session")
def everest_oracle_features(oracle_lib):
    """Oracle SIFT features of the three 1024x1024 everest fixture images (computed once per session)."""
    import helpers
    return [helpers.oracle_sift(oracle_lib, p) for p in helpers.load_everest_pixels()]


@pytest.fixture(scope="session")
def capi():
    """The ctypes plumbing over the C ABI; importing it loads libssrlcv_hip.so (no CPU fallback)."""
    from ssrlcv_amd import capi
    return capi
