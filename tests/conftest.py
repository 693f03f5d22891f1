import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hip(pkg):
    """The HIP library; GPU tests fail loudly (never skip to a fallback) when it is missing."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    return pkg.hip_lib()


@pytest.fixture(scope="session")
def hip_lib_path():
    """Path of the built product library (CPU tests only link against / dlopen it; no compute calls without a GPU)."""
    import __graft_entry__ as g
    g.build_hip()
    return g.HIP_LIB
