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
    """The product package.  PLBA_TEST_FORCE_FUSED=1 (read HERE, by the test harness — the library reads no environment): every problem a
    test creates without saying otherwise takes the fused landmark passes wherever its structure fits (options.lm_fused_min_obs = 1), so
    that the whole default-option suite exercises them on its small windows:  PLBA_TEST_FORCE_FUSED=1 pytest tests -m gpu"""
    import __graft_entry__ as g
    p = g.load_package()
    if os.environ.get("PLBA_TEST_FORCE_FUSED") == "1" and not getattr(p, "_forced_fused", False):
        plain = p.new_problem

        def forced(**opts):
            if "lm_fused" not in opts and "lm_fused_min_obs" not in opts:
                opts["lm_fused_min_obs"] = 1
            return plain(**opts)
        p.new_problem = forced
        p._forced_fused = True
    return p


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
