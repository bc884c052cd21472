import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure). Built on demand from oracle/ with g++."""
    import subprocess
    from hyperslam_amd import _lib
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("capi.cpp", "hs_math.hpp", "hs_factors.hpp", "hs_problem.hpp")]
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return _lib.Library(path, "hso_")


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly if it is missing."""
    from hyperslam_amd import _lib
    return _lib.load()
