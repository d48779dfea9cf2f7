import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_model():
    from mcfost_amd.host import model as M
    return M.build_model(M.small())


@pytest.fixture(scope="session")
def ref41_model():
    from mcfost_amd.host import model as M
    return M.build_model(M.ref41())
