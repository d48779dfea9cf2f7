import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_gpu_first(request):
    """On a GPU box, let torch initialise its HIP context before the engine library makes its first HIP call in
    this process: initialised the other way round, torch has been seen to report "No HIP GPUs are available" when
    a test later wraps the engine's device buffers as tensors.  (No GPU: nothing happens, nothing is imported.)"""
    if os.path.exists("/dev/kfd") and any(item.get_closest_marker("gpu") for item in request.session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
                torch.zeros(1, device="cuda")
        except Exception:
            pass
    yield


@pytest.fixture(scope="session")
def small_model():
    from mcfost_amd.host import model as M
    return M.build_model(M.small())


@pytest.fixture(scope="session")
def ref41_model():
    from mcfost_amd.host import model as M
    return M.build_model(M.ref41())
