import os
import sys

import pytest

# a kernel fault should end the test run, not spend minutes writing a GPU core dump first (must be set before the runtime starts)
os.environ.setdefault("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _attribute_async_gpu_errors(request):
    """Launches are asynchronous: a fault raised by one test's kernels would otherwise surface at the first host
    synchronisation of a LATER test.  Synchronising after every GPU test pins it to the test that caused it."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
