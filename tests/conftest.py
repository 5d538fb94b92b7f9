import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _poison_free_gpu_memory(request):
    """GPU tests run against memory full of NaNs: a kernel that reads a buffer element nobody wrote (fresh
    allocations are zero pages, which hides such reads when a test runs alone) then fails every time instead of
    depending on test order."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        blocks = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 26, 1 << 24, 1 << 24, 1 << 22, 1 << 22, 1 << 20, 1 << 20, 1 << 18, 1 << 16)]
        torch.cuda.synchronize()
        del blocks
    yield
