import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _seed_every_test():
    """Inputs drawn without an explicit generator (a few GPU tests use torch.randn(..., device=...)) must not depend on
    which tests ran before: every test starts from the same CPU and device RNG state."""
    import torch
    torch.manual_seed(20240607)
    yield
