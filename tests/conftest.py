import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # HOS_POISON=1 python -m pytest tests -m gpu: every float torch.empty / empty_like made by the product code is filled with NaN
    # (scripts/soak_poison.py), so a test whose kernels read memory nobody wrote fails instead of passing on stale data
    if os.environ.get("HOS_POISON") == "1":
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import torch
        import soak_poison
        torch.empty, torch.empty_like = soak_poison.pempty, soak_poison.pempty_like
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
