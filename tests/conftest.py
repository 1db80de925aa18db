import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a `-m gpu` run on a box without a GPU must fail loudly, not silently pass on a fallback
    import torch

    if torch.cuda.is_available():
        return
    if (config.getoption("-m") or "").strip() == "gpu":
        return  # GPU tests were asked for explicitly: let them fail loudly without a GPU
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
