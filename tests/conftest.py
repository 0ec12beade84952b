import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (runs on the MI355X box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; if someone runs the whole suite on a machine without a GPU
    # they are skipped rather than failed
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no ROCm device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
