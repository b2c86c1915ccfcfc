import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Test batches are tiny, where the engine's size heuristic would always pick the direct convolution kernel:
# force the Winograd kernel wherever it is legal so whole-network tests cover it (individual tests switch to
# "0" = direct / "1" = heuristic with monkeypatch.setenv).
os.environ.setdefault("SSDE_WINOGRAD", "2")


def pytest_configure(config):
    # the CPU oracle (small torch ops) crawls when oneDNN fans out over the 256 hardware threads of the GPU box
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
