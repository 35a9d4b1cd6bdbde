import os
import sys

import pytest

# no roberta-base files exist offline: the tests run on the random-init text encoder stand-in (explicit opt-in, the
# product default is to raise like the reference; tests/test_host_cpu.py checks that default)
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
