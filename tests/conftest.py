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
    if os.environ.get("TD_EFENCE") == "1":
        # electric-fence run (tests/efence/): every device tensor ends at an unmapped guard range, so an out-of-bounds
        # access by any kernel faults at the access; must be installed before the first device allocation
        sys.path.insert(0, os.path.join(ROOT, "tests", "efence"))
        import install as _efence_install

        _efence_install.install()


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
