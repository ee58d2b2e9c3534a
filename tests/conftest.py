import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_eref():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "eref_toy.npz"))


def _torch_claims_the_gpu_first():
    """Tests that use both torch (sample generation on the device) and libpalace_hip.so in one process: torch's bundled HIP
    runtime has to be the one that is loaded -- if the library's first HIP call came first, the system runtime would be
    loaded beside torch's and torch would later report that no device exists (measured: tools/dbg/torch_order.py).
    bench.py imports torch first for the same reason.  Without a GPU this does nothing."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except ImportError:
        pass


def pytest_collection_modifyitems(config, items):
    # only sessions that really run a GPU test initialise the device (after collection, before the first test touches the
    # library); host-only selections -- parsers, scripts, the sanitizer build -- leave the GPU alone
    if any(it.get_closest_marker("gpu") is not None for it in items if not it.get_closest_marker("skip")):
        deselected = config.getoption("-m") or ""
        if "not gpu" not in deselected:
            _torch_claims_the_gpu_first()
