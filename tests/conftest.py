import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(GOLDEN, "host_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def built():
    """Build the HIP library + C oracle once per session (hipcc cross-compiles on CPU boxes)."""
    # Some GPU tests hand torch device tensors to the engine: torch's bundled ROCm runtime has to be the first
    # one initialised in the process (see chiron_amd/_lib.py), so import it before the library is loaded.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    import __graft_entry__ as g
    g.build()
    return True
