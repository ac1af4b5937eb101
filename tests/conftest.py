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
    import __graft_entry__ as g
    g.build()
    return True
