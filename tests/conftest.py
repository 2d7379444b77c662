import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ctx():
    """One engine context on cuda:0 for the whole GPU session.  No fallback: if the HIP
    library or the device is missing this raises and the GPU tests fail loudly."""
    from splicedice_amd.engine import Context
    c = Context(0)
    yield c
    c.close()
