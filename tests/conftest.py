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
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _libraries_built():
    """Both in-tree libraries (libribca_hip.so and the test-hook library next to it) exist before the first test that loads them: a no-op when
    they are up to date (mtime check), a hipcc run of about a minute on a tree that arrives without them.  Building is not a fallback: a
    product path without the HIP library still raises (multiplexed_image_annotator_amd._lib.lib)."""
    import __graft_entry__
    __graft_entry__.build()
