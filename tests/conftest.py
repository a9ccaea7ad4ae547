import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# the tests' trees are small: the walk's per-entry pre-test bytes (built only for streams of >= 8192 nodes in the
# product, flatmat.hpp) are built for every stream here, so that the model fuzz and the GPU parity tests use them
os.environ.setdefault("WEPP_IX_PRE_MIN_NODES", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_bridge
    return oracle_bridge
