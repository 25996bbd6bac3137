import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture()
def fast_oracle_build():
    """The ORACLE_FAST build of the CPU checker for a test (same sources, -march=native and the fast field multiplication; identical results,
    held by tests/test_oracle_kat.py::test_fast_build_*): the session-sized proofs of the second client's tests cost the plain build 20-40 s each."""
    import oracle_binding as ob
    ob.use_fast_library(True)
    yield
    ob.use_fast_library(False)
