import os, sys
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


def pytest_sessionstart(session):
    # Launch log for the instantiation-coverage check (tests/test_zz_instantiation_coverage.py): set before the library's first
    # launch and inherited by the child processes the suite starts (Fortran host, rank workers).  Only for GPU sessions.
    import os
    import tempfile
    if "GRT_LOG_LAUNCHES" not in os.environ and session.config.getoption("-m") and "gpu" in session.config.getoption("-m") and "not gpu" not in session.config.getoption("-m"):
        fd, path = tempfile.mkstemp(prefix="grt_launches_", suffix=".log")
        os.close(fd)
        os.environ["GRT_LOG_LAUNCHES"] = path
