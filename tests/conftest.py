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


@pytest.fixture(autouse=True)
def _launch_log_test_id(request):
    # every launch the library logs carries the test that caused it (grt_kernels.hip:note_launch reads GRT_TEST_ID; child processes --
    # the Fortran host, rank workers, tools -- inherit it): the coverage check asks not only that an instantiation ran but that a test
    # which compares with the compiled reference, its goldens or the oracle ran it
    old = os.environ.get("GRT_TEST_ID")
    os.environ["GRT_TEST_ID"] = getattr(request.node, "originalname", None) or request.node.name
    yield
    if old is None:
        os.environ.pop("GRT_TEST_ID", None)
    else:
        os.environ["GRT_TEST_ID"] = old
