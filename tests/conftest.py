import os
import sys

import pytest

# fault injection into the persistent launches (mmqg_persist_set_test_fault) is refused unless the process says so
os.environ.setdefault("MMQG_ENABLE_TEST_HOOKS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Observed parity errors of every comparison made through golden_util.close (one line each:
    what, max abs err, max|want|, relative, tolerance)."""
    try:
        import golden_util
        golden_util.dump_parity_log(os.path.join(ROOT, "gpurun_out", "parity_errors.tsv"))
    except Exception:
        pass
