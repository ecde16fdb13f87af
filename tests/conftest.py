import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TESTDATA = os.path.join(GOLDEN, "testdata")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def testdata():
    return TESTDATA


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def tables_at_first_search():
    """The library builds a handle's derived tables only when they can pay (fmx_config_set("tables_after"): after
    n / 64 patterns, or by fmx_prepare).  The parity tests search a few thousand patterns per handle and want every
    table-served code path exercised, so for the test session the threshold is 0 -- tables at the first search, as in
    round 3; test_tables_are_built_lazily sets "auto" itself."""
    try:
        import findex_amd
        findex_amd.config_set("tables_after", "0")
    except Exception:
        pass            # no library built: the tests that need it fail on their own
    yield
