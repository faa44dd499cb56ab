import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The parity tests are about the HIP path: the size dispatch of the drop-in entry points (tiny products on the host,
# gf2_small_host.cpp) is switched off for the whole suite; the tests of the dispatch itself switch it back on.
os.environ.setdefault("M4RI_HIP_HOST_SMALL_WORK", "0")

# Launch census (tests/test_zz_kernel_census.py): child processes of the suite (fuzzers, the C++ friendly-layer test, the two-rank
# bench rehearsal) append their kernel launch counts to this file when they exit; the main process is asked directly.
import tempfile  # noqa: E402

if "M4RI_HIP_KERNEL_CENSUS_FILE" not in os.environ:
    _fd, _path = tempfile.mkstemp(prefix="gf2_kernel_census_", suffix=".txt")
    os.close(_fd)
    os.environ["M4RI_HIP_KERNEL_CENSUS_FILE"] = _path


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure libm4ri_hip.so and the oracle exist (compiles them if needed)."""
    import __graft_entry__ as g
    g.build()
    return True
