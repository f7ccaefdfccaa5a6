import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Native pieces are built before test modules are imported (they import pyopal_amd at
# module level). All three are no-ops when up to date; the driver's build() has normally
# produced them already.
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)   # CPU checker + AVX2 baseline
if not os.path.exists(os.path.join(ROOT, "pyopal_amd", "_results.so")):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "pyopal_amd", "csrc"), "../_results.so"], check=True)
if not os.path.exists(os.path.join(ROOT, "pyopal_amd", "libmiopal.so")):
    subprocess.run(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "pyopal_amd", "csrc")], check=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# Searches of a one-strip query over a few thousand targets are routed to the
# wavefront-per-pair kernels (host.hip, kSmallSearch). The parity tests use small databases
# to exercise the lane-per-target kernels, so that routing is off unless a test asks for it
# (the variable is read at every search).
os.environ.setdefault("MIOPAL_NO_SMALL_SEARCH", "1")


@pytest.fixture
def small_search_routing():
    """Run the test body with the production routing of small searches."""
    saved = os.environ.pop("MIOPAL_NO_SMALL_SEARCH", None)
    try:
        yield
    finally:
        if saved is not None:
            os.environ["MIOPAL_NO_SMALL_SEARCH"] = saved
