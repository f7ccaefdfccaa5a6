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
# to exercise the lane-per-target kernels, so that routing is off unless a test asks for it.
# (The library reads its switches from the environment once, at its first use of any of them;
# changes after that go through miopalSetTuning - the `tuning` fixture below.)
os.environ.setdefault("MIOPAL_NO_SMALL_SEARCH", "1")


class _Tuning:
    """Tuning switches of the library for the duration of one test: the interface of
    monkeypatch.setenv / delenv, through miopalSetTuning (include/miopal.h) instead of the
    process environment, which the library does not read on the search path."""

    def __init__(self):
        from pyopal_amd import _capi
        self._capi = _capi
        self._saved = {}

    def setenv(self, name, value):
        if name not in self._saved:
            self._saved[name] = self._capi.get_tuning(name)
        self._capi.set_tuning(name, value)

    def delenv(self, name, raising=True):
        if name not in self._saved:
            self._saved[name] = self._capi.get_tuning(name)
        self._capi.set_tuning(name, None)

    def restore(self):
        for name, value in self._saved.items():
            self._capi.set_tuning(name, value)
        self._saved.clear()


@pytest.fixture
def tuning():
    t = _Tuning()
    try:
        yield t
    finally:
        t.restore()


@pytest.fixture
def small_search_routing(tuning):
    """Run the test body with the production routing of small searches."""
    tuning.delenv("MIOPAL_NO_SMALL_SEARCH")
    yield
