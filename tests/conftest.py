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
