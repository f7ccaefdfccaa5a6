"""The C-ABI library loads and exports every symbol include/*.h declares.
No compute is attempted here (no GPU in the CPU test tier)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from pyopal_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "pyopal_amd", "csrc")], check=True)
    return _capi


def declared_functions():
    names = []
    for header in ("opal.h", "miopal.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b((?:opal|miopal)[A-Z]\w+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(capi):
    lib = capi.lib()
    names = declared_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(capi.EXPORTS) == names


def test_result_struct_layout(capi):
    # src/pyopal/opal.pxd:24-32: 6 ints, a pointer, an int
    assert ctypes.sizeof(capi.OpalSearchResult) == 40
    r = capi.OpalSearchResult()
    capi.lib().opalInitSearchResult(ctypes.byref(r))
    assert (r.scoreSet, r.endLocationTarget, r.endLocationQuery) == (0, -1, -1)
    assert (r.startLocationTarget, r.startLocationQuery, r.alignmentLength) == (-1, -1, 0)
    assert not r.alignment
    capi.lib().opalSearchResultSetScore(ctypes.byref(r), 7)
    assert (r.scoreSet, r.score) == (1, 7)


def test_fails_loudly_without_device(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert capi.lib().miopalDeviceCount() == 0
    with pytest.raises(RuntimeError, match="no supported SIMD backend"):
        capi.DeviceDatabase(np.zeros(4, dtype=np.uint8), np.array([0, 4], dtype=np.int64), 24)
