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


def test_reference_entry_point_without_device(capi):
    """opalSearchDatabase exactly as the reference binds it (src/pyopal/opal.pxd:38-52): argument errors are
    reported before anything touches a device, an empty database is answered at once (the reference
    returns 0 for it), and without a GPU the call fails with the reference's own "no SIMD support" code -
    never with a CPU fallback. The caches it keeps between calls can be dropped at any time."""
    import torch
    lib = capi.lib()
    q = np.array([0, 1, 2, 3], dtype=np.uint8)
    m = np.ones(24 * 24, dtype=np.int32)
    seq = np.array([1, 2, 3, 4, 5], dtype=np.uint8)
    ptrs = np.array([seq.ctypes.data], dtype=np.uint64)
    lens = np.array([5], dtype=np.int32)
    res = capi.OpalSearchResult()
    lib.opalInitSearchResult(ctypes.byref(res))
    rptr = (ctypes.POINTER(capi.OpalSearchResult) * 1)(ctypes.pointer(res))

    def call(db_length, alphabet=24, lengths=lens):
        return lib.opalSearchDatabase(q.ctypes.data, len(q), ptrs.ctypes.data, db_length, lengths.ctypes.data, 3, 1,
                                      m.ctypes.data, alphabet, ctypes.cast(rptr, ctypes.c_void_p), 0, 3, 1)

    assert call(0) == 0                                   # nothing to search
    assert call(1, alphabet=0) == 101                     # MIOPAL_ERR_BAD_ARGUMENT
    assert "alphabet length" in capi.last_error()
    assert call(1, lengths=np.array([-1], dtype=np.int32)) == 101
    assert "negative sequence length" in capi.last_error()
    lib.miopalReleaseCaches()                             # (nothing kept: a no-op)
    if not torch.cuda.is_available():
        assert call(1) == capi.OPAL_ERR_NO_SIMD_SUPPORT
        assert res.scoreSet == 0                          # no result was made up
        lib.miopalReleaseCaches()


@pytest.mark.timeout(60)
def test_view_cache_survives_a_throwing_builder(capi):
    # host.hip getViewWith: a builder that throws (std::bad_alloc from a million-entry vector, or
    # whatever parallelSlices rethrows) must not leave its placeholder behind - every later search of
    # the slice would wait on it forever. Needs no device: the builders are injected (miopalSelfTest).
    assert capi.lib().miopalSelfTest(1) == 0
    assert capi.lib().miopalSelfTest(99) != 0    # unknown test number


def test_operations_unpacked_from_two_bits_each(capi):
    # the host half of copy_out_packed_kernel (host_workspace.inc, unpack::): table / pdep / AVX-512 VBMI, whichever this
    # CPU has, threaded and with the crew started beforehand, on ranges that start and end anywhere. Needs no device.
    assert capi.lib().miopalSelfTest(2) == 0


def test_tuning_switches_are_arguments_not_environment(capi, monkeypatch):
    """include/miopal.h, miopalSetTuning: the library reads MIOPAL_* from the environment once (tests/conftest.py
    sets MIOPAL_NO_SMALL_SEARCH before the first use); afterwards the environment is not looked at again - a
    putenv on another thread cannot race a search - and switches change through the call. Needs no device."""
    assert capi.get_tuning("MIOPAL_NO_SMALL_SEARCH") == "1"          # from the environment, at first use
    monkeypatch.setenv("MIOPAL_NO_BIASED", "1")
    assert capi.get_tuning("NO_BIASED") is None                       # the environment is not read again
    with capi.tuning(NO_BIASED="1", NO_SMALL_SEARCH=None):
        assert capi.get_tuning("MIOPAL_NO_BIASED") == "1" and capi.get_tuning("NO_SMALL_SEARCH") is None
    assert capi.get_tuning("NO_BIASED") is None and capi.get_tuning("NO_SMALL_SEARCH") == "1"
    assert capi.lib().miopalSetTuning(b"NO_SUCH_SWITCH", b"1") == 101
    assert "unknown tuning switch" in capi.last_error()
    assert capi.lib().miopalDbSetOption(None, b"reserve_cus", 8) == 101
    # every switch of the table is documented by name in tuning.h and none is read with getenv elsewhere
    src = os.path.join(ROOT, "pyopal_amd", "csrc")
    calls = []
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".inc", ".h")):
            text = open(os.path.join(src, name)).read()
            calls += [(name, m.start()) for m in re.finditer(r"\bgetenv\s*\(", re.sub(r"//.*", "", text))]
    assert [c[0] for c in calls] == ["host.hip"], calls             # Tuning::fromEnv, and nothing else
    # the table of switches in INTEGRATION.md is the list of tuning.h, name by name, and the library knows each
    listed = re.findall(r"X\(([A-Z_0-9]+)\)", open(os.path.join(src, "tuning.h")).read().split("enum class Tune")[0])
    documented = re.findall(r"^\| `([A-Z_0-9]+)` \|", open(os.path.join(ROOT, "INTEGRATION.md")).read(), re.M)
    assert sorted(listed) == sorted(documented), set(listed) ^ set(documented)
    for name in listed:
        assert capi.lib().miopalSetTuning(name.encode(), capi.get_tuning(name).encode() if capi.get_tuning(name) else None) == 0, name


def test_logical_devices_hook_needs_a_physical_device(capi):
    import torch
    lib = capi.lib()
    assert lib.miopalTestSetLogicalDevices(65) == 101
    assert lib.miopalTestSetLogicalDevices(3) == 0
    try:
        # ordinals only exist on top of a real gfx950: without one the count stays 0
        assert lib.miopalDeviceCount() == (3 if torch.cuda.is_available() else 0)
    finally:
        assert lib.miopalTestSetLogicalDevices(0) == 0
