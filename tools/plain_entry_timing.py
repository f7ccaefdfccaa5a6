"""Time of the reference's own entry point, opalSearchDatabase (N host pointers per call: upload,
pack, search, N result structs), on the cfg2 workload. Usage: plain_entry_timing.py [targets] [calls]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data  # noqa: E402
from pyopal_amd import _capi  # noqa: E402
from pyopal_amd.matrices import ScoringMatrix  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L = 300
rng = np.random.default_rng(1)
res, off = _data.random_db(rng, np.full(N, L))
q = np.ascontiguousarray(_data.encode(_data.README_QUERY))
m = np.ascontiguousarray(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32).ravel()
lib = _capi.lib()

ptrs = (res.ctypes.data + off[:-1]).astype(np.uint64)
lens = np.diff(off).astype(np.int32)


class Result(ctypes.Structure):
    _fields_ = [("scoreSet", ctypes.c_int), ("score", ctypes.c_int), ("endLocationTarget", ctypes.c_int),
                ("endLocationQuery", ctypes.c_int), ("startLocationTarget", ctypes.c_int),
                ("startLocationQuery", ctypes.c_int), ("alignment", ctypes.c_void_p),
                ("alignmentLength", ctypes.c_int)]


results = np.zeros(N, dtype=np.dtype(Result))
rptrs = (results.ctypes.data + np.arange(N, dtype=np.uint64) * ctypes.sizeof(Result)).astype(np.uint64)
cells = float(len(q)) * float(off[-1])
for k in range(CALLS):
    results["scoreSet"] = 0
    t0 = time.perf_counter()
    rc = lib.opalSearchDatabase(q.ctypes.data, len(q), ptrs.ctypes.data, N, lens.ctypes.data, 3, 1, m.ctypes.data, 24,
                                rptrs.ctypes.data, 0, 3, 1)   # OPAL_SEARCH_SCORE, OPAL_MODE_SW, OPAL_OVERFLOW_BUCKETS
    dt = time.perf_counter() - t0
    assert rc == 0, _capi.last_error()
    assert results["scoreSet"].all()
    print(f"call {k}: {dt * 1e3:8.1f} ms  {cells / dt / 1e9:8.1f} GCUPS (PCIe-inclusive)  checksum {int(results['score'].sum())}")
