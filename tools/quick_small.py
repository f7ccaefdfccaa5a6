"""Scratch: latency of small searches (launch- and sync-bound regime)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
q = _data.encode(_data.README_QUERY)
rng = np.random.default_rng(1)
for N in (4, 1000, 20000):
    res, off = _data.random_db(rng, rng.integers(50, 400, size=N))
    db = _capi.DeviceDatabase(res, off, 24)
    for algo in ("sw", "nw"):
        for mode in ("score", "end", "full"):
            for _ in range(5): db.search(q, m, 3, 1, mode, algo)
            t = time.perf_counter()
            for _ in range(50): db.search(q, m, 3, 1, mode, algo)
            dt = (time.perf_counter() - t) / 50
            print(f"N={N:6d} {algo} {mode:5s}: {dt*1e6:8.1f} us per search", file=sys.stderr)
    db.close()
