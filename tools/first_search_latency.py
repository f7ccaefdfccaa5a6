"""Scratch: T host threads make their FIRST search on their own slice of a fresh database at the same
moment (every slice needs its packed view built). Reports the wall time until the last thread has
its result and the spread of the per-thread latencies. Compare builds with MIOPAL_LIBRARY=..."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
rng = np.random.default_rng(5)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
res, off = _data.random_db(rng, np.full(n, 300))
q = _data.encode(_data.README_QUERY)
for rep in range(3):
    db = _capi.DeviceDatabase(res, off, 24)
    lat = [0.0] * T
    barrier = threading.Barrier(T)
    def worker(t):
        lo, hi = t * (n // T), (t + 1) * (n // T)
        barrier.wait()
        t0 = time.perf_counter()
        db.search(q, m, 3, 1, "score", "sw", lo, hi)
        lat[t] = time.perf_counter() - t0
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    [t.start() for t in threads]; [t.join() for t in threads]
    wall = time.perf_counter() - t0
    print(f"{T} threads, {n} targets, first searches: all done after {wall*1e3:.0f} ms; per thread "
          f"min {min(lat)*1e3:.0f} / median {sorted(lat)[T//2]*1e3:.0f} / max {max(lat)*1e3:.0f} ms", file=sys.stderr)
    db.close()
