"""Scratch: latency of the intra-sequence kernel on a few long targets."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
q = _data.encode(_data.README_QUERY)
rng = np.random.default_rng(7)
n, L = int(sys.argv[1]), int(sys.argv[2])
res, off = _data.random_db(rng, np.full(n, L))
db = _capi.DeviceDatabase(res, off, 24)
for _ in range(3): db.search(q, m, 3, 1, "score", "sw")
t0 = time.perf_counter()
for _ in range(5): db.search(q, m, 3, 1, "score", "sw")
dt = (time.perf_counter() - t0) / 5
print(f"{n} x {L}: {dt*1e3:.3f} ms -> {dt*1e9/(L+63):.1f} ns per step")
