import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
res, off = _data.random_db(np.random.default_rng(1), np.full(1_000_000, 300))
q = _data.encode(_data.README_QUERY)
db = _capi.DeviceDatabase(res, off, 24)
r = None
for _ in range(5): r = db.search(q, m, 3, 1, "full", "sw", reuse=r)
ts = []
for _ in range(101):
    t0 = time.perf_counter(); r = db.search(q, m, 3, 1, "full", "sw", reuse=r); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.sort(ts)
print("cfg3 full, 101 searches: min %.2f p10 %.2f p25 %.2f median %.2f p75 %.2f p90 %.2f max %.2f ms" % (ts[0], ts[10], ts[25], ts[50], ts[75], ts[90], ts[-1]))
with _capi.tuning(PHASE_TIMING="1"):
    db.search(q, m, 3, 1, "full", "sw", reuse=r)
