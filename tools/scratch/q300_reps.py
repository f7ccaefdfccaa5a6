import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
res, off = _data.random_db(np.random.default_rng(1), np.full(1_000_000, 300))
q = _data.random_protein(np.random.default_rng(4), 300)
db = _capi.DeviceDatabase(res, off, 24)
r = None
ts = []
for _ in range(12):
    t0 = time.perf_counter(); r = db.search(q, m, 3, 1, "full", "sw", reuse=r); ts.append(time.perf_counter() - t0)
print(os.environ.get("MIOPAL_PARKED_WORKSPACE_MB"), " ".join(f"{t*1e3:.0f}" for t in ts), flush=True)
