import os, sys, time
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tests/_data.py") else os.getcwd()
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
    for _ in range(5): db.search(q, m, 3, 1, "full", "sw")
    print(f"==== N={N} full sw, routing {_capi.DeviceDatabase.last_full_routing()}", file=sys.stderr, flush=True)
    with _capi.tuning(PHASE_TIMING="1"):
        db.search(q, m, 3, 1, "full", "sw")
    db.close()
