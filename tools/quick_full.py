"""Scratch: steady-state timing of SW full mode."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
res, off = _data.random_db(rng, np.full(N, 300))
db = _capi.DeviceDatabase(res, off, 24)
for rep in range(3):
    t = time.perf_counter(); out = db.search(q, m, 3, 1, "full", "sw"); dt = time.perf_counter() - t
    print(f"--- full #{rep}: {dt*1e3:.1f} ms", file=sys.stderr)
