"""Scratch: SW score / end / full with a multi-strip query (Q > 64)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
rng = np.random.default_rng(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 300
GO = int(sys.argv[3]) if len(sys.argv) > 3 else 3
GE = int(sys.argv[4]) if len(sys.argv) > 4 else 1
q = _data.random_protein(rng, Q)
res, off = _data.random_db(rng, np.full(N, 300))
db = _capi.DeviceDatabase(res, off, 24)
for mode in ("score", "end", "full"):
    for rep in range(3):
        t = time.perf_counter(); out = db.search(q, m, GO, GE, mode, "sw"); dt = time.perf_counter() - t
    print(f"Q={Q} N={N} gaps {GO}/{GE} {mode}: {dt*1e3:.1f} ms -> {Q*300*N/dt/1e9:.0f} GCUPS", file=sys.stderr)
