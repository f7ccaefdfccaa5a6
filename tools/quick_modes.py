"""Scratch timing of the non-headline configs (BASELINE.json configs[2..3])."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix

m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
if which == "cfg3":
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    rng = np.random.default_rng(1)
    q = _data.encode(_data.README_QUERY)
    res, off = _data.random_db(rng, np.full(N, 300))
    db = _capi.DeviceDatabase(res, off, 24)
    for mode in ("score", "end", "full"):
        db.search(q, m, 3, 1, mode, "sw", 0, 1000)
        t = time.time(); out = db.search(q, m, 3, 1, mode, "sw"); dt = time.time() - t
        print(f"sw {mode}: {dt*1e3:.1f} ms for {N} targets -> {53*300*N/dt/1e9:.1f} GCUPS (host API incl. D2H)")
else:
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    Q = L = 2000
    rng = np.random.default_rng(2)
    q = _data.random_protein(rng, Q)
    res, off = _data.random_db(rng, np.full(N, L))
    db = _capi.DeviceDatabase(res, off, 24)
    for algo in ("sw", "nw", "hw", "ov"):
        db.search(q, m, 3, 1, "score", algo, 0, 64)
        t = time.time(); out = db.search(q, m, 3, 1, "score", algo); dt = time.time() - t
        print(f"{algo} score Q={Q} vs {N}x{L}: {dt*1e3:.1f} ms -> {Q*L*N/dt/1e9:.1f} GCUPS")
