"""Round 5: phase timers of one `full` search (MIOPAL_PHASE_TIMING) for a configuration of the switches.
usage: r05_phases.py N Q [NAME=VALUE ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
N = int(sys.argv[1]); Q = int(sys.argv[2])
sw = dict(a.split("=", 1) for a in sys.argv[3:])
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
res, off = _data.random_db(np.random.default_rng(1), np.full(N, 300))
q = _data.encode(_data.README_QUERY) if Q == 53 else _data.random_protein(np.random.default_rng(4), Q)
db = _capi.DeviceDatabase(res, off, 24)
with _capi.tuning(**sw):
    r = None
    for _ in range(3):
        r = db.search(q, m, 3, 1, "full", "sw", reuse=r)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = db.search(q, m, 3, 1, "full", "sw", reuse=r); ts.append(time.perf_counter() - t0)
    print(sw, "median ms", np.median(ts) * 1e3, flush=True)
    with _capi.tuning(PHASE_TIMING="1"):
        r = db.search(q, m, 3, 1, "full", "sw", reuse=r)
